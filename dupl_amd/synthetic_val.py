"""Synthetic validation samples (platform independent, same hash generators as dupl_amd.synthetic): images of
different native sizes with a blocky ground-truth label map, used by the validation / evaluation parity tests."""
from __future__ import annotations

import numpy as np
import torch

from .synthetic import hash_normal, hash_randint, hash_uniform


def synthetic_val_samples(sizes=((75, 100), (96, 64), (110, 90)), num_fg: int = 20, seed: int = 31):
    """[(inputs (1,3,H,W) fp32, labels (1,H,W) int64 in {0..num_fg, 255}, cls_label (1,num_fg) fp32)]."""
    out = []
    for i, (H, W) in enumerate(sizes):
        x = hash_normal(f"val_x{i}", (1, 3, H, W), std=1.0, seed=seed)
        x = torch.as_tensor(np.asarray(x), dtype=torch.float32)
        gh, gw = (H + 15) // 16, (W + 15) // 16
        blocks = hash_randint(f"val_lab{i}", (gh, gw), 0, num_fg + 3, seed=seed)
        blocks = np.where(blocks == num_fg + 1, 255, np.where(blocks == num_fg + 2, 0, blocks))
        lab = np.repeat(np.repeat(blocks, 16, axis=0), 16, axis=1)[:H, :W]
        cls = np.zeros((1, num_fg), dtype=np.float32)
        present = np.unique(lab[(lab > 0) & (lab < 255)])
        u = hash_uniform(f"val_cls{i}", num_fg, seed, 0)
        for c in present:
            if u[c - 1] < 0.6:
                cls[0, c - 1] = 1.0
        if cls.sum() == 0:
            cls[0, (present[0] - 1) if len(present) else 0] = 1.0
        out.append((x, torch.from_numpy(lab.astype(np.int64))[None], torch.from_numpy(cls)))
    return out
