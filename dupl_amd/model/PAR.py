"""PAR -- pixel-adaptive refinement module (reference: model/PAR.py:26-91) on the HIP stencil kernels.

Same constructor (dilations, num_iter), same registered buffer `kernel` (kept only for state_dict
parity: neighbours are gathered by clamped indices, not by one-hot dilated convolutions) and the same
forward(imgs, masks) -> masks contract.  `affinity()` / `propagate()` expose the two halves so the
refine wrappers can build the affinity once per image and run every (student, high/low) job of a
batch together (the reference recomputes the affinity in each of its 4 calls per image).
"""
import numpy as np
import torch
import torch.nn as nn

from .. import ops


def get_kernel():
    """The reference's one-hot 3x3 neighbour selectors (PAR.py:10-24); state_dict parity only."""
    weight = torch.zeros(8, 1, 3, 3)
    for i, (r, c) in enumerate(((0, 0), (0, 1), (0, 2), (1, 0), (1, 2), (2, 0), (2, 1), (2, 2))):
        weight[i, 0, r, c] = 1
    return weight


class PAR(nn.Module):
    def __init__(self, dilations, num_iter):
        super().__init__()
        self.dilations = [int(d) for d in dilations]
        self.num_iter = num_iter
        self.register_buffer("kernel", get_kernel())
        self.dim = 2
        self.w1 = 0.3
        self.w2 = 0.01
        if len(self.dilations) * 8 > 48:
            raise ValueError("at most 6 dilations (48 neighbours) are supported by the HIP kernels")
        self._pos_host = ops.par_pos_term(self.dilations, self.w1, self.w2)
        self._pos_dev = {}

    def get_pos(self):
        ker = torch.ones(1, 1, 8, 1, 1)
        for i in (0, 2, 5, 7):
            ker[0, 0, i, 0, 0] = np.sqrt(2)
        return torch.cat([ker * d for d in self.dilations], dim=2)

    def _pos(self, device):
        key = str(device)
        if key not in self._pos_dev:
            self._pos_dev[key] = torch.from_numpy(self._pos_host).to(device)
        return self._pos_dev[key]

    def affinity(self, imgs):
        """(b,3,h,w) -> (b,48,h,w): colour softmax + w2 * positional softmax (PAR.py:66-85)."""
        return ops.par_affinity(imgs.contiguous().float(), self.dilations, self._pos(imgs.device))

    def propagate(self, aff, masks, job_img, job_K):
        return ops.par_propagate(aff, masks, job_img, job_K, self.dilations, self.num_iter)

    def forward(self, imgs, masks):
        if not imgs.is_cuda:
            raise RuntimeError("dupl_amd.PAR runs on an MI355X only (no CPU path)")
        b, c, h, w = imgs.shape
        if tuple(masks.shape[-2:]) != (h, w):   # F.interpolate(..., align_corners=True) of PAR.py:66
            masks = ops.resize_bilinear(masks.contiguous().float(), h, w, align_corners=True)
        masks = masks.contiguous().float()
        K = masks.shape[1]
        aff = self.affinity(imgs)
        job_img = torch.arange(b, dtype=torch.int32, device=imgs.device)
        job_K = torch.full((b,), K, dtype=torch.int32, device=imgs.device)
        return self.propagate(aff, masks.clone(), job_img, job_K)
