"""Host side of the device input pipeline (SURVEY 8 f-3 ii): the RANDOM DRAWS of the reference's train-item
geometry in the reference's order, and the fixed-point coefficient tables of Pillow's bilinear resize.  No pixel is
touched on the host: the pixel work is csrc/loader.hip.

Reference composition (datasets/voc.py:134-148 `__transforms`, identical in datasets/coco.py:150-167):
    image = transforms.random_scaling(image, scale_range)            # random.uniform -> PIL resize BILINEAR
    image = transforms.random_fliplr(image)                          # random.random() > 0.5 -> np.fliplr
    image, img_box = transforms.random_crop(image, crop_size, mean_rgb=[0,0,0])
                                                                     # np.random.randint x2 (pad), random.randrange x2
    ... photometric jitter (torchvision ColorJitter / RandomGrayscale + PIL GaussianBlur): NOT built, see DESIGN 0 (f-3)
    image = T.Normalize(T.ToTensor(image))
"""
from __future__ import annotations

import math
import random
from dataclasses import dataclass
from typing import Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2      # Pillow src/libImaging/Resample.c


@dataclass
class Geometry:
    """Everything csrc/loader.hip needs for one train item, plus the img_box the item carries."""
    h: int
    w: int
    h2: int
    w2: int
    flip: bool
    h_pad: int
    w_pad: int
    h_start: int
    w_start: int
    crop: int
    img_box: np.ndarray          # int16 [y0, y1, x0, x1] inside the crop (transforms.py:191-195)


def draw_geometry(h: int, w: int, rescale_range=(0.5, 2.0), crop_size: int = 448, img_fliplr: bool = True) -> Geometry:
    """The random numbers of one `__transforms` call, drawn from Python's global `random` and numpy's global RandomState
    in the reference's order (transforms.py:59,104,162-163,172-174), and what they imply."""
    h2, w2 = h, w
    if rescale_range:
        lo, hi = rescale_range
        assert lo <= hi
        ratio = random.uniform(lo, hi)
        w2, h2 = int(ratio * w), int(ratio * h)                       # transforms.py:68: [int(scale*w), int(scale*h)]
    flip = bool(img_fliplr and random.random() > 0.5)                 # transforms.py:104-108
    H, W = max(crop_size, h2), max(crop_size, w2)
    h_pad = int(np.random.randint(H - h2 + 1))
    w_pad = int(np.random.randint(W - w2 + 1))
    h_start = random.randrange(0, H - crop_size + 1, 1)               # get_random_cropbox returns at i = 0 (no label)
    w_start = random.randrange(0, W - crop_size + 1, 1)
    box = np.asarray([max(h_pad - h_start, 0), min(crop_size, h2 + h_pad - h_start),
                      max(w_pad - w_start, 0), min(crop_size, w2 + w_pad - w_start)], dtype=np.int16)
    return Geometry(h, w, h2, w2, flip, h_pad, w_pad, h_start, w_start, crop_size, box)


def resample_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc (src/libImaging/Resample.c) for the BILINEAR filter over the
    whole axis: (coef int32 (out, ksize) in 22-bit fixed point, bounds int32 (out, 2) = (first input index, taps), ksize).
    Third-party algorithm (Pillow, the reference's resize backend, transforms.py:70) restated in double arithmetic in
    Pillow's operation order so that the integers are identical."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale                                       # bilinear support = 1.0
    ksize = int(math.ceil(support)) * 2 + 1
    coef = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    one = float(1 << PRECISION_BITS)
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        ws = []
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            wgt = 1.0 - a if a < 1.0 else 0.0
            ws.append(wgt)
            ww += wgt
        for x in range(xmax):
            k = ws[x] / ww if ww != 0.0 else ws[x]
            coef[xx, x] = int(-0.5 + k * one) if k < 0 else int(0.5 + k * one)
        bounds[xx] = (xmin, xmax)
    return coef, bounds, ksize
