"""Host side of the device input pipeline (SURVEY 8 f-3 ii): the RANDOM DRAWS of the reference's train-item
geometry in the reference's order, and the fixed-point coefficient tables of Pillow's bilinear resize.  No pixel is
touched on the host: the pixel work is csrc/loader.hip.

Reference composition (datasets/voc.py:134-148 `__transforms`, identical in datasets/coco.py:150-167):
    image = transforms.random_scaling(image, scale_range)            # random.uniform -> PIL resize BILINEAR
    image = transforms.random_fliplr(image)                          # random.random() > 0.5 -> np.fliplr
    image, img_box = transforms.random_crop(image, crop_size, mean_rgb=[0,0,0])
                                                                     # np.random.randint x2 (pad), random.randrange x2
    local_image = local_view(image); image = global_view1(image)     # photometric views: torch.rand / randperm /
                                                                     # uniform_ + random.random / uniform (below)
    image = T.Normalize(T.ToTensor(image))
and, back in __getitem__ (voc.py:171-177), global_view2(pil_image) -- never read by the loop, but it consumes random numbers.

Photometric views (voc.py:101-126): torchvision is a third-party dependency of the reference (0.14.1 in requirements.txt)
that is absent from this image; the DRAW ORDER of its RandomApply / ColorJitter / RandomGrayscale / RandomResizedCrop is
restated here from that version's published source (torchvision/transforms/transforms.py), the pixel arithmetic (Pillow's)
is csrc/photometric.hip and is pinned against Pillow itself.
"""
from __future__ import annotations

import math
import random
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch

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
    photometric: Optional["Photometric"] = None     # global_view1's draws (draw_train_views); None = geometry only


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


@dataclass
class Photometric:
    """The draws of one `flip_and_color_jitter` + `GaussianBlur` view (datasets/voc.py:102-114)."""
    jitter: bool                       # RandomApply([ColorJitter], p=0.8) fired
    order: Tuple[int, ...]             # ColorJitter fn_idx: 0 brightness, 1 contrast, 2 saturation, 3 hue
    brightness: float
    contrast: float
    saturation: float
    hue: float
    gray: bool                         # RandomGrayscale(p=0.2) fired
    blur_radius: Optional[float]       # transforms.GaussianBlur: radius, or None when it did not fire

    @property
    def hue_shift(self) -> int:
        """F_pil.adjust_hue: `np_h += np.uint8(hue_factor * 255)` -- the C cast truncates toward zero, uint8 wraps."""
        return int(self.hue * 255) & 0xFF


JITTER = dict(brightness=(1 - 0.4, 1 + 0.4), contrast=(1 - 0.4, 1 + 0.4), saturation=(1 - 0.2, 1 + 0.2), hue=(-0.1, 0.1))


def _tv_uniform(lo: float, hi: float) -> float:
    return float(torch.empty(1).uniform_(lo, hi))


def draw_view(blur_p: float, jitter_p: float = 0.8, gray_p: float = 0.2, radius=(0.1, 2.0)) -> Photometric:
    """One pass through Compose([RandomApply([ColorJitter(0.4, 0.4, 0.2, 0.1)], p), RandomGrayscale(p), GaussianBlur(p)]):
    torch's global generator for the torchvision part (RandomApply.forward: `if self.p < torch.rand(1): return img`;
    ColorJitter.get_params: randperm(4), then uniform_ for brightness, contrast, saturation, hue; RandomGrayscale.forward:
    `torch.rand(1) < self.p`), Python's `random` for the reference's own GaussianBlur (transforms.py:20-28)."""
    jitter = not (jitter_p < float(torch.rand(1)))
    order, b, c, s_, h = (), 1.0, 1.0, 1.0, 0.0
    if jitter:
        order = tuple(int(i) for i in torch.randperm(4))
        b = _tv_uniform(*JITTER["brightness"])
        c = _tv_uniform(*JITTER["contrast"])
        s_ = _tv_uniform(*JITTER["saturation"])
        h = _tv_uniform(*JITTER["hue"])
    gray = bool(float(torch.rand(1)) < gray_p)
    blur = None
    if random.random() <= blur_p:
        blur = random.uniform(radius[0], radius[1])
    return Photometric(jitter, order, b, c, s_, h, gray, blur)


def draw_random_resized_crop(height: int, width: int, scale=(0.4, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """torchvision 0.14.1 RandomResizedCrop.get_params: up to 10 x (uniform_ area, uniform_ log-ratio), randint x2 on
    success, no draw for the central-crop fallback.  Returns (i, j, h, w)."""
    area = height * width
    log_ratio = torch.log(torch.tensor(ratio))
    for _ in range(10):
        target_area = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
        aspect_ratio = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])).item()
        w = int(round(math.sqrt(target_area * aspect_ratio)))
        h = int(round(math.sqrt(target_area / aspect_ratio)))
        if 0 < w <= width and 0 < h <= height:
            i = torch.randint(0, height - h + 1, size=(1,)).item()
            j = torch.randint(0, width - w + 1, size=(1,)).item()
            return i, j, h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def draw_train_views(height: int, width: int) -> Photometric:
    """Every photometric draw of one train item in the reference's order -- local_view (blur p = 0.5), global_view1
    (p = 1.0) inside `__transforms` (voc.py:145-146), then global_view2 on the undistorted image in `__getitem__`
    (voc.py:175: RandomResizedCrop, the jitter view with blur p = 0.1, Solarization's random.random()).  Only global_view1
    reaches the training loop (`image`); the other two are drawn so the generators stay in step with the reference's."""
    draw_view(0.5)                                   # local_view -> crops[2], never read (train_final_voc.py:180)
    g1 = draw_view(1.0)
    draw_random_resized_crop(height, width)          # global_view2 -> crops[1], never read
    draw_view(0.1)
    random.random()                                  # Solarization(p=0.2)
    return g1


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
