"""Input de-normalisation for PAR and the per-step strong augmentation (reference: utils/imutils.py:17-31,305-317)."""
import random

import torch

from .. import ops
from .._lib import lib as _L

# randomaug.augment_list() (utils/randomaug.py:161-198): (op, minval, maxval)
AUGMENT_LIST = (("AutoContrast", 0, 1), ("Equalize", 0, 1), ("Posterize", 0, 6), ("Color", 0.1, 1.9),
                ("Contrast", 0.1, 1.9), ("Brightness", 0.1, 1.9), ("Sharpness", 0.1, 1.9))


def denormalize_img(imgs=None, mean=None, std=None):
    """(x*std+mean) -> uint8 truncation (returned as a uint8 tensor like the reference)."""
    return (ops.denormalize_img(imgs.contiguous().float(), mean, std) * 255.0).round().to(imgs.device).byte()


def denormalize_img2(imgs=None):
    """uint8-truncated image / 255 as float32 (imutils.py:27-31)."""
    return ops.denormalize_img(imgs.contiguous().float())


def rand_augment_ops(n, m):
    """RandAugment.__call__'s draw (utils/randomaug.py:259-265) from Python's global `random` stream."""
    return [(name, (float(m) / 30) * float(hi - lo) + lo) for name, lo, hi in random.choices(AUGMENT_LIST, k=n)]


def augment_data_strong(images, n=4, m=20, ops_per_image=None):
    """imutils.py:305-317 without the GPU -> PIL -> GPU hop: per image ToPILImage -> RandAugment(n, m) -> ToTensor ->
    Normalize -> flip along W, in Pillow's exact 8-bit arithmetic on the device (csrc/augment.hip).  `images`
    (b,3,H,W): de-normalised floats in [0,1] on the GPU; the op draw uses `random.choices` like the reference (pass
    ops_per_image = [[(name, value), ...], ...] to fix it).  Returns a new (b,3,H,W) float32 tensor."""
    b, C, H, W = images.shape
    assert C == 3 and images.is_cuda
    x = images.contiguous().float()
    dev = x.device
    L, st = _L(), ops._stream()
    u8 = torch.empty((b, 3, H, W), device=dev, dtype=torch.uint8)
    tmp = torch.empty((3, H, W), device=dev, dtype=torch.uint8)
    hist = torch.empty((768,), device=dev, dtype=torch.int32)
    lut = torch.empty((768,), device=dev, dtype=torch.uint8)
    lsum = torch.empty((1,), device=dev, dtype=torch.int64)
    out = torch.empty((b, 3, H, W), device=dev, dtype=torch.float32)
    L.dupl_aug_to_u8(x.data_ptr(), u8.data_ptr(), x.numel(), st)
    for i in range(b):
        cur, other = u8[i], tmp
        seq = ops_per_image[i] if ops_per_image is not None else rand_augment_ops(n, m)
        for name, val in seq:
            if name == "AutoContrast":
                L.dupl_aug_lut_op(cur.data_ptr(), H, W, 0, hist.data_ptr(), lut.data_ptr(), st)
            elif name == "Equalize":
                L.dupl_aug_lut_op(cur.data_ptr(), H, W, 1, hist.data_ptr(), lut.data_ptr(), st)
            elif name == "Posterize":
                L.dupl_aug_posterize(cur.data_ptr(), 3 * H * W, max(1, int(val)), st)
            elif name in ("Color", "Contrast", "Brightness"):
                mode = ("Color", "Contrast", "Brightness").index(name)
                L.dupl_aug_enhance(cur.data_ptr(), H, W, mode, float(val), lsum.data_ptr(), st)
            elif name == "Sharpness":
                L.dupl_aug_sharpness(cur.data_ptr(), other.data_ptr(), H, W, float(val), st)
                cur, other = other, cur
            else:
                raise ValueError(name)
        L.dupl_aug_finish(cur.data_ptr(), out[i].data_ptr(), H, W, st)
    return out
