"""Input de-normalisation for PAR (reference: utils/imutils.py:17-31)."""
from .. import ops


def denormalize_img(imgs=None, mean=None, std=None):
    """(x*std+mean) -> uint8 truncation (returned as a uint8 tensor like the reference)."""
    if mean is not None or std is not None:
        raise NotImplementedError("the HIP kernel bakes in the reference's default ImageNet mean/std")
    return (ops.denormalize_img(imgs.contiguous().float()) * 255.0).round().to(imgs.device).byte()


def denormalize_img2(imgs=None):
    """uint8-truncated image / 255 as float32 (imutils.py:27-31)."""
    return ops.denormalize_img(imgs.contiguous().float())
