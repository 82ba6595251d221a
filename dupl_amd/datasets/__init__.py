"""Dataset constants the evaluation tables need (the category names of datasets/voc.py:14 and datasets/coco.py:14).
The loaders / augmentation themselves are outside the hot path."""
from . import voc, coco  # noqa: F401
