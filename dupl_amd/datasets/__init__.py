"""Datasets of the training loops (datasets/voc.py, datasets/coco.py) in a raw-item form: workers decode and draw the
random geometry, the device does the pixel work (device_loader.py, csrc/loader.hip)."""
from . import voc, coco  # noqa: F401
