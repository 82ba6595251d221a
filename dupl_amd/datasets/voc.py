"""PASCAL VOC 2012: category names (datasets/voc.py:14) and the dataset classes of the loops (datasets/voc.py:22-266)."""
class_list = ["bg", "aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "table", "dog",
              "horse", "motorbike", "person", "plant", "sheep", "sofa", "train", "tvmonitor"]


# ---------------------------------------------------------------------------------------------------------------
# Datasets (datasets/voc.py:22-266).  Items are RAW: the decoded uint8 image plus, for train items, the geometry
# drawn on the host in the reference's random-number order; the pixel work happens on the device
# (datasets/device_loader.py).  JPEG / PNG decoding is PIL (imageio, the reference's reader, is a PIL front-end for
# these formats and is absent from this image).
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from .transforms import draw_geometry, draw_train_views


def load_img_name_list(img_name_list_path):
    return np.loadtxt(img_name_list_path, dtype=str)


def load_cls_label_list(name_list_dir):
    return np.load(os.path.join(name_list_dir, "cls_labels_onehot.npy"), allow_pickle=True).item()


def _read_rgb(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.array(im.convert("RGB"))      # a writable copy: the items become torch tensors


def _read_label(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.array(im)


class VOC12Dataset(Dataset):
    """datasets/voc.py:22-62: (img_name, image uint8 HWC, label)."""

    def __init__(self, root_dir=None, name_list_dir=None, split="train", stage="train"):
        super().__init__()
        self.root_dir, self.stage = root_dir, stage
        self.img_dir = os.path.join(root_dir, "JPEGImages")
        self.label_dir = os.path.join(root_dir, "SegmentationClassAug")
        self.name_list_dir = os.path.join(name_list_dir, split + ".txt")
        self.name_list = load_img_name_list(self.name_list_dir)

    def __len__(self):
        return len(self.name_list)

    def __getitem__(self, idx):
        name = str(self.name_list[idx])
        image = _read_rgb(os.path.join(self.img_dir, name + ".jpg"))
        if self.stage in ("train", "val"):
            label = _read_label(os.path.join(self.label_dir, name + ".png"))
        else:
            label = image[:, :, 0]
        return name, image, label


class VOC12ClsDataset(VOC12Dataset):
    """datasets/voc.py:65-186 with aug=True: raw train item `(img_name, raw uint8 (h,w,3) tensor, cls_label, Geometry)`;
    with aug=False `(img_name, raw, cls_label)`.  DeviceLoader turns a batch of them into the reference's tuple."""

    def __init__(self, root_dir=None, name_list_dir=None, split="train", stage="train", resize_range=(512, 640),
                 rescale_range=(0.5, 2.0), crop_size=512, img_fliplr=True, ignore_index=255, num_classes=21, aug=False,
                 **kwargs):
        super().__init__(root_dir, name_list_dir, split, stage)
        self.aug, self.ignore_index = aug, ignore_index
        self.rescale_range, self.crop_size, self.img_fliplr = rescale_range, crop_size, img_fliplr
        self.num_classes = num_classes
        self.photometric = kwargs.get("photometric", True)    # global_view1's jitter / grayscale / blur (voc.py:101-114)
        self.label_list = load_cls_label_list(name_list_dir=name_list_dir)

    def __getitem__(self, idx):
        name = str(self.name_list[idx])
        image = _read_rgb(os.path.join(self.img_dir, name + ".jpg"))     # the label PNG is not needed for a cls item
        cls_label = self.label_list[name]
        raw = torch.from_numpy(np.ascontiguousarray(image))
        if not self.aug:
            return name, raw, cls_label
        geo = draw_geometry(image.shape[0], image.shape[1], self.rescale_range, self.crop_size, self.img_fliplr)
        geo.photometric = draw_train_views(image.shape[0], image.shape[1]) if self.photometric else None
        return name, raw, cls_label, geo


class VOC12SegDataset(VOC12Dataset):
    """datasets/voc.py:189-266 as used by the loops (aug=False val split): raw item `(img_name, raw uint8 (h,w,3),
    label (h,w) uint8, cls_label)`; DeviceValLoader applies normalize_img on the device."""

    def __init__(self, root_dir=None, name_list_dir=None, split="train", stage="train", resize_range=(512, 640),
                 rescale_range=(0.5, 2.0), crop_size=512, img_fliplr=True, ignore_index=255, aug=False, **kwargs):
        super().__init__(root_dir, name_list_dir, split, stage)
        if aug:
            raise NotImplementedError("the training scripts build VOC12SegDataset with aug=False only "
                                      "(train_final_voc.py:135-138); its PhotoMetricDistortion path is not built")
        self.ignore_index = ignore_index
        self.label_list = load_cls_label_list(name_list_dir=name_list_dir)

    def __getitem__(self, idx):
        name, image, label = super().__getitem__(idx)
        cls_label = 0 if self.stage == "test" else self.label_list[name]
        return name, torch.from_numpy(np.ascontiguousarray(image)), label, cls_label
