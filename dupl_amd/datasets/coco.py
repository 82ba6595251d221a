"""MS COCO 2014: category names (datasets/coco.py:14) and the dataset classes of the loops (datasets/coco.py:24-270)."""
class_list = [
    "_background_", "person", "bicycle", "car", "motorcycle", "airplane", "bus", "train", "truck", "boat",
    "traffic light", "fire hydrant", "stop sign", "parking meter", "bench", "bird", "cat", "dog", "horse", "sheep",
    "cow", "elephant", "bear", "zebra", "giraffe", "backpack", "umbrella", "handbag", "tie", "suitcase", "frisbee",
    "skis", "snowboard", "sports ball", "kite", "baseball bat", "baseball glove", "skateboard", "surfboard",
    "tennis racket", "bottle", "wine glass", "cup", "fork", "knife", "spoon", "bowl", "banana", "apple", "sandwich",
    "orange", "broccoli", "carrot", "hot dog", "pizza", "donut", "cake", "chair", "couch", "potted plant", "bed",
    "dining table", "toilet", "tv", "laptop", "mouse", "remote", "keyboard", "cell phone", "microwave", "oven",
    "toaster", "sink", "refrigerator", "book", "clock", "vase", "scissors", "teddy bear", "hair drier", "toothbrush",
]


# ---------------------------------------------------------------------------------------------------------------
# Datasets (datasets/coco.py:24-270): same raw-item design as datasets/voc.py of this build.
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from .transforms import draw_geometry, draw_train_views
from .voc import load_img_name_list, load_cls_label_list, _read_rgb, _read_label


class CocoDataset(Dataset):
    """datasets/coco.py:30-76: images under <img_dir>/{train2014,val2014}, labels under <label_dir>/{train2014,val2014};
    grey-scale JPEGs are replicated to three channels (robust_read_image, coco.py:24-28 == PIL convert("RGB"))."""

    def __init__(self, img_dir=None, label_dir=None, name_list_dir=None, split="train", stage="train"):
        super().__init__()
        self.img_dir, self.label_dir, self.stage = img_dir, label_dir, stage
        self.name_list_dir = os.path.join(name_list_dir, split + ".txt")
        self.name_list = load_img_name_list(self.name_list_dir)

    def __len__(self):
        return len(self.name_list)

    def _paths(self, name):
        sub = "train2014" if self.stage == "train" else "val2014"
        return os.path.join(self.img_dir, sub, name + ".jpg"), os.path.join(self.label_dir, sub, name + ".png")

    def __getitem__(self, idx):
        name = str(self.name_list[idx])
        ip, lp = self._paths(name)
        return name, _read_rgb(ip), _read_label(lp)


class CocoClsDataset(CocoDataset):
    """datasets/coco.py:79-199 with aug=True: raw train item `(img_name, raw uint8 (h,w,3), cls_label, Geometry)`."""

    def __init__(self, img_dir=None, label_dir=None, name_list_dir=None, split="train", stage="train",
                 resize_range=(512, 640), rescale_range=(0.5, 2.0), crop_size=512, img_fliplr=True, ignore_index=255,
                 num_classes=81, aug=False, **kwargs):
        super().__init__(img_dir, label_dir, name_list_dir, split, stage)
        self.aug, self.ignore_index = aug, ignore_index
        self.rescale_range, self.crop_size, self.img_fliplr = rescale_range, crop_size, img_fliplr
        self.num_classes = num_classes
        self.photometric = kwargs.get("photometric", True)    # global_view1's jitter / grayscale / blur (coco.py:117-132)
        self.label_list = load_cls_label_list(name_list_dir=name_list_dir)

    def __getitem__(self, idx):
        name = str(self.name_list[idx])
        image = _read_rgb(self._paths(name)[0])
        cls_label = self.label_list[name]
        raw = torch.from_numpy(np.ascontiguousarray(image))
        if not self.aug:
            return name, raw, cls_label
        geo = draw_geometry(image.shape[0], image.shape[1], self.rescale_range, self.crop_size, self.img_fliplr)
        geo.photometric = draw_train_views(image.shape[0], image.shape[1]) if self.photometric else None
        return name, raw, cls_label, geo


class CocoSegDataset(CocoDataset):
    """datasets/coco.py:202-270 (aug=False val split): raw item `(img_name, raw uint8, label, cls_label)`."""

    def __init__(self, img_dir=None, label_dir=None, name_list_dir=None, split="train", stage="train",
                 resize_range=(512, 640), rescale_range=(0.5, 2.0), crop_size=512, img_fliplr=True, ignore_index=255,
                 aug=False, **kwargs):
        super().__init__(img_dir, label_dir, name_list_dir, split, stage)
        if aug:
            raise NotImplementedError("the training scripts build CocoSegDataset with aug=False only")
        self.label_list = load_cls_label_list(name_list_dir=name_list_dir)

    def __getitem__(self, idx):
        name, image, label = super().__getitem__(idx)
        return name, torch.from_numpy(np.ascontiguousarray(image)), label, self.label_list[name]
