"""PASCAL VOC 2012 category names, index = label id (datasets/voc.py:14)."""
class_list = ["bg", "aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "table", "dog",
              "horse", "motorbike", "person", "plant", "sheep", "sofa", "train", "tvmonitor"]
