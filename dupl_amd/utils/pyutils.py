"""Small host utilities of the training scripts (reference: utils/pyutils.py): AverageMeter, format_tabs, cal_eta.
`format_tabs` draws the same rows/columns as the reference's Texttable output with plain string formatting
(texttable is not a dependency here)."""
import datetime

import numpy as np


class AverageMeter:
    """pyutils.py:59-86."""

    def __init__(self, *keys):
        self._data = {k: [0.0, 0] for k in keys}

    def add(self, d):
        for k, v in d.items():
            if k not in self._data:
                self._data[k] = [0.0, 0]
            self._data[k][0] += v
            self._data[k][1] += 1

    def get(self, *keys):
        if len(keys) == 1:
            return self._data[keys[0]][0] / self._data[keys[0]][1]
        return tuple(self._data[k][0] / self._data[k][1] for k in keys)

    def pop(self, key=None):
        if key is None:
            for k in self._data:
                self._data[k] = [0.0, 0]
            return None
        v = self.get(key)
        self._data[key] = [0.0, 0]
        return v


def format_tabs(scores, name_list, cat_list=None, return_item=False):
    """pyutils.py:7-27: per-class IoU (x100) of every score dict + an mIoU row (plain mean over all classes, nan
    included, exactly as the reference computes `_values.mean(1)`)."""
    keys = list(scores[0]["iou"].keys())
    values = np.array([list(s["iou"].values()) for s in scores]) * 100
    width = max([len("Class")] + [len(str(cat_list[i])) for i in range(len(keys))]) + 2
    head = "Class".ljust(width) + "".join(n.rjust(12) for n in name_list)
    lines = [head, "-" * len(head)]
    for i in range(len(keys)):
        lines.append(str(cat_list[i]).ljust(width) + "".join(f"{v:12.3f}" for v in values[:, i]))
    lines.append("-" * len(head))
    lines.append("mIoU".ljust(width) + "".join(f"{v:12.3f}" for v in values.mean(1)))
    table = "\n".join(lines)
    if return_item:
        return table, list(values.mean(1))
    return table


def cal_eta(time0, cur_iter, total_iter):
    """pyutils.py:46-56 -> (elapsed, eta) strings."""
    time_now = datetime.datetime.now().replace(microsecond=0)
    scale = (total_iter - cur_iter) / float(cur_iter)
    delta = time_now - time0
    time_fin = time_now + delta * scale
    eta = time_fin.replace(microsecond=0) - time_now
    return str(delta), str(eta)
