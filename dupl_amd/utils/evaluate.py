"""Segmentation / classification scores (reference: utils/evaluate.py).

Same functions and return values as the reference (`multilabel_score`, `_fast_hist`, `scores`, `pseudo_scores`) for
host arrays, plus `ConfusionMatrix`: the histogram the reference rebuilds on the host from lists of per-image numpy maps
(`scores(gts, preds)` at the end of validation) is accumulated on the device instead, image by image
(csrc/eval.hip::confusion_kernel); only nc*nc counters are copied back."""
import numpy as np
import torch

from .. import ops


def multilabel_score(y_true, y_pred):
    """evaluate.py:4-6 = sklearn.metrics.f1_score of one multi-hot vector: 2TP / (2TP + FP + FN), 0 when undefined."""
    y_true, y_pred = np.asarray(y_true), np.asarray(y_pred)
    tp = float(((y_true == 1) & (y_pred == 1)).sum())
    fp = float(((y_true == 0) & (y_pred == 1)).sum())
    fn = float(((y_true == 1) & (y_pred == 0)).sum())
    den = 2 * tp + fp + fn
    return 2 * tp / den if den > 0 else 0.0


def _fast_hist(label_true, label_pred, num_classes):
    """evaluate.py:9-16."""
    mask = (label_true >= 0) & (label_true < num_classes)
    hist = np.bincount(num_classes * label_true[mask].astype(int) + label_pred[mask], minlength=num_classes ** 2)
    return hist.reshape(num_classes, num_classes)


def scores_from_hist(hist):
    """evaluate.py:22-36: {"pAcc", "mAcc", "miou" (over classes present in the ground truth), "iou": {class: IoU}}."""
    hist = np.asarray(hist, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        acc = np.diag(hist).sum() / hist.sum()
        acc_cls = np.nanmean(np.diag(hist) / hist.sum(axis=1))
        iu = np.diag(hist) / (hist.sum(axis=1) + hist.sum(axis=0) - np.diag(hist))
    valid = hist.sum(axis=1) > 0
    return {"pAcc": acc, "mAcc": acc_cls, "miou": np.nanmean(iu[valid]), "iou": dict(zip(range(hist.shape[0]), iu))}


def scores(label_trues, label_preds, num_classes=21):
    """evaluate.py:18-36 for lists of host arrays."""
    hist = np.zeros((num_classes, num_classes))
    for lt, lp in zip(label_trues, label_preds):
        hist += _fast_hist(np.asarray(lt).flatten(), np.asarray(lp).flatten(), num_classes)
    return scores_from_hist(hist)


def pseudo_scores(label_trues, label_preds, num_classes=21):
    """evaluate.py:38-62: predictions equal to 255 are excluded (truth set to 255, prediction to 0)."""
    hist = np.zeros((num_classes, num_classes))
    for lt, lp in zip(label_trues, label_preds):
        lt, lp = np.array(lt).flatten(), np.array(lp).flatten()
        lt[lp == 255] = 255
        lp[lp == 255] = 0
        hist += _fast_hist(lt, lp, num_classes)
    return scores_from_hist(hist)


class ConfusionMatrix:
    """Device-resident (nc, nc) int64 histogram; update(gt, pred) with int64 device tensors of equal shape."""

    def __init__(self, num_classes, device):
        self.num_classes = int(num_classes)
        self.hist = torch.zeros((self.num_classes, self.num_classes), device=device, dtype=torch.int64)

    def update(self, gt, pred):
        ops.confusion_accum(gt, pred, self.hist)

    def scores(self):
        return scores_from_hist(self.hist.cpu().numpy())
