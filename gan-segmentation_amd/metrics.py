"""pixAcc / mIoU bookkeeping of the reference's ``SegmentationMetric`` (metrics.py:497-606) on top of a
confusion matrix (``gsa_segmentation_eval`` produces it on the device).

confusion[l, p] = number of pixels with label l >= 0 (ignored pixels never counted) and predicted class p.
The reference's quantities follow exactly: ``batch_pix_accuracy`` -> correct = trace, labelled = sum;
``batch_intersection_union`` -> inter = diagonal, area_pred = column sums (its ``predict`` is zeroed where the
target is ignored), area_lab = row sums, union = pred + lab - inter.
"""
import numpy as np


class SegmentationMetric:
    def __init__(self, nclass, skip_bg=True):
        self.nclass = nclass
        self._skip_bg = skip_bg
        self.reset()

    def reset(self):
        self.total_inter = np.zeros(self.nclass, np.int64)
        self.total_union = np.zeros(self.nclass, np.int64)
        self.total_correct = 0
        self.total_label = 0

    def update_confusion(self, confusion):
        c = np.asarray(confusion).astype(np.int64).reshape(self.nclass, self.nclass)
        inter = np.diag(c)
        self.total_inter += inter
        self.total_union += c.sum(axis=0) + c.sum(axis=1) - inter
        self.total_correct += int(inter.sum())
        self.total_label += int(c.sum())

    def get(self):
        """-> (['accuracy', 'mean-iou'], [pixAcc, mIoU]), the reference's formulas (metrics.py:541-561)."""
        pixAcc = 1.0 * self.total_correct / (np.spacing(1) + self.total_label)
        IoU = 1.0 * self.total_inter / (np.spacing(1) + self.total_union)
        IoU = IoU[self.total_union > 0]
        if self._skip_bg:
            IoU = IoU[1:]      # reference: skip background class (after dropping empty classes, as it does)
        mIoU = IoU.mean() if IoU.size else float("nan")
        return ["accuracy", "mean-iou"], [pixAcc, mIoU]

    def get_name_value(self):
        names, values = self.get()
        return list(zip(names, values))
