"""
Segmentation metrics — the part of reference src/gcn_grabcut/metrics.py that the
hot path reports (IoU, metrics.py:79-84) plus the ratios that follow from the
same confusion counts.  The counts come from ggc_mask_iou on the MI355X.
Boundary F1 and the trimap metrics are evaluation extras outside the hot path
(component 6).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class SegmentationMetrics:
    iou: float
    dice: float
    precision: float
    recall: float
    f1: float
    pixel_accuracy: float
    boundary_f1: float = 0.0

    def __str__(self) -> str:
        return (f"IoU={self.iou:.4f}  Dice={self.dice:.4f}  Prec={self.precision:.4f}  Rec={self.recall:.4f}  "
                f"F1={self.f1:.4f}  PixAcc={self.pixel_accuracy:.4f}  BF1={self.boundary_f1:.4f}")

    def as_dict(self) -> dict:
        return {k: round(getattr(self, k), 4) for k in
                ("iou", "dice", "precision", "recall", "f1", "pixel_accuracy", "boundary_f1")}


def evaluate(pred: np.ndarray, gt: np.ndarray, boundary_width: int = 0, device="cuda") -> SegmentationMetrics:
    """Binary-mask metrics (reference metrics.py:58-102); boundary F1 is not computed here."""
    from ._engine import get_engine
    if pred.shape != gt.shape:
        raise ValueError(f"shape mismatch {pred.shape} vs {gt.shape}")
    eng = get_engine(device)
    p = eng.to_device((np.asarray(pred) != 0).astype(np.uint8)[None])
    g = eng.to_device((np.asarray(gt) != 0).astype(np.uint8)[None])
    iou, cnt = eng.iou(p, g)
    tp, fp, fn = (int(v) for v in cnt[0].cpu().tolist())
    tn = pred.size - tp - fp - fn
    precision = tp / (tp + fp + 1e-8)
    recall = tp / (tp + fn + 1e-8)
    return SegmentationMetrics(
        iou=float(iou[0].item()), dice=float(2 * tp / (2 * tp + fp + fn + 1e-8)),
        precision=float(precision), recall=float(recall),
        f1=float(2 * precision * recall / (precision + recall + 1e-8)),
        pixel_accuracy=float((tp + tn) / (tp + tn + fp + fn + 1e-8)), boundary_f1=0.0)
