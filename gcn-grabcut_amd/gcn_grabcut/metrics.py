"""
Segmentation metrics — host mirror of reference src/gcn_grabcut/metrics.py: `evaluate`
(:58-102), `boundary_f1` (:105-129), `evaluate_trimap` (:152-201), `evaluate_batch`
(:204-229).  Every metric is a ratio of integer tallies; the tallies (confusion counts,
eroded-boundary overlaps, trimap confusion) come from `ggc_eval_counts` on the MI355X,
the ratios use the reference's formulas (+1e-8 denominators) on the host.
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


def _counts(pred: np.ndarray, gt: np.ndarray, trimap=None, boundary_width: int = 0, device="cuda",
            binarize: bool = True) -> np.ndarray:
    """int64 [B,14] tallies of ggc_eval_counts for (B,H,W) or (H,W) inputs; binarize: masks as `!= 0` (astype(bool))."""
    import torch
    from ._engine import get_engine
    pred, gt = np.asarray(pred), np.asarray(gt)
    if pred.shape != gt.shape:
        raise ValueError(f"shape mismatch {pred.shape} vs {gt.shape}")
    if pred.ndim == 2:
        pred, gt = pred[None], gt[None]
        trimap = None if trimap is None else np.asarray(trimap)[None]
    eng = get_engine(device)
    b, h, w = pred.shape
    if binarize:
        pred, gt = pred != 0, gt != 0
    p = eng.to_device(np.ascontiguousarray(pred).astype(np.uint8))
    g = eng.to_device(np.ascontiguousarray(gt).astype(np.uint8))
    t = None if trimap is None else eng.to_device(np.ascontiguousarray(trimap, dtype=np.uint8))
    out = eng.empty(b, 14, dtype=torch.int64)
    eng.ctx.call("ggc_eval_counts", eng._stream(), b, h, w, p.data_ptr(), g.data_ptr(), None if t is None else t.data_ptr(),
                 int(boundary_width), out.data_ptr())
    return out.cpu().numpy()


def _bf1(n_pred: int, n_gt: int, n_both: int) -> float:
    prec = float(n_both / (n_pred + 1e-8))
    rec = float(n_both / (n_gt + 1e-8))
    return float(2 * prec * rec / (prec + rec + 1e-8))


def _from_counts(c: np.ndarray, n_pixels: int, with_boundary: bool) -> SegmentationMetrics:
    tp, fp, fn = int(c[0]), int(c[1]), int(c[2])
    tn = n_pixels - tp - fp - fn
    precision = float(tp / (tp + fp + 1e-8))
    recall = float(tp / (tp + fn + 1e-8))
    return SegmentationMetrics(
        iou=float(tp / (tp + fp + fn + 1e-8)), dice=float(2 * tp / (2 * tp + fp + fn + 1e-8)),
        precision=precision, recall=recall, f1=float(2 * precision * recall / (precision + recall + 1e-8)),
        pixel_accuracy=float((tp + tn) / (tp + tn + fp + fn + 1e-8)),
        boundary_f1=_bf1(int(c[3]), int(c[4]), int(c[5])) if with_boundary else 0.0)


def evaluate(pred: np.ndarray, gt: np.ndarray, boundary_width: int = 3, device="cuda") -> SegmentationMetrics:
    """Binary-mask metrics of one (H, W) pair — reference metrics.py:58-102."""
    c = _counts(pred, gt, None, boundary_width, device)[0]
    return _from_counts(c, int(np.asarray(pred).size), boundary_width > 0)


def boundary_f1(pred_2d: np.ndarray, gt_2d: np.ndarray, width: int = 3, device="cuda") -> float:
    """Alignment of predicted and GT boundaries, boundary = m - erode(m, ones(2*width+1)^2) — reference metrics.py:105-129."""
    c = _counts(pred_2d, gt_2d, None, width, device)[0]
    return _bf1(int(c[3]), int(c[4]), int(c[5]))


@dataclass
class TrimapMetrics:
    fg_recall: float
    fg_precision: float
    bg_recall: float
    bg_precision: float
    bg_contamination: float   # FG-labelled pixels that are actually BG
    unknown_fraction: float
    trimap_accuracy: float    # how much of the trimap matches the GT

    def __str__(self) -> str:
        return (f"FG_rec={self.fg_recall:.3f}  FG_prec={self.fg_precision:.3f}  BG_rec={self.bg_recall:.3f}  "
                f"BG_cont={self.bg_contamination:.3f}  Unk={self.unknown_fraction:.3f}  Acc={self.trimap_accuracy:.3f}")

    def as_dict(self) -> dict:
        return {k: round(v, 4) for k, v in self.__dict__.items()}


def evaluate_trimap(trimap: np.ndarray, gt_mask: np.ndarray, device="cuda") -> TrimapMetrics:
    """Predicted trimap {0 BG, 1 FG, 2 PROB_BG, 3 PROB_FG} against a binary GT mask — reference metrics.py:152-201.
    (gt_mask holds {0, 1}, as in the reference, whose accuracy term compares the raw mask values.)"""
    gt = np.asarray(gt_mask)
    c = _counts((np.asarray(trimap) == 1) | (np.asarray(trimap) == 3), gt, trimap, 0, device, binarize=False)[0]
    n = gt.size
    fg_tp, fg_fp, fg_fn, bg_tp, bg_fp, bg_fn, n_prob, n_match = (int(v) for v in c[6:14])
    return TrimapMetrics(
        fg_recall=float(fg_tp / (fg_tp + fg_fn + 1e-8)), fg_precision=float(fg_tp / (fg_tp + fg_fp + 1e-8)),
        bg_recall=float(bg_tp / (bg_tp + bg_fn + 1e-8)), bg_precision=float(bg_tp / (bg_tp + bg_fp + 1e-8)),
        bg_contamination=float(fg_fp / n), unknown_fraction=float(n_prob / n), trimap_accuracy=float(n_match / n))


def evaluate_batch(results: list[dict], device="cuda") -> dict:
    """Mean and std of IoU / Dice / BF1 over result dicts with "binary_mask" and "gt_mask" — reference metrics.py:204-229.
    Equally sized pairs are tallied in one device call."""
    all_iou, all_dice, all_bf1 = [], [], []
    by_shape: dict[tuple, list[int]] = {}
    for i, r in enumerate(results):
        by_shape.setdefault(np.asarray(r["binary_mask"]).shape, []).append(i)
    metrics: dict[int, SegmentationMetrics] = {}
    for shape, idx in by_shape.items():
        pred = np.stack([np.asarray(results[i]["binary_mask"]) for i in idx])
        gt = np.stack([np.asarray(results[i]["gt_mask"]) for i in idx])
        c = _counts(pred, gt, None, 3, device)
        for j, i in enumerate(idx):
            metrics[i] = _from_counts(c[j], int(np.prod(shape)), True)
    for i in range(len(results)):
        all_iou.append(metrics[i].iou); all_dice.append(metrics[i].dice); all_bf1.append(metrics[i].boundary_f1)
    return {"mean_iou": float(np.mean(all_iou)), "std_iou": float(np.std(all_iou)),
            "mean_dice": float(np.mean(all_dice)), "std_dice": float(np.std(all_dice)),
            "mean_bf1": float(np.mean(all_bf1)), "std_bf1": float(np.std(all_bf1)), "n": len(results)}
