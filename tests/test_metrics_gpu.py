"""GPU evaluation tallies (ggc_eval_counts) against the oracle, and the metric values against the reference's
formulas (metrics.py:58-229) evaluated in numpy/scipy."""
import numpy as np
import pytest
from scipy import ndimage

from test_metrics_oracle import _masks

pytestmark = pytest.mark.gpu


def test_eval_counts_bit_exact(oracle, gpu_ctx):
    from gcn_grabcut import metrics as M
    for seed, width in ((0, 3), (1, 1), (2, 5), (3, 0)):
        pred, gt, tri = _masks(seed, 75, 131)
        got = M._counts(pred, gt, tri, width, binarize=False)[0]
        assert np.array_equal(got, oracle.eval_counts(pred, gt, tri, width)), (seed, width)
    batch_p = np.stack([_masks(s)[0] for s in range(4)]); batch_g = np.stack([_masks(s)[1] for s in range(4)])
    got = M._counts(batch_p, batch_g, None, 3)
    for i in range(4):
        assert np.array_equal(got[i], oracle.eval_counts(batch_p[i], batch_g[i], None, 3))


def test_metrics_match_reference_formulas(gpu_ctx):
    from gcn_grabcut import evaluate, evaluate_batch, evaluate_trimap, boundary_f1
    pred, gt, tri = _masks(7, 90, 120)
    m = evaluate(pred * 255, gt)                            # any non-zero value counts, like astype(bool)
    p, g = pred.astype(bool).ravel(), gt.astype(bool).ravel()
    tp, fp, fn, tn = (p & g).sum(), (p & ~g).sum(), (~p & g).sum(), (~p & ~g).sum()
    prec, rec = tp / (tp + fp + 1e-8), tp / (tp + fn + 1e-8)
    assert m.iou == float(tp / (tp + fp + fn + 1e-8)) and m.dice == float(2 * tp / (2 * tp + fp + fn + 1e-8))
    assert m.precision == float(prec) and m.recall == float(rec) and m.f1 == float(2 * prec * rec / (prec + rec + 1e-8))
    assert m.pixel_accuracy == float((tp + tn) / (tp + tn + fp + fn + 1e-8))
    k = np.ones((7, 7), bool)
    pb = pred.astype(bool) & ~ndimage.binary_erosion(pred.astype(bool), k, border_value=1)
    gb = gt.astype(bool) & ~ndimage.binary_erosion(gt.astype(bool), k, border_value=1)
    btp = (pb & gb).sum(); bp, br = btp / (pb.sum() + 1e-8), btp / (gb.sum() + 1e-8)
    assert m.boundary_f1 == float(2 * bp * br / (bp + br + 1e-8)) == boundary_f1(pred, gt, 3)
    assert evaluate(pred, gt, boundary_width=0).boundary_f1 == 0.0
    t = evaluate_trimap(tri, gt)
    pf, pbg, gg = tri == 1, tri == 0, gt.astype(bool)
    assert t.fg_recall == float((pf & gg).sum() / ((pf & gg).sum() + (~pf & gg).sum() + 1e-8))
    assert t.bg_precision == float((pbg & ~gg).sum() / ((pbg & ~gg).sum() + (pbg & gg).sum() + 1e-8))
    assert t.bg_contamination == float((pf & ~gg).sum() / gt.size)
    assert t.unknown_fraction == float(((tri == 2) | (tri == 3)).sum() / gt.size)
    assert t.trimap_accuracy == float(((pf | (tri == 3)).astype(np.uint8).ravel() == gt.ravel()).mean())
    res = [{"binary_mask": _masks(s)[0], "gt_mask": _masks(s)[1]} for s in range(3)] + [{"binary_mask": pred, "gt_mask": gt}]
    agg = evaluate_batch(res)
    ious = [evaluate(r["binary_mask"], r["gt_mask"]).iou for r in res]
    assert agg["n"] == 4 and agg["mean_iou"] == float(np.mean(ious)) and agg["std_iou"] == float(np.std(ious))
    assert set(agg) == {"mean_iou", "std_iou", "mean_dice", "std_dice", "mean_bf1", "std_bf1", "n"}
