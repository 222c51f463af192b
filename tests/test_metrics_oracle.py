"""Oracle evaluation tallies pinned against numpy / scipy restatements of reference metrics.py:58-201
(cv2.erode's default border = pixels outside the image never erode = binary_erosion(border_value=1))."""
import numpy as np
from scipy import ndimage


def _masks(seed, h=70, w=93):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[:h, :w]
    gt = (((yy - h * 0.5) / (h * 0.3)) ** 2 + ((xx - w * 0.45) / (w * 0.28)) ** 2 < 1).astype(np.uint8)
    pred = np.roll(gt, (2, -3), (0, 1)) | (rng.random((h, w)) < 0.01).astype(np.uint8)
    pred[:, :2] = 1                                         # touches the border: exercises the erode border rule
    tri = rng.integers(0, 4, (h, w)).astype(np.uint8)
    return pred, gt, tri


def test_eval_counts_match_numpy_and_scipy(oracle):
    for seed, width in ((0, 3), (1, 1), (2, 5)):
        pred, gt, tri = _masks(seed)
        got = oracle.eval_counts(pred, gt, tri, width)
        p, g = pred.astype(bool), gt.astype(bool)
        assert list(got[:3]) == [(p & g).sum(), (p & ~g).sum(), (~p & g).sum()]
        k = np.ones((2 * width + 1,) * 2, bool)
        pb = p & ~ndimage.binary_erosion(p, k, border_value=1)
        gb = g & ~ndimage.binary_erosion(g, k, border_value=1)
        assert list(got[3:6]) == [pb.sum(), gb.sum(), (pb & gb).sum()]
        pf, pbg = tri == 1, tri == 0
        want = [(pf & g).sum(), (pf & ~g).sum(), (~pf & g).sum(), (pbg & ~g).sum(), (pbg & g).sum(), (~pbg & ~g).sum(),
                ((tri == 2) | (tri == 3)).sum(), (((tri == 1) | (tri == 3)).astype(np.uint8) == gt).sum()]
        assert list(got[6:]) == want
    pred, gt, _ = _masks(3)
    got = oracle.eval_counts(pred, gt, None, 0)
    assert got[3:].sum() == 0 and got[0] > 0
