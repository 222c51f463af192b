"""CPU tests: oracle/trimap.c — box filter vs scipy's uniform_filter (mirror ==
BORDER_REFLECT_101), guided-filter properties, trimap decisions, seeding."""
import numpy as np
import pytest
from scipy import ndimage as ndi


def test_box_blur_matches_scipy_uniform_filter(oracle):
    rng = np.random.default_rng(0)
    img = rng.random((37, 53)).astype(np.float32)
    for r in (0, 1, 4, 8):
        got = oracle.box_blur(img, r)
        want = ndi.uniform_filter(img.astype(np.float64), size=2 * r + 1, mode="mirror")
        assert np.abs(got - want).max() <= 1e-6


def test_box_blur_radius_larger_than_image(oracle):
    img = np.arange(15, dtype=np.float32).reshape(3, 5)
    got = oracle.box_blur(img, 8)
    assert np.isfinite(got).all() and got.min() >= 0 and got.max() <= 14


def test_guided_filter_constant_source_is_identity(oracle):
    rng = np.random.default_rng(1)
    guide = rng.random((40, 48)).astype(np.float32)
    out = oracle.guided_filter(guide, np.full_like(guide, 0.7), 8, 1e-3)
    assert np.abs(out - 0.7).max() < 1e-5


def test_guided_filter_preserves_guide_edges(oracle):
    guide = np.zeros((48, 64), np.float32); guide[:, 32:] = 1.0
    src = np.zeros_like(guide); src[:, 28:] = 1.0           # source edge is 4 px off the guide edge
    out = oracle.guided_filter(guide, src, 8, 1e-3)
    assert out[:, 32].mean() > 0.95 and (out[:, 32] - out[:, 31]).mean() > 0.4   # the jump sits on the guide edge


def test_refine_trimap_labels_and_precedence(oracle):
    seg = np.zeros((32, 32), np.int32); seg[:, 16:] = 1
    bgr = np.full((32, 32, 3), 128, np.uint8)
    probs = np.array([[0.9, 0.05, 0.05], [0.05, 0.05, 0.9]], np.float32)
    tri = oracle.refine_trimap(probs, seg, bgr)
    assert set(np.unique(tri)) <= {0, 1, 2, 3}
    assert (tri[:, :4] == 0).all() and (tri[:, -4:] == 1).all()
    both = np.array([[0.6, 0.0, 0.6]], np.float32)               # both clear the threshold: FG wins
    assert (oracle.refine_trimap(both, np.zeros((8, 8), np.int32), bgr[:8, :8], edge_aware=False) == 1).all()
    unsure = np.array([[0.3, 0.3, 0.4], [0.4, 0.3, 0.3]], np.float32)
    t = oracle.refine_trimap(unsure, seg, bgr, edge_aware=False)
    assert (t[:, :16] == 3).all() and (t[:, 16:] == 2).all()
    # fewer probability rows than regions: zero / PR_BGD padding (model.py:655-660, 672-677)
    t = oracle.refine_trimap(unsure[:1], seg, bgr, edge_aware=False)
    assert (t[:, 16:] == 2).all()


def test_seed_from_prior(oracle):
    seg = (np.arange(64).reshape(8, 8) // 8).astype(np.int32)     # 8 regions = rows
    prior = np.zeros((8, 3), np.float32)
    prior[:, 0] = [0.1, 0.9, 0.3, 0.9, 0.2, 0.0, 0.5, 0.4]
    prior[:, 1] = [0.8, 0.1, 0.2, 0.3, 0.9, 0.7, 0.1, 0.0]
    tri = np.full((8, 8), 2, np.uint8)                             # no foreground at all
    out = oracle.seed_from_prior(tri, prior, seg, 0.25)            # round(2.0) = 2 seeds
    assert (out[1] == 3).all() and (out[3] == 3).all() and (out == 3).sum() == 16
    out = oracle.seed_from_prior(tri, prior, seg, 0.1)             # max(1, round(0.8)) = 1; tie -> larger index
    assert (out[3] == 3).all() and (out == 3).sum() == 8
    tri = np.full((8, 8), 1, np.uint8)                             # no background
    out = oracle.seed_from_prior(tri, prior, seg, 0.25)
    assert (out[4] == 2).all() and (out[0] == 2).all() and (out == 2).sum() == 16
    mixed = tri.copy(); mixed[0, 0] = 0
    assert np.array_equal(oracle.seed_from_prior(mixed, prior, seg, 0.25), mixed)
