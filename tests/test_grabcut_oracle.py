"""CPU tests: oracle GrabCut — max-flow value and cut pinned against scipy's
maximum_flow on random 8-neighbour grids; GrabCut behaviour against the
reference's own test expectations (tests/test.py:31-82)."""
import numpy as np
import pytest
import scipy.sparse as sp
from scipy.sparse.csgraph import maximum_flow, breadth_first_order

PLANE_OFF = [(0, -1), (-1, -1), (-1, 0), (-1, 1)]   # left, up-left, up, up-right


def scipy_cut(tw, nw):
    h, w = tw.shape
    n = h * w
    s, t = n, n + 1
    rows, cols, caps = [], [], []
    for k, (dy, dx) in enumerate(PLANE_OFF):
        for y in range(h):
            for x in range(w):
                yy, xx = y + dy, x + dx
                if 0 <= yy < h and 0 <= xx < w and nw[k, y, x] > 0:
                    a, b = y * w + x, yy * w + xx
                    rows += [a, b]; cols += [b, a]; caps += [nw[k, y, x]] * 2
    for p, v in enumerate(tw.ravel()):
        if v > 0: rows.append(s); cols.append(p); caps.append(v)
        if v < 0: rows.append(p); cols.append(t); caps.append(-v)
    g = sp.csr_matrix((np.array(caps, np.int32), (rows, cols)), shape=(n + 2, n + 2))
    res = maximum_flow(g, s, t)
    resid = (g - res.flow).tocsr()
    resid.data = np.maximum(resid.data, 0); resid.eliminate_zeros()
    reach = breadth_first_order(resid.T.tocsr(), t, directed=True, return_predecessors=False)  # can reach the sink
    side = np.ones(n + 2, np.uint8); side[reach] = 0
    return int(res.flow_value), side[:n].reshape(h, w)


@pytest.mark.parametrize("h,w,seed", [(8, 8, 0), (12, 9, 1), (16, 16, 2), (24, 32, 3), (32, 32, 4)])
def test_maxflow_value_and_canonical_cut_match_scipy(oracle, h, w, seed):
    rng = np.random.default_rng(seed)
    tw = rng.integers(-60, 61, size=(h, w)).astype(np.int32)
    tw[rng.random((h, w)) < 0.3] = 0
    nw = rng.integers(0, 25, size=(4, h, w)).astype(np.int32)
    flow, side = oracle.grid_maxflow(tw, nw)
    # our tw cancels each pixel's two t-links, so the flow value excludes nothing here
    want_flow, want_side = scipy_cut(tw, nw)
    assert flow == want_flow
    assert np.array_equal(side, want_side)


def test_maxflow_trivial_cases(oracle):
    z = np.zeros((4, 5), np.int32)
    nw = np.ones((4, 4, 5), np.int32)
    flow, side = oracle.grid_maxflow(z, nw)
    assert flow == 0 and side.all()                  # nothing reaches the sink: everything is source side
    tw = z.copy(); tw[0, 0] = 10; tw[3, 4] = -3
    flow, side = oracle.grid_maxflow(tw, nw)
    assert flow == 3 and side.all()                  # the sink link saturates: the sink is unreachable
    tw[0, 0] = 2; tw[3, 4] = -30
    flow, side = oracle.grid_maxflow(tw, nw)
    assert flow == 2 and not side.any()


def _img(h=64, w=64, seed=42):
    return np.random.RandomState(seed).randint(20, 220, (h, w, 3), dtype=np.uint8)   # reference tests/test.py:17-19


def test_bbox_mode_returns_binary(oracle):
    img = _img(100, 100)
    binary, mask, bgd, fgd, rc = oracle.grabcut(img, None, n_iter=1, mode=1, rect=(10, 10, 80, 80))
    assert binary.shape == (100, 100) and set(np.unique(binary)) <= {0, 1}
    assert (mask[:10] == 0).all() and set(np.unique(mask[10:90, 10:90])) <= {2, 3}
    assert bgd.shape == (65,) and abs(bgd[:5].sum() - 1) < 1e-9 and abs(fgd[:5].sum() - 1) < 1e-9


def test_trimap_mode_with_only_probable_labels(oracle):
    # reference tests/test.py:41-49 — exercises the promotion branch grabcut.py:128-133
    img = _img(100, 100)
    tri = np.full((100, 100), 2, np.uint8); tri[30:70, 30:70] = 3
    binary, mask, *_ , rc = oracle.grabcut(img, tri, n_iter=1, mode=0)
    assert rc == 0 and binary.shape == (100, 100)
    assert (binary[30:70, 30:70] == 1).all() and binary[:30].sum() == 0   # promoted to definite labels


def test_degenerate_trimap_is_returned_as_is(oracle):
    img = _img(32, 32)
    tri = np.full((32, 32), 3, np.uint8)
    binary, mask, *_, rc = oracle.grabcut(img, tri, n_iter=5, mode=0)
    assert rc == 1 and binary.all() and (mask == 1).all()


def test_grabcut_separates_an_obvious_object(oracle):
    from gcn_grabcut.synthetic import synthetic_image
    img, gt = synthetic_image(96, 128, 11, return_mask=True)
    tri = np.full(gt.shape, 2, np.uint8)
    tri[gt == 1] = 3
    core = np.zeros_like(gt); core[8:-8, 8:-8] = 1
    tri[0:4] = 0; tri[-4:] = 0; tri[:, 0:4] = 0; tri[:, -4:] = 0
    ys, xs = np.nonzero(gt)
    tri[int(ys.mean()) - 2:int(ys.mean()) + 3, int(xs.mean()) - 2:int(xs.mean()) + 3] = 1
    binary, *_ = oracle.grabcut(img, tri, n_iter=5, mode=0, seed=3)
    assert oracle.iou(binary, gt) > 0.8


def test_eval_mode_reuses_models(oracle):
    img = _img(48, 48)
    tri = np.full((48, 48), 2, np.uint8); tri[12:36, 12:36] = 3; tri[0, 0] = 0; tri[24, 24] = 1
    b1, m1, bgd, fgd, _ = oracle.grabcut(img, tri, n_iter=2, mode=0, seed=1)
    b2, m2, bgd2, fgd2, _ = oracle.grabcut(img, m1, n_iter=1, mode=2, bgd=bgd, fgd=fgd)
    assert b2.shape == b1.shape and set(np.unique(m2)) <= {0, 1, 2, 3}
    b3, *_ = oracle.grabcut(img, tri, n_iter=3, mode=0, seed=1)
    assert np.array_equal(b2, b3)                    # 2 iterations + 1 continued == 3 iterations


def test_clean_mask_and_compose_and_iou(oracle):
    m = np.zeros((40, 50), np.uint8)
    m[5:25, 5:30] = 1          # 500 px
    m[30:32, 40:42] = 1        # 4 px, diagonal neighbour below keeps 8-connectivity
    m[32, 42] = 1
    out = oracle.clean_mask(m, 0.002, False)          # min area 4.0: the 5-px blob survives
    assert out.sum() == 505
    out = oracle.clean_mask(m, 0.01, False)           # min area 20
    assert out.sum() == 500
    out = oracle.clean_mask(m, 0.9, False)            # nothing survives -> largest kept
    assert out.sum() == 500
    assert oracle.clean_mask(m, 0.0, True).sum() == 500
    assert np.array_equal(oracle.clean_mask(m, 0.0, False), m)
    assert np.array_equal(oracle.clean_mask(np.zeros_like(m), 0.002, False), np.zeros_like(m))
    img = _img(40, 50)
    ov, rgba = oracle.compose(img, m)
    want = np.clip(img.astype(np.float32) * (1 - 0.45 * m[..., None].astype(np.float32))
                   + np.array([100, 220, 0], np.float32) * 0.45 * m[..., None].astype(np.float32), 0, 255).astype(np.uint8)
    assert np.array_equal(ov, want)
    assert np.array_equal(rgba[..., :3], img) and np.array_equal(rgba[..., 3], m * 255)
    assert oracle.iou(m, m) == pytest.approx(1.0, abs=1e-4) and oracle.iou(np.zeros_like(m), m) < 0.01
