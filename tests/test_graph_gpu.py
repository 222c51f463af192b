"""GPU parity for G2-G8 (graph construction) through the C ABI: edge_index integer-exact AND every float output identical
to the CPU oracle's (region sums are exact integer / raster-order double sums; the prior's float sums follow one fixed
order and the shared exp of include/ggc_fmath.h on both sides)."""
import numpy as np
import pytest
import torch

import gpu_helpers as gh

pytestmark = pytest.mark.gpu


def _check_image(oracle, g, i, seg_h, lab_h, hsv_h, grad_h, conn, k):
    want = oracle.graph_build(seg_h, lab_h, hsv_h, grad_h, connectivity=conn, n_nonlocal=k)
    n0, n1 = g["node_ptr"][i], g["node_ptr"][i + 1]
    e0, e1 = g["edge_ptr"][i], g["edge_ptr"][i + 1]
    assert n1 - n0 == want["n_nodes"] and e1 - e0 == want["n_edges"]
    x = g["x"][n0:n1].cpu().numpy()
    assert np.array_equal(x[:, :16], want["node_features"])
    assert np.array_equal(x[:, 16:], want["prior"])
    assert np.array_equal(g["centroids"][n0:n1].cpu().numpy(), want["centroids"])
    assert np.array_equal(g["area"][n0:n1].cpu().numpy(), want["area_ratio"])
    ei = np.stack([g["src"][e0:e1].cpu().numpy(), g["dst"][e0:e1].cpu().numpy()]).astype(np.int64)
    assert np.array_equal(ei, want["edge_index"])                      # integer-exact, reference order
    if e1 > e0:
        assert np.array_equal(g["attr"][e0:e1].cpu().numpy(), want["edge_attr"])
    return want


@pytest.mark.parametrize("h,w,b,n_seg,conn,k", [(64, 64, 2, 50, 4, 4), (72, 96, 3, 120, 8, 4), (96, 128, 2, 200, 4, 0),
                                                (300, 400, 2, 600, 4, 4)])
def test_graph_matches_oracle(oracle, gpu_ctx, h, w, b, n_seg, conn, k):
    from gcn_grabcut.synthetic import synthetic_batch
    bgr = synthetic_batch(b, h, w, config_id=4)
    _, lab, hsv, gray, grad = gh.preprocess(gpu_ctx, bgr)
    seg, nn = gh.slic(gpu_ctx, lab, n_seg)
    g = gh.graph(gpu_ctx, seg, nn, lab, hsv, grad, conn, k)
    seg_h, lab_h, hsv_h, grad_h = seg.cpu().numpy(), lab.cpu().numpy(), hsv.cpu().numpy(), grad.cpu().numpy()
    for i in range(b):
        want = _check_image(oracle, g, i, seg_h[i], lab_h[i], hsv_h[i], grad_h[i], conn, k)
        if (h, w) == (300, 400):
            assert 5000 <= want["n_edges"] <= 8000       # E ~ 6.46k at N ~ 600


def test_graph_ragged_batch_of_handmade_segmentations(oracle, gpu_ctx):
    """Images of one batch with very different node counts, incl. a single-region
    image (N=1, E=0) and a two-region image (no non-local edges: N <= k+1)."""
    from gcn_grabcut.synthetic import synthetic_batch
    h, w = 48, 64
    bgr = synthetic_batch(3, h, w, config_id=5)
    _, lab, hsv, gray, grad = gh.preprocess(gpu_ctx, bgr)
    seg = np.zeros((3, h, w), np.int32)
    seg[1, :, 32:] = 1
    yy, xx = np.mgrid[0:h, 0:w]
    seg[2] = (yy // 8) * 8 + xx // 8
    nn = np.array([1, 2, 48], np.int32)
    seg_d, nn_d = torch.as_tensor(seg).cuda(), torch.as_tensor(nn).cuda()
    g = gh.graph(gpu_ctx, seg_d, nn_d, lab, hsv, grad, 4, 4)
    assert list(np.diff(g["node_ptr"])) == [1, 2, 48]
    lab_h, hsv_h, grad_h = lab.cpu().numpy(), hsv.cpu().numpy(), grad.cpu().numpy()
    for i in range(3):
        _check_image(oracle, g, i, seg[i], lab_h[i], hsv_h[i], grad_h[i], 4, 4)
    assert np.diff(g["edge_ptr"])[0] == 0 and np.diff(g["edge_ptr"])[1] == 2


def test_graph_fill_requires_count(gpu_ctx):
    from gcn_grabcut import _native
    fresh = _native.Context(0)
    with pytest.raises(_native.GGCError, match="GGC_E_STATE"):
        fresh.call("ggc_graph_fill", 0, None, None, None, None, None, None, 0)
    fresh.close()
