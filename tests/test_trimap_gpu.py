"""GPU parity for P0-P3 / S0: the trimap is bit-exact against the CPU oracle."""
import numpy as np
import pytest
import torch

import gpu_helpers as gh

pytestmark = pytest.mark.gpu


def _refine(ctx, probs, node_ptr, seg, bgr, thr=0.55, radius=8, eps=1e-3, edge_aware=1):
    b, h, w = seg.shape
    tri = torch.empty(b, h, w, dtype=torch.uint8, device="cuda")
    ctx.call("ggc_refine_trimap", gh.stream(), b, h, w, probs.data_ptr(), node_ptr.data_ptr(), seg.data_ptr(),
             bgr.data_ptr(), thr, thr, radius, eps, edge_aware, tri.data_ptr())
    return tri


@pytest.mark.parametrize("h,w,b,n_seg,radius", [(64, 64, 2, 50, 8), (50, 81, 3, 80, 4), (300, 400, 2, 600, 8), (20, 24, 1, 12, 8),
                                                  (33, 70, 1, 20, 1), (70, 129, 2, 40, 3), (40, 56, 2, 30, 12)])
def test_refine_trimap_bit_exact(oracle, gpu_ctx, h, w, b, n_seg, radius):
    from gcn_grabcut.synthetic import synthetic_batch
    bgr_h = synthetic_batch(b, h, w, config_id=6)
    bgr, lab, hsv, gray, grad = gh.preprocess(gpu_ctx, bgr_h)
    seg, nn = gh.slic(gpu_ctx, lab, n_seg)
    nn_h, seg_h = nn.cpu().numpy(), seg.cpu().numpy()
    node_ptr = np.concatenate([[0], np.cumsum(nn_h)]).astype(np.int32)
    rng = np.random.default_rng(h * w)
    logits = rng.standard_normal((node_ptr[-1], 3)).astype(np.float32) * 2
    probs = np.exp(logits) / np.exp(logits).sum(1, keepdims=True)
    probs = probs.astype(np.float32)
    pd, npd = torch.as_tensor(probs).cuda(), torch.as_tensor(node_ptr).cuda()
    for edge_aware in (1, 0):
        tri = _refine(gpu_ctx, pd, npd, seg, bgr, 0.55, radius, 1e-3, edge_aware).cpu().numpy()
        for i in range(b):
            want = oracle.refine_trimap(probs[node_ptr[i]:node_ptr[i + 1]], seg_h[i], bgr_h[i], 0.55, 0.55, radius,
                                        1e-3, bool(edge_aware))
            assert np.array_equal(tri[i], want), (edge_aware, i, int((tri[i] != want).sum()))
        assert len(np.unique(tri)) >= 3            # the random probabilities exercise several labels


def test_refine_trimap_fewer_probability_rows_than_regions(oracle, gpu_ctx):
    seg = np.zeros((1, 16, 16), np.int32); seg[0, :, 8:] = 1; seg[0, 8:, 8:] = 2
    bgr = np.full((1, 16, 16, 3), 90, np.uint8)
    probs = np.array([[0.2, 0.2, 0.6], [0.7, 0.2, 0.1]], np.float32)       # region 2 has no row
    node_ptr = np.array([0, 2], np.int32)
    args = [torch.as_tensor(a).cuda() for a in (probs, node_ptr, seg, bgr)]
    for ea in (1, 0):
        got = _refine(gpu_ctx, *args, edge_aware=ea).cpu().numpy()[0]
        want = oracle.refine_trimap(probs, seg[0], bgr[0], edge_aware=bool(ea))
        assert np.array_equal(got, want)


def test_seed_from_prior_bit_exact(oracle, gpu_ctx):
    h, w = 24, 32
    yy, xx = np.mgrid[0:h, 0:w]
    seg = np.stack([(yy // 4) * 8 + xx // 4, (yy // 8) * 4 + xx // 8, (yy // 4) * 8 + xx // 4]).astype(np.int32)
    nn = np.array([48, 12, 48])
    node_ptr = np.concatenate([[0], np.cumsum(nn)]).astype(np.int32)
    rng = np.random.default_rng(3)
    prior = rng.random((node_ptr[-1], 3)).astype(np.float32)
    prior[5, 0] = prior[7, 0] = prior[:48, 0].max() + 0.1          # a tie at the top of image 0
    tri = np.stack([np.full((h, w), 2, np.uint8), np.full((h, w), 1, np.uint8), np.full((h, w), 3, np.uint8)])
    tri[2, 0, 0] = 0                                              # image 2 already has both sides
    td = torch.as_tensor(tri).cuda()
    pd, npd, sd = torch.as_tensor(prior).cuda(), torch.as_tensor(node_ptr).cuda(), torch.as_tensor(seg).cuda()
    gpu_ctx.call("ggc_seed_from_prior", gh.stream(), 3, h, w, pd.data_ptr(), npd.data_ptr(), sd.data_ptr(), 0.1,
                 td.data_ptr())
    got = td.cpu().numpy()
    for i in range(3):
        want = oracle.seed_from_prior(tri[i], prior[node_ptr[i]:node_ptr[i + 1]], seg[i], 0.1)
        assert np.array_equal(got[i], want), i
    assert (got[0] == 3).any() and (got[1] == 2).any() and np.array_equal(got[2], tri[2])
