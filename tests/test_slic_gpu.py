"""GPU parity for G0 (colour prep) and G1 (SLIC): bit-exact against the CPU oracle
through the C ABI, on the golden inputs and on DUTS-shaped synthetic images."""
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = np.load(Path(__file__).parent / "golden" / "skimage_0183.npz")


def _pre(gpu_ctx, bgr):
    from gcn_grabcut import _native
    b, h, w, _ = bgr.shape
    d = torch.as_tensor(np.ascontiguousarray(bgr)).cuda().contiguous()
    lab = torch.empty(b, h, w, 3, device="cuda")
    hsv = torch.empty(b, h, w, 3, device="cuda")
    gray = torch.empty(b, h, w, device="cuda")
    grad = torch.empty(b, h, w, device="cuda")
    gpu_ctx.call("ggc_preprocess", _native.current_stream(0), b, h, w, d.data_ptr(), lab.data_ptr(), hsv.data_ptr(),
                 gray.data_ptr(), grad.data_ptr())
    return lab, hsv, gray, grad


def _slic(gpu_ctx, lab, n_segments, compactness=10.0, sigma=1.0, rescale=1):
    from gcn_grabcut import _native
    lab = lab.contiguous()
    b, h, w, _ = lab.shape
    seg = torch.empty(b, h, w, dtype=torch.int32, device="cuda")
    n = torch.empty(b, dtype=torch.int32, device="cuda")
    gpu_ctx.call("ggc_slic", _native.current_stream(0), b, h, w, lab.data_ptr(), n_segments, compactness, sigma, rescale,
                 seg.data_ptr(), n.data_ptr())
    return seg.cpu().numpy(), n.cpu().numpy()


@pytest.mark.parametrize("h,w,b", [(64, 64, 2), (50, 81, 3), (300, 400, 2)])
def test_preprocess_bit_exact(oracle, gpu_ctx, h, w, b):
    from gcn_grabcut.synthetic import synthetic_batch
    bgr = synthetic_batch(b, h, w, config_id=1)
    lab, hsv, gray, grad = [t.cpu().numpy() for t in _pre(gpu_ctx, bgr)]
    for i in range(b):
        wl, wh, wg, wd = oracle.preprocess(bgr[i])
        assert np.array_equal(lab[i], wl)
        assert np.array_equal(hsv[i], wh)
        assert np.array_equal(gray[i], wg)
        assert np.array_equal(grad[i], wd)


def test_preprocess_extreme_colours(oracle, gpu_ctx):
    # all 256 grey levels, saturated primaries, black, white: every branch of Lab / HSV
    vals = np.arange(256, dtype=np.uint8)
    img = np.zeros((1, 8, 256, 3), np.uint8)
    img[0, 0] = vals[:, None]
    img[0, 1, :, 0] = vals
    img[0, 2, :, 1] = vals
    img[0, 3, :, 2] = vals
    img[0, 4, :, 0] = 255 - vals; img[0, 4, :, 1] = vals
    img[0, 5, :, 1] = 255 - vals; img[0, 5, :, 2] = vals
    img[0, 6] = 255
    lab, hsv, gray, grad = [t.cpu().numpy() for t in _pre(gpu_ctx, img)]
    wl, wh, wg, wd = oracle.preprocess(img[0])
    assert np.array_equal(lab[0], wl) and np.array_equal(hsv[0], wh)
    assert np.array_equal(gray[0], wg) and np.array_equal(grad[0], wd)


@pytest.mark.parametrize("i", range(5))
def test_slic_bit_exact_on_golden_inputs(oracle, gpu_ctx, i):
    lab = GOLD[f"c{i}_lab"]
    n_seg = int(GOLD[f"c{i}_n_segments"])
    want, wn = oracle.slic(lab, n_seg, 10.0, 1.0, True)
    got, gn = _slic(gpu_ctx, torch.as_tensor(lab[None]).cuda(), n_seg)
    if not np.array_equal(got[0], want):
        h, w = want.shape
        raw = np.empty((h, w), np.int32)
        gpu_ctx.call("ggc_debug_read_scratch", b"slic_raw_labels", raw.ctypes.data, raw.nbytes)
        scaled = oracle.gaussian(oracle.slic_rescale_lab(lab, True), 1.0) * np.float32(0.1)
        g = oracle.slic_grid(h, w, n_seg)
        ys = g["start_y"] + g["step_y"] * np.arange(g["ny"])
        xs = g["start_x"] + g["step_x"] * np.arange(g["nx"])
        seeds = np.stack(np.meshgrid(ys, xs, indexing="ij"), -1).reshape(-1, 2).astype(np.float32)
        wraw, _ = oracle.slic_kmeans(scaled, seeds, float(max(g["step_y"], g["step_x"])))
        pytest.fail(f"label map differs: raw k-means labels differ at {(raw != wraw).sum()} px, "
                    f"final at {(got[0] != want).sum()} px")
    assert gn[0] == wn
    # the skimage 0.18.3 wrapper agrees except for last-ulp effects of its float32 pow/cbrt
    assert (got[0] == GOLD[f"c{i}_connected"]).mean() > 0.97


def test_slic_bit_exact_duts_shape_batch(oracle, gpu_ctx):
    from gcn_grabcut.synthetic import synthetic_batch
    bgr = synthetic_batch(3, 300, 400, config_id=2)
    lab = _pre(gpu_ctx, bgr)[0]
    got, gn = _slic(gpu_ctx, lab, 600)
    lab_h = lab.cpu().numpy()
    for i in range(3):
        want, wn = oracle.slic(lab_h[i], 600, 10.0, 1.0, True)
        assert np.array_equal(got[i], want), (i, int((got[i] != want).sum()))
        assert gn[i] == wn
        assert np.array_equal(np.unique(got[i]), np.arange(wn))        # reference tests/test.py:112-117
        assert 450 <= wn <= 700


@pytest.mark.parametrize("h,w,n_seg,sigma,comp", [(37, 53, 20, 0.0, 10.0), (64, 64, 300, 1.0, 20.0), (31, 200, 64, 2.0, 5.0)])
def test_slic_odd_shapes_and_parameters(oracle, gpu_ctx, h, w, n_seg, sigma, comp):
    from gcn_grabcut.synthetic import synthetic_image
    bgr = synthetic_image(h, w, 99)[None]
    lab = _pre(gpu_ctx, bgr)[0]
    got, gn = _slic(gpu_ctx, lab, n_seg, comp, sigma)
    want, wn = oracle.slic(lab.cpu().numpy()[0], n_seg, comp, sigma, True)
    assert np.array_equal(got[0], want) and gn[0] == wn


def test_slic_constant_image(oracle, gpu_ctx):
    # max == min: the rescale must not divide by zero; every cluster sees identical colours
    lab = torch.full((1, 48, 64, 3), 42.0, device="cuda")
    got, gn = _slic(gpu_ctx, lab, 30)
    want, wn = oracle.slic(lab.cpu().numpy()[0], 30, 10.0, 1.0, True)
    assert np.array_equal(got[0], want) and gn[0] == wn


def _connectivity(gpu_ctx, raw, mn, mx):
    from gcn_grabcut import _native
    b, h, w = raw.shape
    d = torch.as_tensor(np.ascontiguousarray(raw, dtype=np.int32)).cuda()
    out = torch.empty_like(d)
    n = torch.empty(b, dtype=torch.int32, device="cuda")
    gpu_ctx.call("ggc_slic_enforce_connectivity", _native.current_stream(0), b, h, w, d.data_ptr(), int(mn), int(mx),
                 out.data_ptr(), n.data_ptr())
    return out.cpu().numpy(), n.cpu().numpy()


@pytest.mark.parametrize("tag,mn,mx", [("a", 4, 200), ("b", 12, 60), ("c", 1, 10 ** 6)])
def test_connectivity_stress_vs_skimage_golden(gpu_ctx, tag, mn, mx):
    """components far above max_size (many carve rounds) and many tiny fragments (merge chains)"""
    got, n = _connectivity(gpu_ctx, GOLD["stress_in"][None], mn, mx)
    assert np.array_equal(got[0], GOLD[f"stress_{tag}"])
    assert n[0] == GOLD[f"stress_{tag}"].max() + 1


def test_connectivity_random_label_maps_batch(oracle, gpu_ctx):
    rng = np.random.default_rng(5)
    for trial in range(6):
        h, w, b = int(rng.integers(9, 70)), int(rng.integers(9, 90)), 5
        k = int(rng.integers(2, 7))
        raw = rng.integers(0, k, (b, h, w)).astype(np.int32)
        raw[1] = np.kron(rng.integers(0, k, ((h + 4) // 5, (w + 4) // 5)), np.ones((5, 5), int))[:h, :w]
        raw[2] = 0                                                     # one giant component
        raw[3, ::2] = 1; raw[3, 1::2] = 0                              # stripes
        mn = int(rng.integers(0, 15)); mx = int(rng.integers(max(mn, 1), 120))
        got, n = _connectivity(gpu_ctx, raw, mn, mx)
        for i in range(b):
            want, wn = oracle.slic_connectivity(raw[i], mn, mx)
            assert np.array_equal(got[i], want), (trial, i, h, w, mn, mx, int((got[i] != want).sum()))
            assert n[i] == wn


def test_connectivity_matches_on_real_kmeans_output(oracle, gpu_ctx):
    for i in range(5):
        raw = GOLD[f"c{i}_raw"]
        got, n = _connectivity(gpu_ctx, raw[None], int(GOLD[f"c{i}_min_size"]), int(GOLD[f"c{i}_max_size"]))
        assert np.array_equal(got[0], GOLD[f"c{i}_connected"])


# ---------------------------------------------------------------- use_lab=False: skimage's float64 path
GOLD_RGB = np.load(Path(__file__).parent / "golden" / "skimage_0183_rgb.npz")


def _slic_rgb(gpu_ctx, bgr, n_segments, compactness=10.0, sigma=1.0):
    from gcn_grabcut import _native
    d = torch.as_tensor(np.ascontiguousarray(bgr)).cuda().contiguous()
    b, h, w, _ = d.shape
    seg = torch.empty(b, h, w, dtype=torch.int32, device="cuda")
    n = torch.empty(b, dtype=torch.int32, device="cuda")
    gpu_ctx.call("ggc_slic_rgb", _native.current_stream(0), b, h, w, d.data_ptr(), n_segments, float(compactness), float(sigma),
                 seg.data_ptr(), n.data_ptr())
    return seg.cpu().numpy(), n.cpu().numpy()


@pytest.mark.parametrize("i", range(4))
def test_slic_rgb_golden_inputs_bit_exact_vs_oracle_and_close_to_skimage(oracle, gpu_ctx, i):
    """SuperpixelGraphConfig(use_lab=False) = slic(rgb.astype(float)) (reference graph_builder.py:177-179): label map identical to
    the oracle's float64 restatement, which is pinned against scikit-image 0.18.3's double kernels (test_slic_oracle.py)."""
    bgr = GOLD_RGB[f"r{i}_bgr"]
    n_seg = int(GOLD_RGB[f"r{i}_n_segments"])
    seg, n = _slic_rgb(gpu_ctx, bgr[None], n_seg)
    want, wn = oracle.slic_rgb(bgr, n_seg)
    assert np.array_equal(seg[0], want) and n[0] == wn
    assert (seg[0] == GOLD_RGB[f"r{i}_connected"]).mean() > 0.97       # the whole skimage call: only pow / cbrt ulps differ


@pytest.mark.parametrize("h,w,b,n_seg,sigma,comp", [(300, 400, 3, 600, 1.0, 10.0), (97, 131, 2, 80, 0.0, 10.0), (120, 160, 2, 150, 1.5, 20.0),
                                                    (33, 47, 1, 2000, 1.0, 10.0)])
def test_slic_rgb_batches_bit_exact_vs_oracle(oracle, gpu_ctx, h, w, b, n_seg, sigma, comp):
    from gcn_grabcut.synthetic import synthetic_batch
    bgr = synthetic_batch(b, h, w, config_id=4)
    seg, n = _slic_rgb(gpu_ctx, bgr, n_seg, comp, sigma)
    for k in range(b):
        want, wn = oracle.slic_rgb(bgr[k], n_seg, comp, sigma)
        assert np.array_equal(seg[k], want), (k, int((seg[k] != want).sum()))
        assert n[k] == wn
