"""CPU tests: the C oracle's colour prep and SLIC against the golden vectors made
with scikit-image 0.18.3 / scipy 1.7.1 (tests/golden/make_golden_skimage.py)."""
from pathlib import Path

import numpy as np
import pytest

GOLD = np.load(Path(__file__).parent / "golden" / "skimage_0183.npz")
N_CASES = 5


def case(i):
    pre = f"c{i}_"
    return {k[len(pre):]: GOLD[k] for k in GOLD.files if k.startswith(pre)}


@pytest.mark.parametrize("i", range(N_CASES))
def test_rgb2lab_rgb2hsv_match_skimage(oracle, i):
    c = case(i)
    lab, hsv, gray, grad = oracle.preprocess(c["bgr"])
    # f64 pipeline rounded to f32: our cbrt / pow(.,2.4) differ from libm by < 1e-15 relative
    assert np.abs(lab - c["lab"]).max() <= 2e-5
    assert (lab == c["lab"]).mean() > 0.999
    assert np.abs(hsv - c["hsv"]).max() <= 1e-6
    assert (hsv == c["hsv"]).mean() > 0.9999


@pytest.mark.parametrize("i", range(N_CASES))
def test_second_lab_f32_close_to_skimage(oracle, i):
    c = case(i)
    got = oracle.slic_rescale_lab(c["lab"], rescale_input=True)
    # numpy's float32 power / cbrt are not correctly rounded; ours are (via f64): allow a few ulp
    assert np.abs(got - c["second_lab"]).max() <= 5e-4
    got0 = oracle.slic_rescale_lab(c["rescaled"], rescale_input=False)
    assert np.array_equal(got, got0)


@pytest.mark.parametrize("i", range(N_CASES))
def test_gaussian_bit_exact_vs_scipy(oracle, i):
    c = case(i)
    got = oracle.gaussian(c["second_lab"], 1.0)
    assert np.array_equal(got, c["smoothed"])


@pytest.mark.parametrize("i", range(N_CASES))
def test_grid_seeds_match_regular_grid(oracle, i):
    c = case(i)
    h, w = c["raw"].shape
    g = oracle.slic_grid(h, w, int(c["n_segments"]))
    ys = g["start_y"] + g["step_y"] * np.arange(g["ny"])
    xs = g["start_x"] + g["step_x"] * np.arange(g["nx"])
    seeds = np.stack(np.meshgrid(ys, xs, indexing="ij"), -1).reshape(-1, 2).astype(np.float32)
    assert np.array_equal(seeds, c["seeds"])
    assert float(max(g["step_y"], g["step_x"])) == float(c["step"])


@pytest.mark.parametrize("i", range(N_CASES))
def test_kmeans_bit_exact_vs_slic_cython(oracle, i):
    c = case(i)
    labels, centers = oracle.slic_kmeans(c["scaled"], c["seeds"], float(c["step"]))
    assert np.array_equal(labels, c["raw"])
    want = c["centers_final"][:, 1:]         # (z,y,x,c0,c1,c2) -> drop z
    assert np.array_equal(centers, want, equal_nan=True)


@pytest.mark.parametrize("i", range(N_CASES))
def test_connectivity_bit_exact_vs_skimage(oracle, i):
    c = case(i)
    out, n = oracle.slic_connectivity(c["raw"], int(c["min_size"]), int(c["max_size"]))
    assert np.array_equal(out, c["connected"])
    assert n == c["connected"].max() + 1
    assert np.array_equal(np.unique(out), np.arange(n))       # reference tests/test.py:112-117


@pytest.mark.parametrize("tag,mn,mx", [("a", 4, 200), ("b", 12, 60), ("c", 1, 10 ** 6)])
def test_connectivity_stress(oracle, tag, mn, mx):
    out, _ = oracle.slic_connectivity(GOLD["stress_in"], mn, mx)
    assert np.array_equal(out, GOLD[f"stress_{tag}"])


@pytest.mark.parametrize("i", range(N_CASES))
def test_whole_slic_agrees_with_skimage_wrapper(oracle, i):
    """End to end the only difference is the last-ulp behaviour of the second
    (float32) rgb2lab, so the label maps agree except for a handful of pixels."""
    c = case(i)
    seg, n = oracle.slic(c["lab"], int(c["n_segments"]), 10.0, 1.0, rescale_input=True)
    assert seg.shape == c["connected"].shape
    agree = (seg == c["connected"]).mean()
    assert agree > 0.97, agree
    assert np.array_equal(np.unique(seg), np.arange(n))


# ---------------------------------------------------------------- use_lab=False: the float64 path
# SuperpixelGraphConfig(use_lab=False) hands rgb.astype(float) to slic (reference graph_builder.py:177-179); a float64 input
# keeps rgb2lab, the Gaussian and _slic_cython's fused-type kernel in float64.  Golden: tests/golden/make_golden_skimage_rgb.py.
GOLD_RGB = np.load(Path(__file__).parent / "golden" / "skimage_0183_rgb.npz")
N_RGB = 4


def case_rgb(i):
    pre = f"r{i}_"
    return {k[len(pre):]: GOLD_RGB[k] for k in GOLD_RGB.files if k.startswith(pre)}


@pytest.mark.parametrize("i", range(N_RGB))
def test_f64_second_lab_close_to_skimage(oracle, i):
    c = case_rgb(i)
    rgb = c["bgr"][:, :, ::-1].astype(np.float64)
    got = oracle.slic_rescale_lab64(rgb, rescale_input=True)
    assert np.abs(got - c["second_lab"]).max() <= 1e-12          # numpy's pow / cbrt vs the fixed sequences: a few ulp of 100
    assert np.array_equal(got, oracle.slic_rescale_lab64(c["rescaled"], rescale_input=False))


@pytest.mark.parametrize("i", range(N_RGB))
def test_f64_gaussian_bit_exact_vs_scipy(oracle, i):
    c = case_rgb(i)
    assert np.array_equal(oracle.gaussian64(c["second_lab"], 1.0), c["smoothed"])


@pytest.mark.parametrize("i", range(N_RGB))
def test_f64_kmeans_bit_exact_vs_slic_cython(oracle, i):
    c = case_rgb(i)
    h, w = c["raw"].shape
    g = oracle.slic_grid(h, w, int(c["n_segments"]))
    ys = g["start_y"] + g["step_y"] * np.arange(g["ny"])
    xs = g["start_x"] + g["step_x"] * np.arange(g["nx"])
    seeds = np.stack(np.meshgrid(ys, xs, indexing="ij"), -1).reshape(-1, 2).astype(np.float64)
    labels, centers = oracle.slic_kmeans64(c["scaled"], seeds, float(c["step"]))
    assert np.array_equal(labels, c["raw"])
    assert np.array_equal(centers, c["centers_final"][:, 1:], equal_nan=True)


@pytest.mark.parametrize("i", range(N_RGB))
def test_f64_whole_slic_agrees_with_skimage_wrapper(oracle, i):
    """end to end only the last ulps of pow / cbrt differ from numpy's: the label maps agree except for a handful of pixels"""
    c = case_rgb(i)
    seg, n = oracle.slic_rgb(c["bgr"], int(c["n_segments"]), 10.0, 1.0)
    assert (seg == c["connected"]).mean() > 0.97
    assert np.array_equal(np.unique(seg), np.arange(n))
