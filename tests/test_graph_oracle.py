"""CPU tests: oracle/graph.c against skimage's find_boundaries (golden) and an
independent numpy formulation of the graph stage."""
from pathlib import Path

import numpy as np
import pytest

import numpy_graph_ref as ref

GOLD = np.load(Path(__file__).parent / "golden" / "skimage_0183.npz")


def _case(oracle, i):
    bgr = GOLD[f"c{i}_bgr"]
    seg = GOLD[f"c{i}_connected"]
    lab, hsv, gray, grad = oracle.preprocess(bgr)
    return seg, lab, hsv, grad


@pytest.mark.parametrize("i", range(5))
def test_find_boundaries_matches_skimage(oracle, i):
    seg = GOLD[f"c{i}_connected"]
    assert np.array_equal(oracle.find_boundaries_inner(seg), GOLD[f"c{i}_boundaries"])


@pytest.mark.parametrize("i", range(5))
@pytest.mark.parametrize("conn", [4, 8])
def test_graph_matches_numpy_formulation(oracle, i, conn):
    seg, lab, hsv, grad = _case(oracle, i)
    got = oracle.graph_build(seg, lab, hsv, grad, connectivity=conn, n_nonlocal=4)
    st = ref.region_stats(seg, lab, hsv, grad, oracle.find_boundaries_inner(seg))
    assert got["n_nodes"] == st["n"]
    x = ref.node_features(st)
    assert np.abs(got["node_features"] - x).max() <= 1e-6
    assert (got["node_features"] == x).mean() > 0.98
    assert np.array_equal(got["centroids"], st["cen"])
    assert np.array_equal(got["area_ratio"], st["area"])
    ei, ea = ref.edges(seg, st, conn, 4)
    assert np.array_equal(got["edge_index"], ei)          # integer-exact, same order
    assert np.abs(got["edge_attr"] - ea).max() <= 1e-6
    # reference tests/test.py:87-155: shapes, ranges, symmetric storage
    assert got["edge_index"].shape == (2, got["n_edges"]) and got["edge_attr"].shape == (got["n_edges"], 5)
    half = got["n_edges"] // 2
    assert np.array_equal(got["edge_index"][0, :half], got["edge_index"][1, half:])
    assert got["node_features"][:, :6].min() >= -0.01 and got["node_features"][:, :6].max() <= 1.01


@pytest.mark.parametrize("i", range(5))
def test_auto_prior_matches_numpy_formulation(oracle, i):
    seg, lab, _, _ = _case(oracle, i)
    got = oracle.auto_prior(seg, lab)
    want = ref.auto_prior(seg, lab)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 2e-5
    assert np.isfinite(got).all() and got.min() >= -1e-5 and got.max() <= 1 + 1e-5


def test_region_zero_has_no_boundary_pixels(oracle):
    # skimage's mode="inner" masks label 0: feature 12 of node 0 is 1, feature 14 is 0
    seg, lab, hsv, grad = _case(oracle, 2)
    got = oracle.graph_build(seg, lab, hsv, grad)
    assert got["node_features"][0, 12] == 1.0 and got["node_features"][0, 14] == 0.0


def test_nonlocal_disabled_and_tiny_graphs(oracle):
    seg, lab, hsv, grad = _case(oracle, 0)
    g0 = oracle.graph_build(seg, lab, hsv, grad, n_nonlocal=0)
    assert (g0["edge_attr"][:, 4] == 0).all()
    two = np.zeros((8, 8), np.int32); two[:, 4:] = 1
    l8, h8, _, g8 = oracle.preprocess(GOLD["c0_bgr"][:8, :8])
    gt = oracle.graph_build(two, l8, h8, g8)            # N = 2 <= k + 1: no non-local edges
    assert gt["n_nodes"] == 2 and gt["n_edges"] == 2
    one = np.zeros((8, 8), np.int32)
    g1 = oracle.graph_build(one, l8, h8, g8)
    assert g1["n_nodes"] == 1 and g1["n_edges"] == 0 and np.isfinite(g1["prior"]).all()
