"""CPU tests: oracle/graph.c (and the trimap / seeding / label / metric pieces) pinned against OUTPUTS OF THE REFERENCE'S OWN
FUNCTIONS — tests/golden/reference_functions.npz, written by tests/golden/make_golden_reference.py, which runs
graph_builder.py:190-454, pipeline.py:149-186, model.py:623-678, dataset.py:175-206 and metrics.py:58-102 of the reference
on numpy + scikit-image 0.18.3 — and against skimage's find_boundaries (tests/golden/skimage_0183.npz)."""
from pathlib import Path

import numpy as np
import pytest

GOLD = np.load(Path(__file__).parent / "golden" / "skimage_0183.npz")
REF = np.load(Path(__file__).parent / "golden" / "reference_functions.npz")


def _case(i):
    return GOLD[f"c{i}_connected"].astype(np.int32), GOLD[f"c{i}_lab"], GOLD[f"c{i}_hsv"], REF[f"c{i}_grad"]


@pytest.mark.parametrize("i", range(5))
def test_find_boundaries_matches_skimage(oracle, i):
    seg = GOLD[f"c{i}_connected"]
    assert np.array_equal(oracle.find_boundaries_inner(seg), GOLD[f"c{i}_boundaries"])


@pytest.mark.parametrize("i", range(5))
def test_preprocess_gradient_plane_is_the_fixture_input(oracle, i):
    # the fixture's gradient plane (input data of the reference run) is what the oracle's colour prep produces
    _, _, _, grad = oracle.preprocess(GOLD[f"c{i}_bgr"])
    assert np.array_equal(grad, REF[f"c{i}_grad"])


@pytest.mark.parametrize("i", range(5))
def test_node_features_match_the_reference(oracle, i):
    seg, lab, hsv, grad = _case(i)
    got = oracle.graph_build(seg, lab, hsv, grad, connectivity=4, n_nonlocal=4)
    assert got["n_nodes"] == REF[f"c{i}_stat_counts"].shape[0]
    want = REF[f"c{i}_node_features"]                     # graph_builder.py:190-255
    assert got["node_features"].shape == want.shape
    assert np.abs(got["node_features"] - want).max() <= 1e-6
    assert (got["node_features"] == want).mean() > 0.98
    assert np.array_equal(got["centroids"], REF[f"c{i}_stat_centroids"])
    assert np.array_equal(got["area_ratio"], REF[f"c{i}_stat_area_ratio"])
    x = np.concatenate([got["node_features"], got["prior"]], 1)         # graph_builder.py:93-98
    assert np.abs(x - REF[f"c{i}_node_input"]).max() <= 2e-5


@pytest.mark.parametrize("i", range(5))
@pytest.mark.parametrize("conn", [4, 8])
@pytest.mark.parametrize("k", [0, 4])
def test_edges_match_the_reference(oracle, i, conn, k):
    seg, lab, hsv, grad = _case(i)
    got = oracle.graph_build(seg, lab, hsv, grad, connectivity=conn, n_nonlocal=k)
    ei, ea = REF[f"c{i}_conn{conn}_k{k}_edge_index"], REF[f"c{i}_conn{conn}_k{k}_edge_attr"]     # graph_builder.py:257-350
    assert np.array_equal(got["edge_index"], ei)           # integer-exact, same order
    assert np.abs(got["edge_attr"] - ea).max() <= 1e-6
    # reference tests/test.py:87-155: shapes, ranges, symmetric storage
    assert got["edge_index"].shape == (2, got["n_edges"]) and got["edge_attr"].shape == (got["n_edges"], 5)
    half = got["n_edges"] // 2
    assert np.array_equal(got["edge_index"][0, :half], got["edge_index"][1, half:])
    if k == 0:
        assert (got["edge_attr"][:, 4] == 0).all()


@pytest.mark.parametrize("i", range(5))
def test_auto_prior_matches_the_reference(oracle, i):
    seg, lab, _, _ = _case(i)
    got = oracle.auto_prior(seg, lab)
    want = REF[f"c{i}_prior"]                             # graph_builder.py:357-454
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 2e-5
    assert np.isfinite(got).all() and got.min() >= -1e-5 and got.max() <= 1 + 1e-5
    got = oracle.auto_prior(seg, lab, centre_sigma=0.30, contrast_sigma=0.50)
    assert np.abs(got - REF[f"c{i}_prior_s30_50"]).max() <= 2e-5


@pytest.mark.parametrize("i", range(5))
def test_seed_from_prior_matches_the_reference(oracle, i):
    """pipeline.py:149-186.  The reference ranks regions with np.argsort (introsort: the order of EQUAL priors depends on the
    numpy build), so where the cut falls inside a group of equal priors only the promoted priors are comparable, not the
    region ids; everywhere else the trimaps are identical."""
    seg = GOLD[f"c{i}_connected"].astype(np.int32)
    prior = REF[f"c{i}_prior"]
    n = prior.shape[0]
    exact = 0
    for name, tri in (("allbg", np.full(seg.shape, 2, np.uint8)), ("allfg", np.full(seg.shape, 3, np.uint8)),
                      ("mixed", np.where(seg % 2 == 0, 2, 3).astype(np.uint8))):
        for frac in (0.1, 0.3):
            want = REF[f"c{i}_seed_{name}_{int(frac * 10)}"]
            got = oracle.seed_from_prior(tri, prior, seg, frac)
            if name == "mixed":
                assert np.array_equal(got, want) and np.array_equal(got, tri)
                continue
            col, lab = (0, 3) if name == "allbg" else (1, 2)
            n_seed = max(1, int(round(frac * n)))
            v = np.sort(prior[:, col])[::-1]
            tie_at_cut = n_seed < n and v[n_seed - 1] == v[n_seed]
            if not tie_at_cut:
                assert np.array_equal(got, want), (name, frac)
                exact += 1
            r_got, r_want = np.unique(seg[got == lab]), np.unique(seg[want == lab])
            assert len(r_got) == len(r_want) == n_seed
            assert np.array_equal(np.sort(prior[r_got, col]), np.sort(prior[r_want, col]))
            assert set(np.unique(got)) == {2, 3}
    assert exact >= 1 or i == 3


@pytest.mark.parametrize("i", range(5))
def test_probs_to_trimap_matches_the_reference(oracle, i):
    seg = GOLD[f"c{i}_connected"].astype(np.int32)
    probs, bgr = REF[f"c{i}_probs"], GOLD[f"c{i}_bgr"]
    got = oracle.refine_trimap(probs, seg, bgr, 0.55, 0.55, edge_aware=False)      # model.py:623-678
    assert np.array_equal(got, REF[f"c{i}_pixel_trimap"])
    n = probs.shape[0]
    got = oracle.refine_trimap(probs[: max(1, n - 3)], seg, bgr, 0.55, 0.55, edge_aware=False)   # fewer rows than regions
    assert np.array_equal(got, REF[f"c{i}_pixel_trimap_short"])
    # node labels (model.py:623-645) are the pixel labels of any pixel of the region
    first = np.array([np.flatnonzero(seg.ravel() == r)[0] for r in range(n)])
    assert np.array_equal(REF[f"c{i}_pixel_trimap"].ravel()[first], REF[f"c{i}_node_trimap"])


def test_region_zero_has_no_boundary_pixels(oracle):
    # skimage's mode="inner" masks label 0: feature 12 of node 0 is 1, feature 14 is 0
    seg, lab, hsv, grad = _case(2)
    got = oracle.graph_build(seg, lab, hsv, grad)
    assert got["node_features"][0, 12] == 1.0 and got["node_features"][0, 14] == 0.0
    assert REF["c2_node_features"][0, 12] == 1.0 and REF["c2_node_features"][0, 14] == 0.0


def test_tiny_graphs(oracle):
    two = np.zeros((8, 8), np.int32); two[:, 4:] = 1
    l8, h8, _, g8 = oracle.preprocess(GOLD["c0_bgr"][:8, :8])
    gt = oracle.graph_build(two, l8, h8, g8)            # N = 2 <= k + 1: no non-local edges
    assert gt["n_nodes"] == 2 and gt["n_edges"] == 2
    one = np.zeros((8, 8), np.int32)
    g1 = oracle.graph_build(one, l8, h8, g8)
    assert g1["n_nodes"] == 1 and g1["n_edges"] == 0 and np.isfinite(g1["prior"]).all()


@pytest.mark.parametrize("i", range(5))
def test_trimap_labels_match_the_reference(i):
    """dataset.py:175-206 through the product's host formula (the per-region counts it needs come from ggc_region_label_stats
    on the device; tests/test_dataset_gpu.py checks those counts)."""
    from gcn_grabcut import dataset as ds
    seg, gt = GOLD[f"c{i}_connected"].astype(np.int32), REF[f"c{i}_gt"]
    got = ds.derive_trimap_labels(seg, gt, 0.75, 0.75)
    assert got.dtype == np.int64 and np.array_equal(got, REF[f"c{i}_trimap_labels"])
    assert np.array_equal(ds.derive_trimap_labels(seg, gt * 255, 0.6, 0.9), REF[f"c{i}_trimap_labels_60_90"])


@pytest.mark.parametrize("i", range(5))
def test_metric_formulas_match_the_reference(oracle, i):
    """metrics.py:58-102 (boundary_width=0): oracle tallies -> the product's formulas == the reference's numbers, to the bit"""
    from gcn_grabcut import metrics as gm
    pred, gt = REF[f"c{i}_pred"], REF[f"c{i}_gt"]
    c = oracle.eval_counts(pred, gt, None, 0)
    m = gm._from_counts(np.asarray(c), pred.size, False)
    got = np.array([m.iou, m.dice, m.precision, m.recall, m.f1, m.pixel_accuracy], np.float64)
    assert np.array_equal(got, REF[f"c{i}_metrics"])
    assert oracle.iou(pred, gt) == REF[f"c{i}_metrics"][0]
