"""
Generates tests/golden/reference_functions.npz by running the REFERENCE's own functions on fixed inputs.

Run in the build container only (the reference never travels):
    /opt/conda/bin/python3.9 tests/golden/make_golden_reference.py
python3.9 there has scikit-image 0.18.3 + numpy + networkx but no OpenCV, torch or PyG.  `cv2` is registered as an EMPTY
module so that the reference's "import cv2" lines succeed; nothing in it can be called, so every function recorded here
runs on numpy / scikit-image only.  Functions whose arithmetic goes through cv2 or PyG (GraphBuilder.__init__, slic(...,
channel_axis=), guided_filter, refine_trimap, clean_mask, GrabCut, boundary_f1, the networks) are NOT recorded: their
parity stays unpinned.

Recorded (reference file:line):
  graph_builder.py:190-226  GraphBuilder._region_statistics     (incl. skimage find_boundaries)
  graph_builder.py:228-255  GraphBuilder._assemble_node_features
  graph_builder.py:257-350  GraphBuilder._compute_edges / _pair_features / _nonlocal_pairs  (connectivity 4|8, k 0|4)
  graph_builder.py:357-454  compute_auto_prior (default and non-default sigmas), _unit_norm
  graph_builder.py:93-98    SuperpixelGraph.node_input
  pipeline.py:149-186       _seed_from_prior
  model.py:623-678          probs_to_node_trimap, project_to_pixels, _probs_to_trimap
  dataset.py:175-206        derive_trimap_labels
  metrics.py:58-102         evaluate (boundary_width=0: the confusion-count part)
Inputs: the five images / label maps of tests/golden/skimage_0183.npz (Lab / HSV planes as scikit-image 0.18.3 computed
them); the gradient plane, which the reference gets from cv2.Sobel, is INPUT DATA here (integer Sobel of the 4.x BGR2GRAY
fixed-point gray, written out below) and is stored with the outputs.
"""
import importlib.util
import pathlib
import sys
import types

import numpy as np

HERE = pathlib.Path(__file__).resolve().parent
REF = pathlib.Path("/root/reference/src/gcn_grabcut")

sys.modules["cv2"] = types.ModuleType("cv2")              # import succeeds, nothing callable
pkg = types.ModuleType("refpkg")
pkg.__path__ = [str(REF)]
sys.modules["refpkg"] = pkg


def load(name):
    spec = importlib.util.spec_from_file_location(f"refpkg.{name}", REF / f"{name}.py")
    m = importlib.util.module_from_spec(spec)
    sys.modules[f"refpkg.{name}"] = m
    spec.loader.exec_module(m)
    return m


gb = load("graph_builder")
load("grabcut")
metrics = load("metrics")
model = load("model")
dataset = load("dataset")
pipeline = load("pipeline")

gold = np.load(HERE / "skimage_0183.npz")
out = {}


def grad_plane(bgr):
    """input data for the statistics (not a reference function): |Sobel| of the integer gray plane, reflect-101 border"""
    b, g, r = (bgr[..., k].astype(np.int64) for k in range(3))
    gray = ((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15).astype(np.float32)
    p = np.pad(gray, 1, mode="reflect")
    gx = (p[:-2, 2:] + 2 * p[1:-1, 2:] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[1:-1, :-2] + p[2:, :-2])
    gy = (p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[:-2, 1:-1] + p[:-2, 2:])
    return gray, np.sqrt(gx ** 2 + gy ** 2).astype(np.float32)


for i in range(5):
    bgr = gold[f"c{i}_bgr"]
    seg = gold[f"c{i}_connected"].astype(np.int32)
    lab, hsv = gold[f"c{i}_lab"], gold[f"c{i}_hsv"]
    gray, grad = grad_plane(bgr)
    n = int(seg.max()) + 1
    out[f"c{i}_grad"] = grad
    for conn in (4, 8):
        for k in (0, 4):
            b = object.__new__(gb.GraphBuilder)           # __init__ needs cv2: the planes are attached instead
            b.bgr, b.rgb = bgr, bgr[..., ::-1]
            b.config = gb.SuperpixelGraphConfig(connectivity=conn, n_nonlocal=k)
            b._lab, b._hsv, b._gray, b._grad = lab, hsv, gray, grad
            st = b._region_statistics(seg, n)
            feats = b._assemble_node_features(seg, st)
            ei, ea = b._compute_edges(seg, st)
            tag = f"c{i}_conn{conn}_k{k}"
            out[f"{tag}_edge_index"] = np.asarray(ei, np.int64)
            out[f"{tag}_edge_attr"] = np.asarray(ea, np.float32)
            if conn == 4 and k == 4:
                for key in ("counts", "area_ratio", "mean_lab", "std_lab", "mean_hsv", "centroids", "boundary_px", "mean_grad", "mean_grad_n"):
                    out[f"c{i}_stat_{key}"] = np.asarray(st[key])
                out[f"c{i}_node_features"] = np.asarray(feats, np.float32)
    prior = gb.compute_auto_prior(seg, lab)
    out[f"c{i}_prior"] = np.asarray(prior, np.float32)
    out[f"c{i}_prior_s30_50"] = np.asarray(gb.compute_auto_prior(seg, lab, centre_sigma=0.30, contrast_sigma=0.50), np.float32)
    g = gb.SuperpixelGraph(segments=seg, node_features=out[f"c{i}_node_features"], edge_index=out[f"c{i}_conn4_k4_edge_index"],
                           edge_attr=out[f"c{i}_conn4_k4_edge_attr"], n_nodes=n, n_edges=out[f"c{i}_conn4_k4_edge_index"].shape[1],
                           node_centroids=out[f"c{i}_stat_centroids"], prior_features=prior, node_areas=out[f"c{i}_stat_area_ratio"])
    out[f"c{i}_node_input"] = np.asarray(g.node_input(), np.float32)

    # ---- _seed_from_prior: one-sided trimaps (all probable BG / all probable FG), a two-sided one, seed_frac 0.1 and 0.3
    for name, tri in (("allbg", np.full(seg.shape, 2, np.uint8)), ("allfg", np.full(seg.shape, 3, np.uint8)),
                      ("mixed", np.where(seg % 2 == 0, 2, 3).astype(np.uint8))):
        for frac in (0.1, 0.3):
            out[f"c{i}_seed_{name}_{int(frac * 10)}"] = pipeline._seed_from_prior(tri, g, frac)

    # ---- probs -> labels -> pixels
    rng = np.random.default_rng(100 + i)
    probs = rng.dirichlet((1.0, 1.0, 1.0), size=n).astype(np.float32)
    probs[0] = (0.55, 0.0, 0.45); probs[1 % n] = (0.45, 0.0, 0.55); probs[2 % n] = (0.5, 0.0, 0.5)      # threshold / tie cases
    out[f"c{i}_probs"] = probs
    out[f"c{i}_node_trimap"] = model.probs_to_node_trimap(probs, 0.55, 0.55)
    out[f"c{i}_node_trimap_65_60"] = model.probs_to_node_trimap(probs, 0.65, 0.60)
    out[f"c{i}_pixel_trimap"] = model._probs_to_trimap(probs, seg, 0.55, 0.55)
    out[f"c{i}_pixel_trimap_short"] = model._probs_to_trimap(probs[: max(1, n - 3)], seg, 0.55, 0.55)    # fewer rows than regions
    out[f"c{i}_projected_fg"] = model.project_to_pixels(probs[:, 2], seg)
    out[f"c{i}_projected_short"] = model.project_to_pixels(probs[: max(1, n - 3), 0], seg)

    # ---- dataset labels and the confusion-count metrics
    yy, xx = np.mgrid[0:seg.shape[0], 0:seg.shape[1]]
    gt = (((yy - seg.shape[0] * 0.45) / (seg.shape[0] * 0.3)) ** 2 + ((xx - seg.shape[1] * 0.55) / (seg.shape[1] * 0.25)) ** 2 <= 1).astype(np.uint8)
    out[f"c{i}_gt"] = gt
    out[f"c{i}_trimap_labels"] = dataset.derive_trimap_labels(seg, gt)
    out[f"c{i}_trimap_labels_60_90"] = dataset.derive_trimap_labels(seg, gt * 255, fg_threshold=0.6, bg_threshold=0.9)
    pred = np.roll(gt, (3, -5), (0, 1))
    out[f"c{i}_pred"] = pred
    m = metrics.evaluate(pred, gt, boundary_width=0)
    out[f"c{i}_metrics"] = np.array([m.iou, m.dice, m.precision, m.recall, m.f1, m.pixel_accuracy], np.float64)

import skimage
out["versions"] = np.array([f"numpy {np.__version__}", f"skimage {skimage.__version__}", f"python {sys.version.split()[0]}"])
np.savez_compressed(HERE / "reference_functions.npz", **out)
print(f"wrote {HERE / 'reference_functions.npz'}: {len(out)} arrays")
