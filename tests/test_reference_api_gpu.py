"""The reference's own API-level tests (tests/test.py) for the rows widened into this round — metrics (:204-248),
derive_trimap_labels (:193-202), model variants (:275-306, :330-345), pipeline extras (:450-467) — re-expressed for
this build (PyG-free Data/Batch, device "cuda"; the GAT variant is a documented gap)."""
import numpy as np
import pytest
import torch

from helpers import chain_graph

pytestmark = pytest.mark.gpu


def _img(h=64, w=64, seed=42):
    return np.random.RandomState(seed).randint(20, 220, (h, w, 3), dtype=np.uint8)       # tests/test.py:16-18


def _circle_mask(h=64, w=64, r=20):
    yy, xx = np.mgrid[:h, :w]                                                            # cv2.circle(..., 1, -1)
    return (((yy - h // 2) ** 2 + (xx - w // 2) ** 2) <= r * r).astype(np.uint8)


def _data(n=80, seed=None):
    from gcn_grabcut.data import Data
    x, ei, ea = chain_graph(n, seed=seed)
    return Data(x=torch.as_tensor(x), edge_index=torch.as_tensor(ei), edge_attr=torch.as_tensor(ea)).to("cuda")


def test_metrics_like_reference():
    from gcn_grabcut.grabcut import Label
    from gcn_grabcut.metrics import boundary_f1, evaluate, evaluate_trimap
    gt = _circle_mask()
    m = evaluate(gt, gt)
    assert m.iou == pytest.approx(1.0, abs=1e-4) and m.dice == pytest.approx(1.0, abs=1e-4) and m.recall == pytest.approx(1.0, abs=1e-4)
    assert evaluate(np.zeros_like(gt), gt).iou < 0.01
    pred = (np.random.RandomState(5).rand(64, 64) > 0.5).astype(np.uint8)
    assert 0 <= evaluate(pred, gt).iou <= 1
    assert boundary_f1(gt, gt) == pytest.approx(1.0, abs=1e-3)
    tm = evaluate_trimap(np.where(gt, Label.FG_DEFINITE, Label.BG_DEFINITE).astype(np.uint8), gt)
    assert tm.fg_recall > 0.95 and tm.bg_recall > 0.95 and tm.bg_contamination < 0.01
    d = m.as_dict()
    assert "iou" in d and "dice" in d


def test_derive_trimap_labels_like_reference():
    from gcn_grabcut.dataset import derive_trimap_labels
    from gcn_grabcut.graph_builder import GraphBuilder
    graph = GraphBuilder(_img(64, 64)).build()
    labels = derive_trimap_labels(graph.segments, _circle_mask())
    assert labels.shape == (graph.n_nodes,) and set(np.unique(labels)).issubset({0, 1, 2})


@pytest.mark.parametrize("variant", ["gcn", "resgcn"])
def test_model_variants_like_reference(variant):
    from gcn_grabcut.data import Batch
    from gcn_grabcut.model import build_model
    model = build_model(variant=variant, hidden_channels=32, n_layers=2).to("cuda").eval()
    assert model(_data()).shape == (80, 3)                                              # tests/test.py:275-280
    graphs = [_data(40, seed=s) for s in (1, 2, 3)]
    one_by_one = torch.cat([model(g) for g in graphs])
    batched = model(Batch.from_data_list(graphs))
    assert torch.allclose(one_by_one, batched, atol=1e-4), variant                       # :294-306
    segs = np.zeros((32, 32), dtype=np.int32); segs[16:, :] = 1
    tri = model.predict_trimap(_data(2, seed=3), segs)                                   # :330-345
    assert tri.shape == (32, 32) and set(np.unique(tri)).issubset({0, 1, 2, 3})


def test_pipeline_extras_like_reference():
    from gcn_grabcut.model import GCNTrimapNet
    from gcn_grabcut.pipeline import GCNGrabCutPipeline
    img, gt = _img(100, 100), _circle_mask(100, 100)
    pipeline = GCNGrabCutPipeline(GCNTrimapNet(hidden_channels=16, n_layers=2).eval(), device="cuda")   # the reference's sizes
    assert pipeline.segment_bbox(img, (10, 10, 80, 80)).binary_mask.shape == (100, 100)  # :450-458
    seg_m, tri_m = pipeline.segment(img).evaluate_against(gt)                            # :460-467
    assert 0 <= seg_m.iou <= 1 and 0 <= tri_m.trimap_accuracy <= 1
    with pytest.raises(ValueError, match="hidden_channels"):
        GCNTrimapNet(hidden_channels=130, n_layers=2)                                    # wider than the kernels are built for


def test_new_entries_fail_loudly_on_bad_arguments(gpu_ctx):
    """Every C-ABI entry added for the section 8(f) rows reports misuse through its return code (no crash, no silent no-op)."""
    from gcn_grabcut import _native
    st = _native.current_stream(0)
    buf = torch.zeros(64, dtype=torch.uint8, device="cuda")
    p = buf.data_ptr()
    bad = [
        ("ggc_eval_counts", (st, 0, 4, 4, p, p, None, 3, p)),                  # batch 0
        ("ggc_eval_counts", (st, 1, 4, 4, None, p, None, 3, p)),               # null prediction
        ("ggc_eval_counts", (st, 1, 4, 4, p, p, None, 1000, p)),               # boundary width out of range
        ("ggc_region_label_stats", (st, 1, 4, 4, None, p, p, p, p)),           # null segments
        ("ggc_convert_color8", (st, 16, p, 2, p)),                             # unknown mode
        ("ggc_convert_color8", (st, 0, p, 0, p)),                              # no pixels
        ("ggc_gcnnet_configure", (48, 2)),                                     # width the MFMA tiling cannot take
        ("ggc_gcnnet_configure", (32, 0)),                                     # no layers
    ]
    for name, args in bad:
        with pytest.raises(_native.GGCError):
            gpu_ctx.call(name, *args)
    gpu_ctx.call("ggc_gcnnet_configure", 64, 7)                                # a configuration no other test has loaded
    x = torch.zeros(4, 19, device="cuda")
    out = torch.zeros(4, 3, device="cuda")
    with pytest.raises(_native.GGCError, match="missing weight"):
        gpu_ctx.call("ggc_gcnnet_forward", st, 4, 0, x.data_ptr(), None, None, None, out.data_ptr(), None)   # weights never loaded
