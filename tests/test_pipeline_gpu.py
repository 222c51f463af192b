"""GPU end-to-end parity: GCNGrabCutPipeline (host mirror + HIP kernels) against
the CPU oracle chain, with the parity gates of SURVEY section 8(d); plus the
reference's own API-level tests (tests/test.py) re-expressed for this build."""
import numpy as np
import pytest
import torch

from helpers import seeded_state_dict

pytestmark = pytest.mark.gpu


def _np_state(sd):
    return {k: v.numpy() for k, v in sd.items() if v.dtype.is_floating_point}


def _img(h=64, w=64, seed=42):
    return np.random.RandomState(seed).randint(20, 220, (h, w, 3), dtype=np.uint8)   # reference tests/test.py:17-19


@pytest.fixture(scope="module")
def pipe128():
    from gcn_grabcut import GCNGrabCutPipeline, SuperpixelGraphConfig
    model, sd = seeded_state_dict(128, 6, seed=0)
    return GCNGrabCutPipeline(model.eval(), sp_config=SuperpixelGraphConfig(n_segments=600), device="cuda"), sd


def test_full_pipeline_matches_oracle_on_duts_shape_batch(oracle, pipe128):
    from gcn_grabcut.synthetic import synthetic_batch
    pipe, sd = pipe128
    imgs = synthetic_batch(3, 300, 400, config_id=3)
    res = pipe.segment_batch(list(imgs))
    out = pipe.segment_batch_device(pipe._eng.to_device(imgs))
    g = out["graphs"]
    st = _np_state(sd)
    for i in range(3):
        want = oracle.segment(imgs[i], st, 128, 6, n_segments=600, seed=i)
        r = res[i]
        assert np.array_equal(r.segments, want["segments"])                       # label map: bit-exact
        n0, n1, e0, e1 = g.node_ptr_host[i], g.node_ptr_host[i + 1], g.edge_ptr_host[i], g.edge_ptr_host[i + 1]
        ei = np.stack([g.edge_src[e0:e1].cpu().numpy(), g.edge_dst[e0:e1].cpu().numpy()]).astype(np.int64) - n0
        assert np.array_equal(ei, want["graph"]["edge_index"])                    # edge_index: integer-exact
        # node input and probabilities: the oracle sums in the kernels' order with the shared exp / GELU sequences
        # (include/ggc_fmath.h), so they are identical, not merely within the north star's 1e-4 ...
        assert np.array_equal(g.x[n0:n1].cpu().numpy(), want["x"])
        assert np.array_equal(out["probs"][n0:n1].cpu().numpy(), want["probs"])
        # ... and so are the trimap and the mask, pixel for pixel (north star: bit-exact integer outputs)
        assert np.array_equal(r.trimap, want["trimap"])
        assert np.array_equal(r.binary_mask, want["binary_mask"])
        assert r.overlay.shape == (300, 400, 3) and r.rgba.shape == (300, 400, 4)
        assert set(np.unique(r.binary_mask)) <= {0, 1} and set(np.unique(r.trimap)) <= {0, 1, 2, 3}


@pytest.mark.parametrize("b,h,w,n_seg", [(3, 75, 101, 60), (40, 64, 96, 50), (1, 130, 67, 90)])
def test_full_pipeline_matches_oracle_on_odd_shapes(oracle, b, h, w, n_seg):
    """Widths / pixel counts that are not multiples of 64, one image, and a batch large enough for the GrabCut lanes
    (b >= 32): every tile-edge, run-boundary and lane-split path against the oracle."""
    from gcn_grabcut import GCNGrabCutPipeline, SuperpixelGraphConfig
    from gcn_grabcut.synthetic import synthetic_batch
    model, sd = seeded_state_dict(32, 2, seed=5)
    pipe = GCNGrabCutPipeline(model.eval(), sp_config=SuperpixelGraphConfig(n_segments=n_seg), device="cuda")
    imgs = synthetic_batch(b, h, w, config_id=9)
    out = pipe.segment_batch_device(pipe._eng.to_device(imgs))
    seg, tri, binm = out["segments"].cpu().numpy(), out["trimap"].cpu().numpy(), out["binary_mask"].cpu().numpy()
    st = _np_state(sd)
    check = range(b) if b <= 4 else (0, 9, 10, 19, 20, 29, 30, b - 1)            # lane boundaries of a 40-image batch
    for i in check:
        want = oracle.segment(imgs[i], st, 32, 2, n_segments=n_seg, seed=i)
        assert np.array_equal(seg[i], want["segments"]), i
        assert np.array_equal(tri[i], want["trimap"]), i
        assert np.array_equal(binm[i], want["binary_mask"]), i


def test_segment_returns_result_like_reference():
    # reference tests/test.py:435-448
    from gcn_grabcut.model import ResGCNNet
    from gcn_grabcut.pipeline import GCNGrabCutPipeline
    img = _img(100, 100)
    pipeline = GCNGrabCutPipeline(ResGCNNet(hidden_channels=32, n_layers=2).eval(), device="cuda")
    result = pipeline.segment(img)
    assert result.binary_mask.shape == (100, 100) and result.trimap.shape == (100, 100)
    assert result.overlay.shape == (100, 100, 3) and result.rgba.shape == (100, 100, 4)
    for key in ("graph_build", "data_prep", "gcn_inference", "grabcut", "postprocess"):
        assert key in result.timing
    m, tm = result.evaluate_against((np.indices((100, 100)).sum(0) % 2).astype(np.uint8))
    assert 0 <= m.iou <= 1 and 0 <= tm.unknown_fraction <= 1
    r2 = pipeline.segment(img, edge_aware=False, keep_largest=True, refine_iters=1)
    assert r2.binary_mask.shape == (100, 100)
    rb = pipeline.segment_bbox(img, (10, 10, 80, 80))
    assert rb.binary_mask.shape == (100, 100) and (rb.trimap[30:60, 30:60] == 1).all()


def test_graph_builder_like_reference(oracle):
    # reference tests/test.py:87-155
    from gcn_grabcut.graph_builder import GraphBuilder, SuperpixelGraphConfig, N_EDGE_FEATS, N_NODE_FEATS, encode_user_hints
    img = _img(64, 64)
    graph = GraphBuilder(img, SuperpixelGraphConfig(n_segments=50)).build()
    assert graph.n_nodes > 0 and graph.node_features.shape == (graph.n_nodes, 16)
    assert graph.edge_index.shape[0] == 2 and graph.n_edges == graph.edge_index.shape[1]
    assert graph.edge_index.dtype == np.int64 and graph.segments.dtype == np.int32
    graph = GraphBuilder(img).build()
    assert graph.node_features[:, :6].min() >= -0.01 and graph.node_features[:, :6].max() <= 1.01
    assert graph.edge_attr.shape == (graph.n_edges, N_EDGE_FEATS)
    assert np.unique(graph.segments).min() == 0 and np.unique(graph.segments).max() == graph.n_nodes - 1
    prior = graph.prior_features
    assert prior.shape == (graph.n_nodes, 3) and np.isfinite(prior).all() and prior.min() >= -1e-5 and prior.max() <= 1 + 1e-5
    assert graph.node_input().shape == (graph.n_nodes, N_NODE_FEATS)
    hints = encode_user_hints(graph.segments, [(32, 32)], [(2, 2)])
    assert hints[int(graph.segments[32, 32]), 0] == 1.0 and hints[int(graph.segments[2, 2]), 1] == 1.0
    for conn in (4, 8):
        assert GraphBuilder(img, SuperpixelGraphConfig(n_segments=50, connectivity=conn)).build().n_edges > 0
    # the whole builder equals the oracle chain
    lab, hsv, gray, grad = oracle.preprocess(img)
    seg, n = oracle.slic(lab, 300)
    want = oracle.graph_build(seg, lab, hsv, grad)
    assert np.array_equal(graph.segments, seg) and np.array_equal(graph.edge_index, want["edge_index"])
    from gcn_grabcut.graph_builder import compute_auto_prior
    assert np.array_equal(compute_auto_prior(seg, lab), want["prior"])
    assert np.array_equal(compute_auto_prior(seg, lab, 0.30, 0.50), oracle.auto_prior(seg, lab, 0.30, 0.50))     # non-default sigmas


def test_grabcut_class_like_reference(oracle):
    # reference tests/test.py:31-82
    from gcn_grabcut.grabcut import GrabCut, GrabCutConfig
    img = _img(100, 100)
    gc = GrabCut(img, GrabCutConfig(n_iter=1))
    mask = gc.run_with_bbox((10, 10, 80, 80))
    assert mask.shape == (100, 100) and set(np.unique(mask)) <= {0, 1}
    assert len(gc.history) == 1 and 0 <= gc.history[0].fg_ratio <= 1
    assert gc.overlay_mask().shape == (100, 100, 3) and gc.crop_foreground().shape == (100, 100, 4)
    tri = np.full((100, 100), 2, np.uint8); tri[30:70, 30:70] = 3
    gc2 = GrabCut(img, GrabCutConfig(n_iter=1))
    m2 = gc2.run_with_trimap(tri)
    wb, *_ = oracle.grabcut(img, tri, 1, 0)
    assert np.array_equal(m2, wb)
    m3 = gc2.refine(1)
    assert m3.shape == (100, 100) and len(gc2.history) == 2
    with pytest.raises(ValueError):
        gc2.run_with_trimap(np.zeros((10, 10), np.uint8))
    with pytest.raises(RuntimeError):
        GrabCut(img).refine(1)
    with pytest.raises(ValueError):
        GrabCut(img, GrabCutConfig(color_space="xyz"))


@pytest.mark.parametrize("cs", ["rgb", "hsv", "lab"])
def test_grabcut_color_spaces_like_reference(oracle, cs):
    # reference tests/test.py:52-57, plus: the device conversion equals the oracle's byte for byte and GrabCut on the
    # converted image equals the oracle's GrabCut on the oracle's conversion
    from gcn_grabcut import GCNGrabCutPipeline, ResGCNNet
    from gcn_grabcut._engine import get_engine
    from gcn_grabcut.grabcut import GrabCut, GrabCutConfig
    from gcn_grabcut.synthetic import synthetic_image
    img = synthetic_image(96, 128, 321)
    gc = GrabCut(img, GrabCutConfig(n_iter=2, color_space=cs))
    m = gc.run_with_bbox((10, 10, 100, 70))
    assert m.shape == (96, 128) and set(np.unique(m)) <= {0, 1}
    conv = img if cs == "rgb" else oracle.convert_color8(img, cs)
    if cs != "rgb":
        eng = get_engine("cuda")
        assert np.array_equal(eng.convert_color8(eng.to_device(img[None]), cs)[0].cpu().numpy(), conv)
    wb, *_ = oracle.grabcut(conv, None, n_iter=2, mode=1, rect=(10, 10, 100, 70))
    assert np.array_equal(m, wb)
    pipe = GCNGrabCutPipeline(ResGCNNet(hidden_channels=32, n_layers=2).eval(), gc_config=GrabCutConfig(color_space=cs), device="cuda")
    assert set(np.unique(pipe.segment(img).binary_mask)) <= {0, 1}


def test_helper_functions_match_oracle(oracle):
    from gcn_grabcut import guided_filter, refine_trimap, clean_mask, evaluate
    rng = np.random.default_rng(0)
    guide = rng.random((40, 56)).astype(np.float32)
    src = rng.random((40, 56)).astype(np.float32)
    assert np.array_equal(guided_filter(guide, src, 4, 1e-3), oracle.guided_filter(guide, src, 4, 1e-3))
    seg = (np.arange(40 * 56).reshape(40, 56) // 280).astype(np.int32)
    probs = rng.dirichlet([1, 1, 1], size=8).astype(np.float32)
    img = _img(40, 56)
    assert np.array_equal(refine_trimap(probs, seg, img), oracle.refine_trimap(probs, seg, img))
    m = (rng.random((40, 56)) < 0.1).astype(np.uint8); m[5:20, 5:30] = 1
    assert np.array_equal(clean_mask(m, 0.01), oracle.clean_mask(m, 0.01))
    ev = evaluate(m, np.roll(m, 2, 1))
    assert ev.iou == pytest.approx(oracle.iou(m, np.roll(m, 2, 1)), abs=1e-9)
    assert evaluate(m, m).iou == pytest.approx(1.0, abs=1e-4) and evaluate(np.zeros_like(m), m).iou < 0.01


def test_one_sided_trimap_is_reseeded_and_cpu_device_is_refused():
    from gcn_grabcut.model import ResGCNNet
    from gcn_grabcut.pipeline import GCNGrabCutPipeline
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        GCNGrabCutPipeline(ResGCNNet(hidden_channels=32, n_layers=2), device="cpu")
    with torch.no_grad():
        model = ResGCNNet(hidden_channels=32, n_layers=2).eval()
        model.head.bias.copy_(torch.tensor([-20.0, -20.0, 20.0]))     # everything "foreground"
    res = GCNGrabCutPipeline(model, device="cuda").segment(_img(80, 80))
    assert set(np.unique(res.trimap)) & {0, 2}                        # _seed_from_prior put background back


def test_pipeline_replicas_run_concurrently_with_identical_results(pipe128):
    """replica(): private library contexts, own HIP streams and host threads — two different batches segmented at the
    same time give exactly what each gives alone (scratch arenas, weights and streams are per replica)."""
    import threading
    from gcn_grabcut.synthetic import synthetic_batch
    pipe, _ = pipe128
    dev = pipe._eng.device
    batches = [pipe._eng.to_device(synthetic_batch(6, 150, 200, config_id=3, first_index=10 * i)) for i in range(3)]
    keys = ("segments", "trimap", "binary_mask", "gc_mask")
    alone = [{k: pipe.segment_batch_device(b)[k].clone() for k in keys} for b in batches]
    torch.cuda.synchronize(dev)
    pipes = [pipe, pipe.replica(), pipe.replica()]
    streams = [torch.cuda.Stream(dev) for _ in pipes]
    got, errors = [None] * 3, []

    def worker(i):
        try:
            torch.cuda.set_device(dev)
            with torch.cuda.stream(streams[i]):
                for _ in range(3):                                   # a few rounds each, so the stages really interleave
                    got[i] = pipes[i].segment_batch_device(batches[i])
            streams[i].synchronize()
        except Exception as exc:                                     # surfaced in the main thread
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(3):
        for k in keys:
            assert torch.equal(got[i][k], alone[i][k]), (i, k)
    # the packaged form: results in input order, whatever pipeline took which batch
    outs = pipe.segment_batches_overlapped(batches + batches[:2], n_pipelines=3)
    assert len(outs) == 5
    for j, o in enumerate(outs):
        for k in keys:
            assert torch.equal(o[k], alone[j % 3][k]), (j, k)
    assert pipe.grabcut_lanes == 4                                   # restored


def _check_against_oracle(oracle, pipe, sd, imgs, hidden, layers, n_seg):
    """label map, edge_index, node input, probabilities, trimap and mask of a device batch against the oracle chain"""
    out = pipe.segment_batch_device(pipe._eng.to_device(imgs))
    g = out["graphs"]
    seg, tri, binm = out["segments"].cpu().numpy(), out["trimap"].cpu().numpy(), out["binary_mask"].cpu().numpy()
    st = _np_state(sd)
    for i in range(len(imgs)):
        want = oracle.segment(imgs[i], st, hidden, layers, n_segments=n_seg, seed=i)
        assert np.array_equal(seg[i], want["segments"]), i                        # label map: bit-exact
        n0, n1, e0, e1 = g.node_ptr_host[i], g.node_ptr_host[i + 1], g.edge_ptr_host[i], g.edge_ptr_host[i + 1]
        ei = np.stack([g.edge_src[e0:e1].cpu().numpy(), g.edge_dst[e0:e1].cpu().numpy()]).astype(np.int64) - n0
        assert np.array_equal(ei, want["graph"]["edge_index"]), i                 # edge_index: integer-exact, reference order
        assert np.array_equal(g.x[n0:n1].cpu().numpy(), want["x"]), i
        assert np.array_equal(out["probs"][n0:n1].cpu().numpy(), want["probs"]), i
        assert np.array_equal(tri[i], want["trimap"]), i                          # trimap and mask: bit-exact end to end
        assert np.array_equal(binm[i], want["binary_mask"]), i
    return out


def test_config0_single_320x240_500_superpixels(oracle):
    """BASELINE.json configs[0]: one 320x240 image, ~500 superpixels, through the reference's single-image entry point
    (reference inference.py:118-141 -> pipeline.py:265-352)."""
    from gcn_grabcut import GCNGrabCutPipeline, SuperpixelGraphConfig
    from gcn_grabcut.synthetic import synthetic_batch
    model, sd = seeded_state_dict(128, 6, seed=0)
    pipe = GCNGrabCutPipeline(model.eval(), sp_config=SuperpixelGraphConfig(n_segments=500), device="cuda")
    img = synthetic_batch(1, 240, 320, config_id=1)[0]
    r = pipe.segment(img)
    want = oracle.segment(img, _np_state(sd), 128, 6, n_segments=500, seed=0)
    assert 480 <= want["graph"]["n_nodes"] <= 560                                  # N ~ 532
    assert np.array_equal(r.segments, want["segments"])
    assert np.array_equal(r.trimap, want["trimap"])
    assert np.array_equal(r.binary_mask, want["binary_mask"])
    assert r.overlay.shape == (240, 320, 3) and r.rgba.shape == (240, 320, 4)
    for key in ("graph_build", "data_prep", "gcn_inference", "grabcut", "postprocess"):
        assert key in r.timing
    _check_against_oracle(oracle, pipe, sd, img[None], 128, 6, 500)


def test_config4_1080p_4000_superpixels(oracle):
    """BASELINE.json configs[4]'s per-image shape: 1920x1080, ~4000 superpixels (N ~ 3.8k: the > 782-node aggregation route,
    14.7 M kNN distances, 8.3 M n-links per image), one image and a batch of two against the oracle."""
    from gcn_grabcut import GCNGrabCutPipeline, SuperpixelGraphConfig
    from gcn_grabcut.synthetic import synthetic_batch
    model, sd = seeded_state_dict(128, 6, seed=0)
    pipe = GCNGrabCutPipeline(model.eval(), sp_config=SuperpixelGraphConfig(n_segments=4000), device="cuda")
    imgs = synthetic_batch(2, 1080, 1920, config_id=5)
    out = _check_against_oracle(oracle, pipe, sd, imgs, 128, 6, 4000)
    n = np.diff(out["graphs"].node_ptr_host)
    assert (n > 3500).all() and (n < 4100).all()                                   # N ~ 3833
    one = pipe.segment_batch_device(pipe._eng.to_device(imgs[1:]))                 # batch of one == its row of the batch of two
    assert torch.equal(one["segments"][0], out["segments"][1]) and torch.equal(one["trimap"][0], out["trimap"][1])
    assert torch.equal(one["binary_mask"][0], out["binary_mask"][1])


def test_software_pipeline_gives_the_one_chunk_outputs(oracle, gpu_ctx):
    """segment_batch_device(chunks=n): the front stages of chunk k+1 run under the GrabCut of chunk k.  Images are
    independent and image b keeps seed + b, so every output — label maps, packed graphs, probabilities, trimaps, masks,
    overlays — equals the one-chunk run bit for bit, for equal and for unequal chunks."""
    import torch
    from helpers import seeded_state_dict
    from gcn_grabcut import GCNGrabCutPipeline, SuperpixelGraphConfig
    from gcn_grabcut.synthetic import synthetic_batch
    model, _ = seeded_state_dict(64, 3, seed=2)
    pipe = GCNGrabCutPipeline(model.eval(), sp_config=SuperpixelGraphConfig(n_segments=120), device="cuda:0")
    bgr = torch.from_numpy(synthetic_batch(22, 96, 128, config_id=5)).cuda()
    ref = pipe.segment_batch_device(bgr)
    assert pipe.chunk_plan(22, 3, 1.0) == [(0, 7), (7, 15), (15, 22)] and pipe.chunk_plan(5, 9, 0.8)[-1][1] == 5
    for chunks, ratio in ((3, 1.0), (4, 0.6), (2, 0.8)):
        pipe.chunk_ratio = ratio
        timing = {}
        out = pipe.segment_batch_device(bgr, chunks=chunks, timing=timing if chunks == 3 else None)
        for k in ("segments", "trimap", "gc_mask", "binary_mask", "probs", "overlay", "rgba"):
            assert torch.equal(out[k], ref[k]), (chunks, k)
        g, r = out["graphs"], ref["graphs"]
        assert torch.equal(g.x, r.x) and torch.equal(g.edge_src, r.edge_src) and torch.equal(g.edge_dst, r.edge_dst)
        assert torch.equal(g.edge_attr, r.edge_attr) and torch.equal(g.node_ptr, r.node_ptr)
        assert (g.node_ptr_host == r.node_ptr_host).all() and (g.edge_ptr_host == r.edge_ptr_host).all()
        if chunks == 3:
            assert {"graph_build", "grabcut", "postprocess", "wall"} <= set(timing) and timing["grabcut"] > 0


def test_use_lab_false_runs_the_float64_slic_end_to_end(oracle, gpu_ctx):
    """SuperpixelGraphConfig(use_lab=False) (reference graph_builder.py:177-179): GraphBuilder, the pipeline and the graph-cache
    writer take skimage's float64 SLIC on the RGB image; everything downstream is unchanged.  Label map, trimap and mask equal
    the oracle's."""
    import torch
    from helpers import seeded_state_dict
    from gcn_grabcut import GCNGrabCutPipeline, GraphBuilder, SuperpixelGraphConfig
    from gcn_grabcut.synthetic import synthetic_batch
    model, sd = seeded_state_dict(64, 3, seed=4)
    st = {k: v.numpy() for k, v in sd.items() if v.dtype.is_floating_point}
    cfg = SuperpixelGraphConfig(n_segments=150, use_lab=False)
    pipe = GCNGrabCutPipeline(model.eval(), sp_config=cfg, device="cuda:0")
    imgs = synthetic_batch(2, 120, 160, config_id=6)
    res = pipe.segment_batch(list(imgs))
    for i, r in enumerate(res):
        want = oracle.segment(imgs[i], st, 64, 3, n_segments=150, seed=i, use_lab=False)
        assert np.array_equal(r.segments, want["segments"]) and np.array_equal(r.trimap, want["trimap"])
        assert np.array_equal(r.binary_mask, want["binary_mask"])
    lab_path = GCNGrabCutPipeline(model, sp_config=SuperpixelGraphConfig(n_segments=150), device="cuda:0").segment_batch(list(imgs))
    assert not np.array_equal(lab_path[0].segments, res[0].segments)       # it IS another segmentation than the Lab path's
    g = GraphBuilder(imgs[0], cfg).build()
    assert np.array_equal(g.segments, res[0].segments)
