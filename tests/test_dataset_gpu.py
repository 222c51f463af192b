"""GPU side of the graph-cache writer: batched prepare_dataset equals the per-image GraphBuilder (itself bit-exact
against the oracle, test_graph_gpu.py) and the reference's label formula; the cache round-trips."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _samples(n, h, w, seed0):
    from gcn_grabcut.synthetic import synthetic_image
    out = []
    for i in range(n):
        img, gt = synthetic_image(h, w, seed0 + i, return_mask=True)
        out.append({"image": np.ascontiguousarray(img), "gt_mask": np.ascontiguousarray(gt.astype(np.uint8)), "name": f"s{i}"})
    return out


def test_prepare_dataset_matches_graph_builder_and_label_formula(tmp_path, gpu_ctx):
    from gcn_grabcut import dataset as ds
    from gcn_grabcut.graph_builder import GraphBuilder, SuperpixelGraphConfig
    cfg = SuperpixelGraphConfig(n_segments=120)
    samples = _samples(3, 96, 128, 500) + _samples(2, 80, 100, 600)        # two shapes -> two device batches
    recs = ds.prepare_dataset(samples, cfg, 0.70, 0.70, cache_dir=tmp_path / "cache", workers=2, keep_segments=True)
    assert len(recs) == 5
    for s, (data, labels, seg) in zip(samples, recs):
        g = GraphBuilder(s["image"], cfg).build()
        assert np.array_equal(seg, g.segments)
        assert np.array_equal(data.x.numpy(), g.node_input())
        assert np.array_equal(data.edge_index.numpy(), g.edge_index) and data.edge_index.dtype == torch.int64
        assert np.array_equal(data.edge_attr.numpy(), g.edge_attr)
        assert np.array_equal(data.node_area.numpy(), g.node_areas)
        # reference dataset.py:245-250
        flat = seg.ravel()
        counts = np.bincount(flat, minlength=g.n_nodes).astype(np.float32)
        fg_ratio = (np.bincount(flat, weights=(s["gt_mask"].ravel() > 0).astype(np.float64), minlength=g.n_nodes)
                    / np.maximum(counts, 1.0)).astype(np.float32)
        assert np.array_equal(data.fg_ratio.numpy(), fg_ratio)
        assert np.array_equal(labels.numpy(), ds.derive_trimap_labels(seg, s["gt_mask"], 0.70, 0.70))
        assert labels.dtype == torch.int64 and set(np.unique(labels.numpy())) <= {0, 1, 2}
    files = sorted((tmp_path / "cache").glob("*.pt"))
    assert {f.stem for f in files} == {ds._cache_key(s, cfg, 0.70, 0.70) for s in samples}        # the reference's file names
    # second run: everything comes from the cache (readable without executing anything from the file)
    blob = torch.load(files[0], map_location="cpu", weights_only=True)
    assert blob["format"] == ds.CACHE_FORMAT
    again = ds.prepare_dataset(samples, cfg, 0.70, 0.70, cache_dir=tmp_path / "cache", keep_segments=False)
    for (d1, l1, _), (d2, l2, s2) in zip(recs, again):
        assert s2 is None and torch.equal(d1.x, d2.x) and torch.equal(d1.edge_index, d2.edge_index) and torch.equal(l1, l2)
    one = ds.prepare_sample(samples[3], cfg, 0.70, 0.70)
    assert torch.equal(one[0].x, recs[3][0].x) and torch.equal(one[1], recs[3][1])


def test_prepare_graphs_tool_end_to_end(tmp_path):
    """The CLI with the reference's flags: PNG pairs on disk -> one cache entry per pair, second run hits the cache."""
    import os, subprocess, sys
    from pathlib import Path
    from PIL import Image
    from gcn_grabcut import dataset as ds
    from gcn_grabcut.graph_builder import SuperpixelGraphConfig
    from gcn_grabcut.synthetic import synthetic_image
    (tmp_path / "im").mkdir(); (tmp_path / "mk").mkdir()
    for i in range(5):
        img, gt = synthetic_image(120, 160, 900 + i, return_mask=True)
        Image.fromarray(np.ascontiguousarray(img[:, :, ::-1])).save(tmp_path / "im" / f"p{i}.png")
        Image.fromarray((gt * 255).astype(np.uint8)).save(tmp_path / "mk" / f"p{i}.png")
    root = Path(__file__).resolve().parents[1]
    cmd = [sys.executable, str(root / "tools" / "prepare_graphs.py"), "--images", str(tmp_path / "im"), "--masks",
           str(tmp_path / "mk"), "--cache", str(tmp_path / "cache"), "--workers", "2", "--max-size", "384",
           "--superpixels", "100", "--limit", "4"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "4 graphs (0 from cache, 0 dropped)" in out.stdout
    files = sorted((tmp_path / "cache").glob("*.pt"))
    assert len(files) == 4
    samples = ds.list_image_mask_pairs(tmp_path / "im", tmp_path / "mk", max_size=384)[:4]
    recs = ds.prepare_dataset(samples, SuperpixelGraphConfig(n_segments=100), cache_dir=tmp_path / "cache", keep_segments=False)
    assert len(recs) == 4 and all(r[0].x.shape[1] == 19 and r[2] is None for r in recs)
    img0 = ds.materialise(samples[0])
    one = ds.prepare_sample(img0, SuperpixelGraphConfig(n_segments=100))
    assert torch.equal(one[0].x, recs[0][0].x) and torch.equal(one[0].edge_index, recs[0][0].edge_index)
