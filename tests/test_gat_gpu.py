"""GPU parity of GATTrimapNet (`--model gat`, reference model.py:323-414) through the C ABI: logits and probabilities
IDENTICAL to the CPU oracle's (same summation order, shared exp / GELU / sigmoid sequences), batched == single, and the
pipeline end to end."""
import numpy as np
import pytest
import torch

from helpers import superpixel_like_graph
from test_gat_oracle import seeded_gat

pytestmark = pytest.mark.gpu


def _data(x, ei, ea, **kw):
    from gcn_grabcut.data import Data
    return Data(x=torch.as_tensor(x), edge_index=torch.as_tensor(ei), edge_attr=torch.as_tensor(ea), **kw).to("cuda")


def _st(sd):
    return {k: v.numpy() for k, v in sd.items() if v.dtype.is_floating_point}


@pytest.mark.parametrize("hidden,layers,n", [(32, 2, 80), (64, 3, 257), (128, 5, 601)])
def test_forward_matches_oracle(oracle, gpu_ctx, hidden, layers, n):
    m, sd = seeded_gat(hidden, layers, seed=hidden + layers)
    m = m.to("cuda").eval()
    x, ei, ea = superpixel_like_graph(n=n, seed=n)
    want, want_p = oracle.gat_forward(_st(sd), hidden, layers, x, ei, ea)
    d = _data(x, ei, ea)
    got = m(d).cpu().numpy()
    assert got.shape == (n, 3) and np.array_equal(got, want)
    assert np.array_equal(m.predict_probs(d), want_p)


@pytest.mark.parametrize("hidden,heads", [(32, 1), (32, 4), (64, 2), (64, 1), (128, 4), (128, 2), (128, 1)])
def test_other_head_counts_match_oracle(oracle, gpu_ctx, hidden, heads):
    m, sd = seeded_gat(hidden, 2, seed=hidden + heads, heads=heads)
    m = m.to("cuda").eval()
    x, ei, ea = superpixel_like_graph(n=257, seed=heads)
    want, want_p = oracle.gat_forward(_st(sd), hidden, 2, x, ei, ea, heads=heads)
    d = _data(x, ei, ea)
    assert np.array_equal(m(d).cpu().numpy(), want) and np.array_equal(m.predict_probs(d), want_p)


def test_batched_equals_single_and_oracle(oracle, gpu_ctx):
    from gcn_grabcut.data import Batch
    m, sd = seeded_gat(128, 5, seed=7)
    m = m.to("cuda").eval()
    graphs = [superpixel_like_graph(n=n, seed=n) for n in (590, 37, 615)]
    datas = [_data(*g) for g in graphs]
    one = torch.cat([m(d) for d in datas]).cpu().numpy()
    both = m(Batch.from_data_list(datas)).cpu().numpy()
    assert np.array_equal(one, both)                       # reference tests/test.py:294-306 asks for 1e-4
    off = np.cumsum([0] + [g[0].shape[0] for g in graphs])
    want, _ = oracle.gat_forward(_st(sd), 128, 5, np.concatenate([g[0] for g in graphs]),
                                 np.concatenate([g[1] + off[i] for i, g in enumerate(graphs)], 1),
                                 np.concatenate([g[2] for g in graphs]),
                                 np.concatenate([np.full(g[0].shape[0], i) for i, g in enumerate(graphs)]))
    assert np.array_equal(both, want)


def test_isolated_node_and_reference_shapes(oracle, gpu_ctx):
    from gcn_grabcut.model import build_model
    from helpers import chain_graph
    m, sd = seeded_gat(32, 2, seed=1)
    m = m.to("cuda").eval()
    x, ei, ea = chain_graph(10, seed=4)
    keep = ei[1] != 9
    ei, ea = ei[:, keep], ea[keep]
    want, _ = oracle.gat_forward(_st(sd), 32, 2, x.numpy(), ei.numpy(), ea.numpy())
    assert np.array_equal(m(_data(x, ei, ea)).cpu().numpy(), want)
    g = build_model("gat", hidden_channels=32, n_layers=2).to("cuda").eval()      # reference tests/test.py:274-280
    assert g(_data(*chain_graph(80, seed=1))).shape == (80, 3)
    m.train()
    with pytest.raises(RuntimeError):
        m(_data(x, ei, ea))


def test_pipeline_with_gat(oracle, gpu_ctx):
    """`--model gat` through the whole path: the trimap and the mask follow from the oracle's GAT probabilities"""
    from gcn_grabcut import GCNGrabCutPipeline, SuperpixelGraphConfig
    from gcn_grabcut.synthetic import synthetic_batch
    m, sd = seeded_gat(64, 2, seed=3)
    pipe = GCNGrabCutPipeline(m.to("cuda").eval(), sp_config=SuperpixelGraphConfig(n_segments=120), device="cuda")
    imgs = synthetic_batch(2, 96, 128, config_id=8)
    out = pipe.segment_batch_device(pipe._eng.to_device(imgs))
    g = out["graphs"]
    for i in range(2):
        n0, n1, e0, e1 = g.node_ptr_host[i], g.node_ptr_host[i + 1], g.edge_ptr_host[i], g.edge_ptr_host[i + 1]
        x = g.x[n0:n1].cpu().numpy()
        ei = np.stack([g.edge_src[e0:e1].cpu().numpy(), g.edge_dst[e0:e1].cpu().numpy()]).astype(np.int64) - n0
        _, probs = oracle.gat_forward(_st(sd), 64, 2, x, ei, g.edge_attr[e0:e1].cpu().numpy())
        assert np.array_equal(out["probs"][n0:n1].cpu().numpy(), probs)
        seg = out["segments"][i].cpu().numpy()
        tri = oracle.refine_trimap(probs, seg, imgs[i])
        tri = oracle.seed_from_prior(tri, x[:, 16:19], seg, 0.1)        # pipeline.py:300-304: no definite seed -> prior
        assert np.array_equal(out["trimap"][i].cpu().numpy(), tri)
        binary, *_ = oracle.grabcut(imgs[i], tri, 5, 0, None, i)
        assert np.array_equal(out["binary_mask"][i].cpu().numpy(), oracle.clean_mask(binary, 0.002, False))


def test_input_self_loops_are_refused(gpu_ctx):
    """PyG's GATv2Conv removes i -> i edges before it adds its own mean-filled loops; this build keeps every input edge, so
    it refuses such input (the graph builder never produces one) instead of answering differently from the reference."""
    m, _ = seeded_gat(32, 2, seed=1)
    m = m.to("cuda").eval()
    x, ei, ea = superpixel_like_graph(n=40, seed=3)
    ei = np.concatenate([ei, np.array([[5], [5]], ei.dtype)], 1)
    ea = np.concatenate([ea, ea[:1]], 0)
    with pytest.raises(ValueError, match="self-loops"):
        m(_data(x, ei, ea))
