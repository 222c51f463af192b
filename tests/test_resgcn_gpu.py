"""GPU parity: ResGCNNet forward and the GCNConv scatter-gather through the
C ABI of libggc_hip.so against the CPU oracle.  The north star asks for 1e-4 on logits; the oracle sums in the kernels'
order (fma chains of the matrix pipe, butterflies of the LayerNorm statistics) with the shared exp / GELU sequences of
include/ggc_fmath.h, so logits and probabilities are compared for EQUALITY."""
import numpy as np
import pytest
import torch

from helpers import chain_graph, superpixel_like_graph, seeded_state_dict

pytestmark = pytest.mark.gpu


def _np_state(sd):
    return {k: v.numpy() for k, v in sd.items() if v.dtype.is_floating_point}


def _data(x, ei, ea, **kw):
    from gcn_grabcut.data import Data
    return Data(x=torch.as_tensor(x), edge_index=torch.as_tensor(ei), edge_attr=torch.as_tensor(ea), **kw).to("cuda")


@pytest.mark.parametrize("hidden,layers,n", [(32, 2, 80), (64, 2, 257), (96, 3, 300), (128, 6, 601)])
def test_forward_matches_oracle(oracle, gpu_ctx, hidden, layers, n):
    model, sd = seeded_state_dict(hidden, layers, seed=hidden + layers)
    model = model.to("cuda").eval()
    x, ei, ea = superpixel_like_graph(n=n, seed=n)
    want, want_p = oracle.resgcn_forward(_np_state(sd), hidden, layers, x, ei, ea)
    d = _data(x, ei, ea)
    got = model(d).cpu().numpy()
    assert got.shape == (n, 3)
    assert np.array_equal(got, want)
    probs = model.predict_probs(d)
    assert np.array_equal(probs, want_p)
    assert np.allclose(probs.sum(1), 1.0, atol=1e-6)


@pytest.mark.parametrize("hidden,layers,n", [(16, 2, 90), (48, 2, 257), (100, 3, 300), (33, 2, 64), (35, 2, 130), (9, 1, 40)])
def test_widths_that_are_not_a_multiple_of_32_run_zero_padded(oracle, gpu_ctx, hidden, layers, n):
    """reference model.py:449-455 takes any width.  The library pads the width to the next multiple of 32 with zeros and keeps
    the LayerNorm statistics on the true width; the padded channels are exact zeros, so the result is the unpadded model's.
    The oracle runs at the TRUE width with its sums split where the padded kernels split them: equal bit for bit."""
    model, sd = seeded_state_dict(hidden, layers, seed=3 * hidden + layers)
    model = model.to("cuda").eval()
    assert sum(p.numel() for p in model.parameters()) == sum(v.numel() for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k)
    x, ei, ea = superpixel_like_graph(n=n, seed=n + 1)
    want, want_p = oracle.resgcn_forward(_np_state(sd), hidden, layers, x, ei, ea)
    d = _data(x, ei, ea)
    got = model(d).cpu().numpy()
    assert got.shape == (n, 3) and np.isfinite(got).all()
    assert np.array_equal(got, want), np.abs(got - want).max()       # the oracle splits its sums at half the PADDED width too
    assert np.array_equal(model.predict_probs(d), want_p)
    # batched == single, as for the native widths
    from gcn_grabcut.data import Batch, Data
    x2, ei2, ea2 = superpixel_like_graph(n=70, seed=5)
    b = Batch.from_data_list([Data(x=torch.as_tensor(x), edge_index=torch.as_tensor(ei), edge_attr=torch.as_tensor(ea)),
                              Data(x=torch.as_tensor(x2), edge_index=torch.as_tensor(ei2), edge_attr=torch.as_tensor(ea2))]).to("cuda")
    both = model(b).cpu().numpy()
    assert np.array_equal(both[:n], got)


def test_two_models_alternate_on_one_context(gpu_ctx):
    """The library context holds ONE set of ResGCN weights: a model must notice that another one replaced its weights
    there (the record of what is resident lives on the context, not on the model)."""
    a, _ = seeded_state_dict(64, 2, seed=11)
    b, _ = seeded_state_dict(32, 3, seed=12)
    a, b = a.to("cuda").eval(), b.to("cuda").eval()
    x, ei, ea = superpixel_like_graph(n=150, seed=4)
    d = _data(x, ei, ea)
    first_a, first_b = a(d).clone(), b(d).clone()
    assert torch.equal(a(d), first_a)            # a again after b used the same context
    assert torch.equal(b(d), first_b)
    assert not torch.equal(first_a, first_b)


def test_batched_equals_single_and_oracle(oracle, gpu_ctx):
    # reference tests/test.py:294-306 (atol 1e-4), here on the HIP path
    from gcn_grabcut.data import Batch
    model, sd = seeded_state_dict(128, 6, seed=7)
    model = model.to("cuda").eval()
    graphs = [superpixel_like_graph(n=n, seed=n) for n in (590, 601, 37, 615)]
    datas = [_data(*g) for g in graphs]
    one = torch.cat([model(d) for d in datas]).cpu().numpy()
    both = model(Batch.from_data_list(datas)).cpu().numpy()
    assert np.array_equal(one, both)                       # reference tests/test.py:294-306 asks for 1e-4; here: identical
    off = np.cumsum([0] + [g[0].shape[0] for g in graphs])
    want, _ = oracle.resgcn_forward(
        _np_state(sd), 128, 6, np.concatenate([g[0] for g in graphs]),
        np.concatenate([g[1] + off[i] for i, g in enumerate(graphs)], 1),
        np.concatenate([g[2] for g in graphs]),
        np.concatenate([np.full(g[0].shape[0], i) for i, g in enumerate(graphs)]))
    assert np.array_equal(both, want)


def test_reference_chain_graph_shapes(gpu_ctx):
    # reference tests/test.py:274-292
    from gcn_grabcut.model import build_model
    model = build_model("resgcn", hidden_channels=32, n_layers=2).to("cuda").eval()
    o1 = model(_data(*chain_graph(80, seed=1)))
    o2 = model(_data(*chain_graph(80, seed=2)))
    assert o1.shape == (80, 3) and not torch.allclose(o1, o2)


def test_isolated_node(oracle, gpu_ctx):
    model, sd = seeded_state_dict(32, 2, seed=1)
    model = model.to("cuda").eval()
    x, ei, ea = chain_graph(10, seed=4)
    keep = ei[1] != 9
    ei, ea = ei[:, keep], ea[keep]
    want, _ = oracle.resgcn_forward(_np_state(sd), 32, 2, x.numpy(), ei.numpy(), ea.numpy())
    got = model(_data(x, ei, ea)).cpu().numpy()
    assert np.isfinite(got).all() and np.array_equal(got, want)


def _hub_graph(n, hub_deg, seed):
    """superpixel-like graph plus a hub: node 3 receives hub_deg extra mirrored edges (in-degree > 16)."""
    x, ei, ea = superpixel_like_graph(n=n, seed=seed)
    rng = np.random.default_rng(seed + 1)
    have = set(map(tuple, ei.T.tolist()))
    extra = [j for j in rng.permutation(n) if j != 3 and (int(j), 3) not in have][:hub_deg]
    src = np.concatenate([ei[0], np.array(extra), np.full(len(extra), 3)])
    dst = np.concatenate([ei[1], np.full(len(extra), 3), np.array(extra)])
    ea2 = rng.random((2 * len(extra), 5)).astype(np.float32)
    return x, np.stack([src, dst]).astype(np.int64), np.concatenate([ea, ea2], 0)


@pytest.mark.parametrize("case", ["graph_over_tile", "mean_over_tile", "direct", "hub_rows", "edges_leave_graph"])
def test_forward_aggregation_routes(oracle, gpu_ctx, case):
    """The forward pass aggregates with the graph-resident LDS kernel; every route around it must give the same
    logits: a graph larger than the 619-row tile inside a batch of small ones (per-block fallback), a mean size
    just above the tile (the direct gather since the 16-float slices went), one far above it, rows with more than 16 neighbours, and a
    batch vector that cuts through edges (neighbours outside the block's graph)."""
    from gcn_grabcut.data import Batch
    model, sd = seeded_state_dict(128, 2, seed=3)
    model = model.to("cuda").eval()
    if case == "graph_over_tile":
        graphs = [superpixel_like_graph(n=n, seed=n) for n in (700, 420, 380)]
    elif case == "mean_over_tile":
        graphs = [superpixel_like_graph(n=n, seed=n) for n in (760, 700)]
    elif case == "direct":
        graphs = [superpixel_like_graph(n=1203, seed=5)]
    elif case == "hub_rows":
        graphs = [_hub_graph(500, 40, seed=9), superpixel_like_graph(n=300, seed=2)]
    else:
        graphs = [superpixel_like_graph(n=600, seed=21)]
    off = np.cumsum([0] + [g[0].shape[0] for g in graphs])
    x = np.concatenate([g[0] for g in graphs])
    ei = np.concatenate([g[1] + off[i] for i, g in enumerate(graphs)], 1)
    ea = np.concatenate([g[2] for g in graphs])
    batch = np.concatenate([np.full(g[0].shape[0], i) for i, g in enumerate(graphs)])
    if case == "edges_leave_graph":
        batch = (np.arange(600) >= 280).astype(np.int64)          # two "graphs" with edges across the cut
    want, _ = oracle.resgcn_forward(_np_state(sd), 128, 2, x, ei, ea, batch)
    d = _data(x, ei, ea, batch=torch.as_tensor(batch))
    d.num_graphs = int(batch.max()) + 1
    got = model(d).cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("d", [32, 64, 96, 128])
def test_aggregate_bit_exact_and_fused(oracle, gpu_ctx, d):
    """M3 alone: CSR build + gather.  Bit-identical to the oracle with and without the fused epilogue (same edge order,
    one rounding per op, the shared GELU sequence of include/ggc_fmath.h)."""
    from gcn_grabcut import _native
    n = 1203
    x, ei, _ = superpixel_like_graph(n=n, seed=11)
    rng = np.random.default_rng(d)
    xw = rng.standard_normal((n, d)).astype(np.float32)
    bias = rng.standard_normal(d).astype(np.float32)
    gate = rng.random((n, d)).astype(np.float32)
    h = rng.standard_normal((n, d)).astype(np.float32)
    e = ei.shape[1]
    dev = "cuda"
    t = lambda a, dt=None: torch.as_tensor(a, dtype=dt).to(dev).contiguous()
    src, dst = t(ei[0], torch.int32), t(ei[1], torch.int32)
    row_ptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    col = torch.empty(e, dtype=torch.int32, device=dev)
    dis = torch.empty(n, dtype=torch.float32, device=dev)
    st = _native.current_stream(0)
    gpu_ctx.call("ggc_build_csr", st, n, e, src.data_ptr(), dst.data_ptr(), row_ptr.data_ptr(), col.data_ptr(),
                 dis.data_ptr())
    rp = row_ptr.cpu().numpy()
    assert rp[0] == 0 and rp[-1] == e
    assert (np.diff(rp) == np.bincount(ei[1], minlength=n)).all()
    # stable: inside every row, sources appear in edge order
    order = np.argsort(ei[1], kind="stable")
    assert (col.cpu().numpy() == ei[0][order]).all()
    xw_d, out = t(xw), torch.empty(n, d, device=dev)
    bias_d, gate_d, h_d = t(bias), t(gate), t(h)      # keep alive across the async launches
    gpu_ctx.call("ggc_gcn_aggregate", st, n, d, xw_d.data_ptr(), row_ptr.data_ptr(), col.data_ptr(), dis.data_ptr(),
                 bias_d.data_ptr(), None, None, out.data_ptr())
    want = oracle.gcn_aggregate(xw, ei, bias)
    assert np.array_equal(out.cpu().numpy(), want)
    gpu_ctx.call("ggc_gcn_aggregate", st, n, d, xw_d.data_ptr(), row_ptr.data_ptr(), col.data_ptr(), dis.data_ptr(),
                 bias_d.data_ptr(), gate_d.data_ptr(), h_d.data_ptr(), out.data_ptr())
    want = oracle.gcn_aggregate(xw, ei, bias, gate, h)
    assert np.array_equal(out.cpu().numpy(), want)


def test_errors_are_loud(gpu_ctx):
    from gcn_grabcut import _native
    from gcn_grabcut.model import ResGCNNet, build_model
    with pytest.raises(ValueError):
        build_model("nope")
    m = ResGCNNet(hidden_channels=32, n_layers=2).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(_data(*chain_graph(8)).cpu())
    with pytest.raises(_native.GGCError, match="GGC_E_UNSUPPORTED"):
        gpu_ctx.call("ggc_gcn_aggregate", 0, 4, 48, 1, 1, 1, 1, None, None, None, 1)
