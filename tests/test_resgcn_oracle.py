"""CPU tests: the C oracle's ResGCNNet against the PyG-free torch restatement,
the reference's structural known-answers, and its batched==single property."""
import numpy as np
import pytest
import torch

import torch_ref
from helpers import chain_graph, superpixel_like_graph, seeded_state_dict


def _np_state(sd):
    return {k: v.numpy() for k, v in sd.items() if v.dtype.is_floating_point}


@pytest.mark.parametrize("hidden,layers,expected", [(128, 6, 187826), (96, 6, 107090)])
def test_parameter_counts_match_reference_readme(hidden, layers, expected):
    # reference README.md:564-566,579
    from gcn_grabcut.model import ResGCNNet
    m = ResGCNNet(hidden_channels=hidden, n_layers=layers)
    assert sum(p.numel() for p in m.parameters()) == expected


def test_state_dict_keys_match_reference_contract():
    # SURVEY section 8 row M0 (reference model.py:449-499)
    from gcn_grabcut.model import ResGCNNet
    sd = ResGCNNet(hidden_channels=32, n_layers=2).state_dict()
    for k, shape in {
        "in_norm.norm.running_var": (19,), "in_norm.norm.num_batches_tracked": (),
        "input_proj.0.weight": (32, 19), "input_proj.1.bias": (32,),
        "prior_booster.0.weight": (8, 3), "prior_booster.2.weight": (32, 8),
        "edge_ctx.encode.0.weight": (16, 5), "edge_ctx.encode.2.weight": (16, 16),
        "edge_ctx.to_gate.0.weight": (16,), "edge_ctx.to_gate.1.weight": (32, 16),
        "gcn_layers.1.bias": (32,), "gcn_layers.1.lin.weight": (32, 32), "norms.1.weight": (32,),
        "sage.lin_l.weight": (32, 32), "sage.lin_l.bias": (32,), "sage.lin_r.weight": (32, 32),
        "sage_norm.bias": (32,), "jk_logits": (4,), "ctx.attn.weight": (1, 32), "ctx.attn.bias": (1,),
        "ctx.compress.weight": (16, 32), "ctx.expand.weight": (32, 16),
        "fuse.0.weight": (32,), "fuse.1.weight": (32, 32), "head.weight": (3, 32), "head.bias": (3,),
    }.items():
        assert tuple(sd[k].shape) == shape, k
    assert "sage.lin_r.bias" not in sd and "gcn_layers.0.lin.bias" not in sd


# (widths that are not a multiple of 32 or of 4: the oracle splits its sums where the zero-padded kernels do, the values are
# those of the true width)
@pytest.mark.parametrize("hidden,layers", [(32, 2), (96, 3), (128, 6), (9, 1), (16, 2), (33, 2), (35, 2), (100, 3), (127, 2)])
def test_oracle_matches_torch_restatement(oracle, hidden, layers):
    _, sd = seeded_state_dict(hidden, layers, seed=3)
    x, ei, ea = superpixel_like_graph(n=300, seed=5)
    ref = torch_ref.resgcn_forward(sd, layers, torch.from_numpy(x), torch.from_numpy(ei), torch.from_numpy(ea))
    logits, probs = oracle.resgcn_forward(_np_state(sd), hidden, layers, x, ei, ea)
    assert np.abs(logits - ref.numpy()).max() < 1e-4
    assert np.abs(probs - torch.softmax(ref, -1).numpy()).max() < 1e-5


def test_oracle_gcn_conv_matches_torch(oracle):
    x, ei, _ = superpixel_like_graph(n=200, seed=1)
    rng = np.random.default_rng(0)
    h = rng.standard_normal((200, 64)).astype(np.float32)
    w = (rng.standard_normal((64, 64)) * 0.1).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    ref = torch_ref.gcn_conv(torch.from_numpy(h), torch.from_numpy(ei), torch.from_numpy(w), torch.from_numpy(b))
    got = oracle.gcn_conv(h, ei, w, b)
    assert np.abs(got - ref.numpy()).max() < 2e-5


def test_oracle_batched_equals_single(oracle):
    # reference tests/test.py:294-306, atol 1e-4
    _, sd = seeded_state_dict(32, 2, seed=0)
    st = _np_state(sd)
    graphs = [chain_graph(40, seed=s) for s in (1, 2, 3)]
    one = np.concatenate([oracle.resgcn_forward(st, 32, 2, x.numpy(), ei.numpy(), ea.numpy())[0]
                          for x, ei, ea in graphs])
    off = np.cumsum([0] + [g[0].size(0) for g in graphs])
    x = np.concatenate([g[0].numpy() for g in graphs])
    ei = np.concatenate([g[1].numpy() + off[i] for i, g in enumerate(graphs)], 1)
    ea = np.concatenate([g[2].numpy() for g in graphs])
    batch = np.concatenate([np.full(g[0].size(0), i) for i, g in enumerate(graphs)])
    both, _ = oracle.resgcn_forward(st, 32, 2, x, ei, ea, batch)
    assert np.abs(one - both).max() < 1e-4


def test_isolated_node_and_empty_incoming(oracle):
    # a node with no incoming edge: scatter_mean clamps its count to 1 (model.py:73)
    _, sd = seeded_state_dict(32, 2, seed=1)
    x, ei, ea = chain_graph(10, seed=4)
    keep = ei[1] != 9                       # node 9 receives nothing
    ei, ea = ei[:, keep], ea[keep]
    ref = torch_ref.resgcn_forward(sd, 2, x, ei, ea)
    got, _ = oracle.resgcn_forward(_np_state(sd), 32, 2, x.numpy(), ei.numpy(), ea.numpy())
    assert np.isfinite(got).all()
    assert np.abs(got - ref.numpy()).max() < 1e-4
