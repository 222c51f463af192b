"""GATTrimapNet (last rank): the C oracle against a PyG-free torch restatement of reference model.py:323-414
(GATv2Conv from PyG's documented semantics: parity with the library itself is unpinned, it is absent), the host module's
state_dict layout and parameter count."""
import numpy as np
import pytest
import torch

import torch_ref
from helpers import superpixel_like_graph


def seeded_gat(hidden=64, n_layers=3, seed=0, heads=8):
    """Reference-style init plus perturbed norm statistics / biases so that every term is exercised."""
    from gcn_grabcut.model import GATTrimapNet
    torch.manual_seed(seed)
    m = GATTrimapNet(hidden_channels=hidden, n_layers=n_layers, n_heads=heads).eval()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for k, v in m.state_dict().items():
            if not v.dtype.is_floating_point:
                continue
            if k.endswith("running_var"):
                v.copy_(0.5 + torch.rand(v.shape, generator=g))
            elif k.endswith("running_mean") or k.endswith("bias"):
                v.copy_(0.2 * torch.randn(v.shape, generator=g))
            elif v.dim() == 1:                      # norm weights
                v.copy_(1.0 + 0.2 * torch.randn(v.shape, generator=g))
    return m, {k: v.clone() for k, v in m.state_dict().items()}


def test_state_dict_layout_and_parameter_count():
    from gcn_grabcut.model import GATTrimapNet, build_model
    from oracle import oracle as orc
    m = GATTrimapNet(hidden_channels=128, n_layers=5)
    keys = [k for k, v in m.state_dict().items() if v.dtype.is_floating_point]
    assert sorted(keys) == sorted(orc.gat_param_order(5))
    sd = m.state_dict()
    assert sd["convs.0.att"].shape == (1, 8, 16) and sd["convs.4.lin_edge.weight"].shape == (128, 5)
    assert sd["convs.0.lin_l.weight"].shape == (128, 128) and "convs.0.lin_edge.bias" not in sd
    assert sd["skip_proj.weight"].shape == (128, 128) and "skip_proj.bias" not in sd and sd["head.3.weight"].shape == (3, 128)
    g = build_model("gat", hidden_channels=32, n_layers=2)          # reference model.py:615-616: 8 heads
    assert isinstance(g, GATTrimapNet) and g.n_heads == 8 and g.n_layers == 2
    d, n = 128, 5
    conv = d + 2 * (d * d + d) + 5 * d + d                            # att, lin_l, lin_r, lin_edge, bias
    want = 38 + (19 * d + d + 2 * d) + n * (conv + 2 * d + (5 * d + d + d * d + d)) + d * d + (d + 1 + d * d // 2 + d // 2 + d * d // 2 + d) \
        + (d * d + d + 3 * d + 3)
    assert sum(p.numel() for p in m.parameters()) == want
    assert GATTrimapNet(hidden_channels=64, n_heads=2, n_layers=1).state_dict()["convs.0.att"].shape == (1, 2, 32)
    for bad in (dict(hidden_channels=96), dict(n_heads=3), dict(n_heads=16), dict(hidden_channels=48)):
        with pytest.raises(ValueError):
            GATTrimapNet(**bad)


@pytest.mark.parametrize("hidden,layers,n", [(32, 2, 70), (64, 3, 200), (128, 5, 300)])
def test_oracle_matches_torch_restatement(oracle, hidden, layers, n):
    m, sd = seeded_gat(hidden, layers, seed=hidden)
    x, ei, ea = superpixel_like_graph(n=n, seed=n)
    want_l, want_p = torch_ref.gat_forward(sd, layers, torch.as_tensor(x), torch.as_tensor(ei), torch.as_tensor(ea))
    st = {k: v.numpy() for k, v in sd.items() if v.dtype.is_floating_point}
    got_l, got_p = oracle.gat_forward(st, hidden, layers, x, ei, ea)
    assert np.abs(got_l - want_l.numpy()).max() <= 1e-4
    assert np.abs(got_p - want_p.numpy()).max() <= 1e-5
    assert np.allclose(got_p.sum(1), 1.0, atol=1e-6)


@pytest.mark.parametrize("hidden,heads", [(32, 1), (32, 4), (64, 2), (64, 1), (128, 4), (128, 2), (128, 1)])
def test_other_head_counts_match_the_torch_restatement(oracle, hidden, heads):
    """reference model.py:323-414 takes any n_heads that divides hidden_channels; this build runs 1, 2, 4 and 8 heads
    (a head of C = hidden / heads channels sits on C consecutive lanes; one head at 128 spans both registers of a lane)"""
    m, sd = seeded_gat(hidden, 2, seed=hidden + heads, heads=heads)
    x, ei, ea = superpixel_like_graph(n=90, seed=heads)
    want_l, want_p = torch_ref.gat_forward(sd, 2, torch.as_tensor(x), torch.as_tensor(ei), torch.as_tensor(ea), heads=heads)
    st = {k: v.numpy() for k, v in sd.items() if v.dtype.is_floating_point}
    got_l, got_p = oracle.gat_forward(st, hidden, 2, x, ei, ea, heads=heads)
    assert np.abs(got_l - want_l.numpy()).max() <= 1e-4 and np.abs(got_p - want_p.numpy()).max() <= 1e-5


def test_oracle_batched_equals_single_and_isolated_nodes(oracle):
    """reference tests/test.py:294-306 for the attention variant; a node without incoming edges attends to itself only"""
    m, sd = seeded_gat(32, 2, seed=9)
    st = {k: v.numpy() for k, v in sd.items() if v.dtype.is_floating_point}
    graphs = [superpixel_like_graph(n=n, seed=n) for n in (40, 61)]
    singles = np.concatenate([oracle.gat_forward(st, 32, 2, *g)[0] for g in graphs])
    off = [0, 40]
    x = np.concatenate([g[0] for g in graphs]); ea = np.concatenate([g[2] for g in graphs])
    ei = np.concatenate([g[1] + off[i] for i, g in enumerate(graphs)], 1)
    batch = np.concatenate([np.full(g[0].shape[0], i) for i, g in enumerate(graphs)])
    both, _ = oracle.gat_forward(st, 32, 2, x, ei, ea, batch)
    assert np.abs(both - singles).max() <= 1e-4
    want, _ = torch_ref.gat_forward(sd, 2, torch.as_tensor(x), torch.as_tensor(ei), torch.as_tensor(ea), torch.as_tensor(batch))
    assert np.abs(both - want.numpy()).max() <= 1e-4
    x1, ei1, ea1 = graphs[0]
    keep = ei1[1] != 7                                      # node 7 loses its incoming edges
    got, _ = oracle.gat_forward(st, 32, 2, x1, ei1[:, keep], ea1[keep])
    want, _ = torch_ref.gat_forward(sd, 2, torch.as_tensor(x1), torch.as_tensor(ei1[:, keep]), torch.as_tensor(ea1[keep]))
    assert np.isfinite(got).all() and np.abs(got - want.numpy()).max() <= 1e-4
