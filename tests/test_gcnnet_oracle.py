"""GCNTrimapNet: the C oracle against a PyG-free torch restatement of reference model.py:239-316,
the host module's state_dict layout and parameter count."""
import numpy as np
import pytest
import torch

import torch_ref
from helpers import superpixel_like_graph


def seeded_gcnnet(hidden=64, n_layers=3, seed=0):
    """Reference-style init plus perturbed BatchNorm statistics / biases so that every term is exercised."""
    from gcn_grabcut.model import GCNTrimapNet
    torch.manual_seed(seed)
    m = GCNTrimapNet(hidden_channels=hidden, n_layers=n_layers).eval()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for k, v in m.state_dict().items():
            if not v.dtype.is_floating_point:
                continue
            if k.endswith("running_var"):
                v.copy_(0.5 + torch.rand(v.shape, generator=g))
            elif k.endswith("running_mean") or k.endswith("bias"):
                v.copy_(0.2 * torch.randn(v.shape, generator=g))
            elif v.dim() == 1:                      # BatchNorm weights
                v.copy_(1.0 + 0.2 * torch.randn(v.shape, generator=g))
    return m, {k: v.clone() for k, v in m.state_dict().items()}


def test_state_dict_layout_and_parameter_count():
    from gcn_grabcut.model import GCNTrimapNet, build_model
    m = GCNTrimapNet(hidden_channels=128, n_layers=6)
    keys = [k for k, v in m.state_dict().items() if v.dtype.is_floating_point]
    from oracle import oracle as orc
    assert sorted(keys) == sorted(orc.gcnnet_param_order(6))
    sd = m.state_dict()
    assert sd["blocks.0.conv.lin.weight"].shape == (128, 128) and sd["blocks.5.edge_inject.proj.0.weight"].shape == (128, 5)
    assert sd["head.0.weight"].shape == (128, 128 * 7) and sd["head.6.weight"].shape == (3, 64)
    assert isinstance(build_model("gcn", hidden_channels=32, n_layers=2), GCNTrimapNet)
    with pytest.raises(ValueError):
        build_model("transformer")
    # learnable parameters: in_norm 38 + input 19D+D+2D + blocks n(D^2 + D + 2D + 5D + D + D^2 + D) + head
    d, n = 128, 6
    want = 38 + (19 * d + d + 2 * d) + n * (2 * d * d + 10 * d) + (d * d * (n + 1) + d + 2 * d) + (d * d // 2 + d // 2) + (3 * d // 2 + 3)
    assert sum(p.numel() for p in m.parameters()) == want


@pytest.mark.parametrize("hidden,layers,n", [(32, 2, 70), (64, 3, 200), (128, 6, 300)])
def test_oracle_matches_torch_restatement(oracle, hidden, layers, n):
    m, sd = seeded_gcnnet(hidden, layers, seed=hidden)
    x, ei, ea = superpixel_like_graph(n=n, seed=n)
    want_l, want_p = torch_ref.gcnnet_forward(sd, layers, torch.as_tensor(x), torch.as_tensor(ei), torch.as_tensor(ea))
    st = {k: v.numpy() for k, v in sd.items() if v.dtype.is_floating_point}
    got_l, got_p = oracle.gcnnet_forward(st, hidden, layers, x, ei, ea)
    assert np.abs(got_l - want_l.numpy()).max() <= 1e-4
    assert np.abs(got_p - want_p.numpy()).max() <= 1e-5
    assert np.allclose(got_p.sum(1), 1.0, atol=1e-6)
