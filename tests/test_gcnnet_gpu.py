"""GPU parity of GCNTrimapNet through the C ABI against the CPU oracle: 1e-4 on logits."""
import numpy as np
import pytest
import torch

from helpers import chain_graph, superpixel_like_graph
from test_gcnnet_oracle import seeded_gcnnet

pytestmark = pytest.mark.gpu
TOL_LOGITS = 1e-4


def _data(x, ei, ea):
    from gcn_grabcut.data import Data
    return Data(x=torch.as_tensor(x), edge_index=torch.as_tensor(ei), edge_attr=torch.as_tensor(ea)).to("cuda")


@pytest.mark.parametrize("hidden,layers,n", [(32, 2, 70), (64, 3, 257), (96, 2, 300), (128, 6, 601), (16, 2, 90), (48, 3, 150)])
def test_forward_matches_oracle(oracle, gpu_ctx, hidden, layers, n):
    # 16 and 48 are not multiples of the MFMA tile: they run zero-padded to 32 / 64 and must still equal the oracle at width 16 / 48
    m, sd = seeded_gcnnet(hidden, layers, seed=hidden + layers)
    m = m.to("cuda").eval()
    x, ei, ea = superpixel_like_graph(n=n, seed=n)
    st = {k: v.numpy() for k, v in sd.items() if v.dtype.is_floating_point}
    want, want_p = oracle.gcnnet_forward(st, hidden, layers, x, ei, ea)
    d = _data(x, ei, ea)
    got = m(d).cpu().numpy()
    assert got.shape == (n, 3) and np.abs(got - want).max() <= TOL_LOGITS
    probs = m.predict_probs(d)
    assert np.abs(probs - want_p).max() <= 1e-5 and np.allclose(probs.sum(1), 1.0, atol=1e-6)


def test_batch_is_concatenation_isolated_node_and_errors(oracle, gpu_ctx):
    from gcn_grabcut import _native
    from gcn_grabcut.data import Batch
    from gcn_grabcut.model import GCNTrimapNet
    m, sd = seeded_gcnnet(64, 2, seed=4)
    m = m.to("cuda").eval()
    graphs = [superpixel_like_graph(n=n, seed=n) for n in (120, 37, 200)]
    datas = [_data(*g) for g in graphs]
    one = torch.cat([m(d) for d in datas]).cpu().numpy()
    both = m(Batch.from_data_list(datas)).cpu().numpy()          # no per-graph readout: batching = concatenation
    assert np.abs(one - both).max() <= TOL_LOGITS
    x, ei, ea = chain_graph(10, seed=4)
    keep = ei[1] != 9
    ei, ea = ei[:, keep], ea[keep]                                # node 9 receives nothing: gate 0, like scatter_mean
    st = {k: v.numpy() for k, v in sd.items() if v.dtype.is_floating_point}
    want, _ = oracle.gcnnet_forward(st, 64, 2, x.numpy(), ei.numpy(), ea.numpy())
    got = m(_data(x, ei, ea)).cpu().numpy()
    assert np.isfinite(got).all() and np.abs(got - want).max() <= TOL_LOGITS
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        GCNTrimapNet(hidden_channels=32, n_layers=2).eval()(_data(*chain_graph(8)).cpu())
    with pytest.raises(_native.GGCError):
        gpu_ctx.call("ggc_gcnnet_load_weight", b"not.a.key", None, 0)


def test_pipeline_runs_with_gcn_trimap_net(oracle, gpu_ctx):
    """`--model gcn` end to end: the pipeline's probabilities equal the oracle's GCNTrimapNet on the pipeline's own graph,
    and the rest of the chain (trimap, GrabCut, clean-up) produces a well-formed result."""
    from gcn_grabcut import GCNGrabCutPipeline, SuperpixelGraphConfig
    from gcn_grabcut.synthetic import synthetic_batch
    m, sd = seeded_gcnnet(64, 3, seed=9)
    pipe = GCNGrabCutPipeline(m.eval(), sp_config=SuperpixelGraphConfig(n_segments=150), device="cuda")
    imgs = synthetic_batch(2, 120, 160, config_id=5)
    out = pipe.segment_batch_device(pipe._eng.to_device(imgs))
    g = out["graphs"]
    st = {k: v.numpy() for k, v in sd.items() if v.dtype.is_floating_point}
    ei = torch.stack([g.edge_src, g.edge_dst]).cpu().numpy().astype(np.int64)
    _, want_p = oracle.gcnnet_forward(st, 64, 3, g.x.cpu().numpy(), ei, g.edge_attr.cpu().numpy())
    assert np.abs(out["probs"].cpu().numpy() - want_p).max() <= 1e-4
    r = pipe.segment(imgs[0])
    assert r.binary_mask.shape == (120, 160) and set(np.unique(r.trimap)) <= {0, 1, 2, 3} and set(np.unique(r.binary_mask)) <= {0, 1}
