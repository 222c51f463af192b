"""How far are the GPU's ResGCNNet outputs and the end-to-end trimaps from the oracle's, bit for bit?"""
import sys, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "src")); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
from helpers import superpixel_like_graph, seeded_state_dict
from gcn_grabcut.data import Data, Batch
from gcn_grabcut import GCNGrabCutPipeline, SuperpixelGraphConfig
from gcn_grabcut.synthetic import synthetic_batch
from oracle import oracle
st = lambda sd: {k: v.numpy() for k, v in sd.items() if v.dtype.is_floating_point}
for hidden, layers, n in ((32, 2, 80), (64, 2, 257), (96, 3, 300), (128, 6, 601)):
    model, sd = seeded_state_dict(hidden, layers, seed=hidden + layers)
    model = model.to("cuda").eval()
    x, ei, ea = superpixel_like_graph(n=n, seed=n)
    want, want_p = oracle.resgcn_forward(st(sd), hidden, layers, x, ei, ea)
    d = Data(x=torch.as_tensor(x), edge_index=torch.as_tensor(ei), edge_attr=torch.as_tensor(ea)).to("cuda")
    got = model(d).cpu().numpy(); probs = model.predict_probs(d)
    print(f"D={hidden} n={layers} N={n}: logits max diff {np.abs(got - want).max():.3g} exact {np.mean(got == want):.4f}; probs max diff {np.abs(probs - want_p).max():.3g} exact {np.mean(probs == want_p):.4f}", flush=True)
model, sd = seeded_state_dict(128, 6, seed=0)
pipe = GCNGrabCutPipeline(model.eval(), sp_config=SuperpixelGraphConfig(n_segments=600), device="cuda")
imgs = synthetic_batch(6, 300, 400, config_id=3)
out = pipe.segment_batch_device(pipe._eng.to_device(imgs))
g = out["graphs"]
for i in range(6):
    want = oracle.segment(imgs[i], st(sd), 128, 6, n_segments=600, seed=i)
    n0, n1 = g.node_ptr_host[i], g.node_ptr_host[i + 1]
    xg = g.x[n0:n1].cpu().numpy(); pg = out["probs"][n0:n1].cpu().numpy()
    print(f"image {i}: x exact {np.mean(xg == want['x']):.4f} (prior cols {np.mean(xg[:, 16:] == want['x'][:, 16:]):.4f}); probs exact {np.mean(pg == want['probs']):.4f} maxdiff {np.abs(pg - want['probs']).max():.3g}; "
          f"trimap equal {np.array_equal(out['trimap'][i].cpu().numpy(), want['trimap'])}; mask equal {np.array_equal(out['binary_mask'][i].cpu().numpy(), want['binary_mask'])}", flush=True)
