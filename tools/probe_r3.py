"""round-3 probe: (1) exact early-exit potential (mask AND GMMs a fixed point before iteration 5?), (2) GrabCut stage time vs lanes / images per lane"""
import os, sys, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "src"))
import torch
from gcn_grabcut import GCNGrabCutPipeline, SuperpixelGraphConfig, ResGCNNet
from gcn_grabcut.synthetic import synthetic_batch
torch.manual_seed(0)
B = 256
pipe = GCNGrabCutPipeline(ResGCNNet().eval(), sp_config=SuperpixelGraphConfig(n_segments=600), grabcut_lanes=4)
bgr = torch.from_numpy(synthetic_batch(B, 300, 400, 3)).cuda()
out = pipe.segment_batch_device(bgr)
eng, trimap = pipe._eng, out["trimap"]
if os.environ.get("PROBE_CONVERGE", "1") == "1":
    masks, models = [], []
    for k in range(1, 6):
        m = trimap.clone()
        binary, m, bgd, fgd = eng.grabcut_lanes(bgr, m, k, 0, pipe.gc_config.seed, 4)
        masks.append(m.clone()); models.append(torch.cat([bgd, fgd], 1).clone())
    for k in range(1, 5):
        same_m = (masks[k] == masks[k - 1]).flatten(1).all(1)
        same_g = (models[k] == models[k - 1]).all(1)
        print(f"iteration {k + 1} vs {k}: mask unchanged {int(same_m.sum())}, GMMs unchanged {int(same_g.sum())}, both {int((same_m & same_g).sum())} of {B}", flush=True)
for lanes, nb in ((1, 8), (1, 16), (1, 32), (1, 64), (1, 128), (2, 128), (4, 64), (4, 128), (4, 256), (6, 256), (8, 256), (3, 256)):
    ts = []
    for r in range(3):
        mask = trimap[:nb].clone()
        torch.cuda.synchronize(); t = time.perf_counter()
        eng.grabcut_lanes(bgr[:nb], mask, 5, 0, pipe.gc_config.seed, lanes)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print(f"grabcut stage: {lanes} lanes x {nb // lanes} images (batch {nb}): {min(ts):.2f} ms  ({', '.join(f'{t:.1f}' for t in ts)})  HWQ={os.environ.get('GPU_MAX_HW_QUEUES')}", flush=True)
