"""How many images of the bench batch stop changing before the 5th GrabCut iteration?  (masks after k = 1..5 iterations)"""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "src"))
import torch
from gcn_grabcut import GCNGrabCutPipeline, SuperpixelGraphConfig, ResGCNNet
from gcn_grabcut.synthetic import synthetic_batch
torch.manual_seed(0)
B = int(os.environ.get("MF_BATCH", "256"))
pipe = GCNGrabCutPipeline(ResGCNNet().eval(), sp_config=SuperpixelGraphConfig(n_segments=600), grabcut_lanes=4)
bgr = torch.from_numpy(synthetic_batch(B, 300, 400, 3)).cuda()
out = pipe.segment_batch_device(bgr)
eng, trimap = pipe._eng, out["trimap"]
masks = []
for k in range(1, 6):
    m = trimap.clone()
    binary, m, bgd, fgd = eng.grabcut_lanes(bgr, m, k, 0, pipe.gc_config.seed, 4)
    masks.append(m.clone())
for k in range(1, 5):
    same = (masks[k] == masks[k - 1]).flatten(1).all(1)
    diff = (masks[k] != masks[k - 1]).flatten(1).sum(1).float()
    print(f"iteration {k + 1} vs {k}: {int(same.sum())} of {B} images unchanged; changed pixels per image: mean {diff.mean():.1f}, median {diff.median():.0f}, max {int(diff.max())}")
