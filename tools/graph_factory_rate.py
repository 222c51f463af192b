"""experiment helper: graphs/s of the batched cache writer (in-memory samples, no disk) at the bench's image shape"""
import os, sys, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "src"))
import numpy as np, torch
from gcn_grabcut import dataset as ds
from gcn_grabcut.graph_builder import SuperpixelGraphConfig
from gcn_grabcut.synthetic import synthetic_image
n = int(os.environ.get("N", "512"))
samples = []
for i in range(n):
    img, gt = synthetic_image(300, 400, 40_000 + i, return_mask=True)
    samples.append({"image": np.ascontiguousarray(img), "gt_mask": np.ascontiguousarray(gt.astype(np.uint8)), "name": str(i)})
cfg = SuperpixelGraphConfig(n_segments=600)
ds.prepare_dataset(samples[:64], cfg, batch_size=64)            # warm-up
for bs in (64, 256):
    torch.cuda.synchronize(); t = time.perf_counter()
    recs = ds.prepare_dataset(samples, cfg, batch_size=bs, keep_segments=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"batch {bs}: {len(recs)} graphs in {dt:.2f} s = {len(recs) / dt:.0f} graphs/s (device stages + host slicing into per-image Data)")
