"""Times the GrabCut stage alone (trimaps prepared once) — schedule sweeps of the max-flow drivers.
   env: MF_BATCH (256), LANES (1), REPS (3); every GGC_MF* variable is read by the library itself."""
import os, sys, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "src"))
import torch
from gcn_grabcut import GCNGrabCutPipeline, SuperpixelGraphConfig
from gcn_grabcut import ResGCNNet
from gcn_grabcut.synthetic import synthetic_batch

torch.manual_seed(0)
B, lanes, reps = int(os.environ.get("MF_BATCH", "256")), int(os.environ.get("LANES", "1")), int(os.environ.get("REPS", "3"))
H, W, nseg = int(os.environ.get("MF_H", "300")), int(os.environ.get("MF_W", "400")), int(os.environ.get("MF_NSEG", "600"))
pipe = GCNGrabCutPipeline(ResGCNNet().eval(), sp_config=SuperpixelGraphConfig(n_segments=nseg), grabcut_lanes=lanes)
bgr = torch.from_numpy(synthetic_batch(B, H, W, 3)).cuda()
out = pipe.segment_batch_device(bgr)
torch.cuda.synchronize()
eng, trimap = pipe._eng, out["trimap"]
ref = out["gc_mask"].clone()
ts = []
for r in range(reps):
    mask = trimap.clone()
    torch.cuda.synchronize(); t = time.perf_counter()
    binary, mask, bgd, fgd = eng.grabcut_lanes(bgr, mask, 5, 0, pipe.gc_config.seed, lanes)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    assert torch.equal(mask, ref)
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("GGC_MF"))
print(f"grabcut stage B={B} lanes={lanes}: {min(ts):.2f} ms (all: {', '.join(f'{t:.1f}' for t in ts)}) [{tag}]", flush=True)
