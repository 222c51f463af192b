"""software pipeline of segment_batch_device: step time per (chunks, ratio) plan, outputs compared with the one-chunk run"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("HWQ", "8"))
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "src"))
import torch
from gcn_grabcut import GCNGrabCutPipeline, SuperpixelGraphConfig, ResGCNNet
from gcn_grabcut.synthetic import synthetic_batch
torch.manual_seed(0)
B = int(os.environ.get("B", "256"))
pipe = GCNGrabCutPipeline(ResGCNNet().eval(), sp_config=SuperpixelGraphConfig(n_segments=600), grabcut_lanes=4)
bgr = torch.from_numpy(synthetic_batch(B, 300, 400, 3)).cuda()
plans = [tuple(float(v) for v in p.split(":")) for p in os.environ.get("PLANS", "4:1.0,4:0.85,4:0.7,5:0.8,6:0.8,8:0.85,3:0.7").split(",")]
def run(chunks, ratio, steps=int(os.environ.get('STEPS', '5'))):
    pipe.chunk_ratio = ratio
    out = pipe.segment_batch_device(bgr, chunks=chunks)      # warm (lanes, arenas)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps):
        out = pipe.segment_batch_device(bgr, chunks=chunks)
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t) / steps * 1e3
ref, ms = run(1, 1.0)
print(f"one chunk (4 GrabCut lanes): {ms:.2f} ms per step", flush=True)
for chunks, ratio in plans:
    out, ms = run(int(chunks), ratio)
    same = all(torch.equal(out[k], ref[k]) for k in ("binary_mask", "trimap", "segments", "gc_mask", "probs", "overlay", "rgba"))
    g, r = out["graphs"], ref["graphs"]
    same_g = torch.equal(g.x, r.x) and torch.equal(g.edge_src, r.edge_src) and torch.equal(g.edge_dst, r.edge_dst) and torch.equal(g.edge_attr, r.edge_attr) \
        and torch.equal(g.node_ptr, r.node_ptr) and (g.edge_ptr_host == r.edge_ptr_host).all()
    print(f"chunks {int(chunks)} ratio {ratio}: plan {[hi - lo for lo, hi in pipe.chunk_plan(B, int(chunks), ratio)]}: {ms:.2f} ms per step, outputs identical {same}, graphs identical {bool(same_g)}", flush=True)
if os.environ.get("STEPS"): sys.exit(0)
ref, ms = run(1, 1.0)
print(f"one chunk again: {ms:.2f} ms per step", flush=True)
