"""experiment helper: wall time per pipeline stage (with a device sync after each) on the bench workload"""
import os, sys, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "src"))
import torch
from gcn_grabcut import GCNGrabCutPipeline, SuperpixelGraphConfig, ResGCNNet
from gcn_grabcut.synthetic import synthetic_batch
torch.manual_seed(0)
B = int(os.environ.get("B", "256"))
H, W, NSEG = int(os.environ.get("H", "300")), int(os.environ.get("W", "400")), int(os.environ.get("NSEG", "600"))
model = ResGCNNet().eval()
pipe = GCNGrabCutPipeline(model, sp_config=SuperpixelGraphConfig(n_segments=NSEG), grabcut_lanes=int(os.environ.get("LANES", "4")))
bgr = torch.from_numpy(synthetic_batch(B, H, W, 3)).cuda()
eng, cfg = pipe._eng, pipe.sp_config
def T(f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, (time.perf_counter() - t) * 1e3
for rep in range(3):
    tt = {}
    (lab, hsv, gray, grad), tt["preprocess"] = T(lambda: eng.preprocess(bgr))
    (seg, n_nodes), tt["slic"] = T(lambda: eng.slic(lab, cfg.n_segments, cfg.compactness, cfg.sigma))
    graphs, tt["graph"] = T(lambda: eng.build_graphs(seg, n_nodes, lab, hsv, grad, cfg.connectivity, cfg.n_nonlocal))
    probs, tt["gcn"] = T(lambda: eng.predict_probs(model.cuda(), graphs))
    trimap, tt["trimap"] = T(lambda: eng.refine_trimap(probs, graphs.node_ptr, seg, bgr, 0.55, 0.55, 8, 1e-3, True))
    trimap, tt["seed_prior"] = T(lambda: eng.seed_from_prior(trimap, graphs.x[:, 16:19], graphs.node_ptr, seg, 0.1))
    mask = trimap.clone()
    (binary, mask, bgd, fgd), tt["grabcut"] = T(lambda: eng.grabcut_lanes(bgr, mask, 5, 0, 0, pipe.grabcut_lanes))
    cleaned, tt["clean"] = T(lambda: eng.clean_mask(binary, 0.002, False))
    _, tt["compose"] = T(lambda: eng.compose(bgr, cleaned))
    print({k: round(v, 2) for k, v in tt.items()}, "total", round(sum(tt.values()), 1))
