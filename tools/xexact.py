import sys, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "src")); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import gpu_helpers as gh
from gcn_grabcut import _native
from gcn_grabcut.synthetic import synthetic_batch
from oracle import oracle
ctx = _native.get_context(0)
bgr = synthetic_batch(3, 300, 400, config_id=3)
_, lab, hsv, gray, grad = gh.preprocess(ctx, bgr)
seg, nn = gh.slic(ctx, lab, 600)
g = gh.graph(ctx, seg, nn, lab, hsv, grad, 4, 4)
seg_h, lab_h, hsv_h, grad_h = seg.cpu().numpy(), lab.cpu().numpy(), hsv.cpu().numpy(), grad.cpu().numpy()
for i in range(3):
    want = oracle.graph_build(seg_h[i], lab_h[i], hsv_h[i], grad_h[i], 4, 4)
    n0, n1 = g["node_ptr"][i], g["node_ptr"][i + 1]; e0, e1 = g["edge_ptr"][i], g["edge_ptr"][i + 1]
    x = g["x"][n0:n1].cpu().numpy(); ea = g["attr"][e0:e1].cpu().numpy()
    wx = np.concatenate([want["node_features"], want["prior"]], 1)
    print("image", i, "x exact per column:", np.round((x == wx).mean(0), 3), "max diff", np.abs(x - wx).max(0).max())
    print("   edge_attr exact per column:", np.round((ea == want["edge_attr"]).mean(0), 3), "max diff", np.abs(ea - want["edge_attr"]).max())
    # are lab/hsv/grad from the GPU preprocess equal to the oracle's?
    lo, ho, go_, gro = oracle.preprocess(bgr[i])
    print("   preprocess exact: lab", (lab_h[i] == lo).mean(), "hsv", (hsv_h[i] == ho).mean(), "grad", (grad_h[i] == gro).mean())
