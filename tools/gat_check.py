import sys, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "src")); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
from helpers import superpixel_like_graph
from test_gat_oracle import seeded_gat
from gcn_grabcut.data import Data
from gcn_grabcut import _native
from oracle import oracle
st = lambda sd: {k: v.numpy() for k, v in sd.items() if v.dtype.is_floating_point}
hidden, layers, n = 32, 1, 80
x, ei, ea = superpixel_like_graph(n=n, seed=n)
d = Data(x=torch.as_tensor(x), edge_index=torch.as_tensor(ei), edge_attr=torch.as_tensor(ea)).to("cuda")
m, sd = seeded_gat(hidden, layers, seed=5)
m = m.to("cuda").eval()
os.environ["GGO_GAT_DUMP"] = "/tmp/gat_dump.bin"
want, _ = oracle.gat_forward(st(sd), hidden, layers, x, ei, ea)
got = m(d).cpu().numpy()
ctx = _native.get_context(0)
def rd(name, count):
    a = np.empty(count, np.float32)
    ctx.call("ggc_debug_read_scratch", name.encode(), a.ctypes.data, a.nbytes)
    return a
ND = n * hidden
dump = np.fromfile("/tmp/gat_dump.bin", np.float32)
o = {}
off = 0
for k, c in (("skip", ND), ("xl", ND), ("xr", ND), ("act", ND), ("h", ND), ("hs", ND), ("score", n), ("gs", hidden)):
    o[k] = dump[off:off + c]; off += c
states = rd("net_states", 4 * ND)
g = {"h_after": states[ND:2 * ND], "h0": states[:ND], "skip": states[2 * ND:3 * ND], "act": states[3 * ND:4 * ND],
     "xl": rd("net_xw", ND), "xr": rd("net_agg", ND), "hs": rd("net_hjk", ND), "score": rd("net_score", n), "gs": rd("net_gvec", hidden)}
for k, (a, b) in {"skip": (g["skip"], o["skip"]), "xl": (g["xl"], o["xl"]), "xr": (g["xr"], o["xr"]), "act": (g["act"], o["act"]),
                  "h (after layer)": (g["h_after"], o["h"]), "hs": (g["hs"], o["hs"]), "score": (g["score"], o["score"]), "gvec": (g["gs"], o["gs"])}.items():
    print(f"{k}: exact {np.mean(a == b):.4f} max diff {np.abs(a - b).max():.3g}")
print("logits exact", np.mean(got == want))
