"""experiment helper: GCNTrimapNet forward time on a batch of DUTS-shape graphs (random weights)"""
import os, sys, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "src"))
import numpy as np, torch
sys.path.insert(0, root)
from bench import synthetic_region_graph
from gcn_grabcut.data import Batch, Data
from gcn_grabcut.model import GCNTrimapNet
B = int(os.environ.get("B", "256"))
rng = np.random.default_rng(1)
graphs = [synthetic_region_graph(int(rng.integers(585, 618)), rng) for _ in range(B)]
batch = Batch.from_data_list([Data(x=torch.from_numpy(x), edge_index=torch.from_numpy(ei), edge_attr=torch.from_numpy(ea))
                              for x, ei, ea in graphs]).to("cuda")
torch.manual_seed(0)
m = GCNTrimapNet(hidden_channels=128, n_layers=6).to("cuda").eval()
for _ in range(2): m.predict_probs_device(batch)
torch.cuda.synchronize(); t = time.perf_counter()
K = 10
for _ in range(K): m.predict_probs_device(batch)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / K
n, e = batch.x.size(0), batch.edge_index.size(1)
flop = 6 * (2 * n * 128 * 128 + 2 * e * 128 * 128) + 2 * n * 128 * 128 * 7
print(f"GCNTrimapNet(D=128, n=6) batch {B}: {n} nodes, {e} edges: {dt*1e3:.2f} ms per forward = {B/dt:.0f} graphs/s, "
      f"{flop/dt/1e12:.1f} TFLOP/s on the products (edge MLP dominates: {6*2*e*128*128/1e9:.0f} GFLOP)")
