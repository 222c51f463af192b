"""Does a GrabCut lane slow down because of OTHER streams' kernel boundaries (tiny kernels, no work) or because of their work?
One lane of 64 images alone, then with N side streams spamming (a) empty-ish kernels, (b) a bandwidth kernel."""
import os, sys, time, threading
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "src"))
import torch
from gcn_grabcut import GCNGrabCutPipeline, SuperpixelGraphConfig, ResGCNNet
from gcn_grabcut.synthetic import synthetic_batch
torch.manual_seed(0)
pipe = GCNGrabCutPipeline(ResGCNNet().eval(), sp_config=SuperpixelGraphConfig(n_segments=600), grabcut_lanes=1)
bgr = torch.from_numpy(synthetic_batch(64, 300, 400, 3)).cuda()
out = pipe.segment_batch_device(bgr)
eng, trimap = pipe._eng, out["trimap"]
def lane():
    ts = []
    for r in range(4):
        mask = trimap.clone()
        torch.cuda.synchronize(); t = time.perf_counter()
        eng.grabcut(bgr, mask, 5, 0, None, 0)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    return min(ts)
print(f"alone: {lane():.2f} ms", flush=True)
def make_graph():
    st = torch.cuda.Stream()
    small = torch.zeros(64, device="cuda")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        small.add_(1.0)
    st.synchronize()
    with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
        for _ in range(500): small.add_(1.0)       # the same tiny kernels, 500 per graph launch: same boundaries on the GPU, almost nothing on the host
    return st, g, small
graphs = [make_graph() for _ in range(3)]
torch.cuda.synchronize()
for kind, n_side in (("tiny", 3), ("graph", 1), ("graph", 3)):
    stop = False
    counts = [0] * n_side
    def spam(i):
        torch.cuda.set_device(0)
        if kind == "graph":
            st, g, _ = graphs[i]
            with torch.cuda.stream(st):
                while not stop:
                    g.replay(); counts[i] += 500; st.synchronize()
            return
        st = torch.cuda.Stream()
        small = torch.zeros(64, device="cuda")
        with torch.cuda.stream(st):
            while not stop:
                small.add_(1.0)
                counts[i] += 1
                if counts[i] % 64 == 0: st.synchronize()
        st.synchronize()
    th = [threading.Thread(target=spam, args=(i,)) for i in range(n_side)]
    for t_ in th: t_.start()
    time.sleep(0.2)
    c0 = sum(counts); t0 = time.perf_counter()
    ms = lane()
    rate = (sum(counts) - c0) / (time.perf_counter() - t0)
    stop = True
    for t_ in th: t_.join()
    print(f"{n_side} side stream(s) of {kind} kernels ({rate / 1e3:.1f} k kernels/s): lane {ms:.2f} ms", flush=True)
