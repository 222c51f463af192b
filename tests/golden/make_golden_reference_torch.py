"""
Generates tests/golden/reference_modules.npz by running the REFERENCE's own torch-only network blocks on fixed inputs.

Run in the build container only (the reference never travels), with the main interpreter (torch-CPU):
    python3 tests/golden/make_golden_reference_torch.py
`cv2` and the scikit-image names the reference imports at module level are registered as placeholders whose names are
None — the import lines succeed, nothing in them can be called — and /root/reference/src/gcn_grabcut/{graph_builder,
model}.py are loaded by path.  torch_geometric is absent: model.py guards that import (model.py:47-52), so the classes
that need it (GCNConv / SAGEConv / GATv2Conv users) cannot be built and are NOT recorded — their parity stays unpinned
(DESIGN.md section 2).  The blocks below touch torch only.

Recorded (reference file:line), for D in {32, 128}, one graph and a batch of three graphs (one node without incoming
edges in each case, one single-node graph in the batch), seeded weights:
  model.py:69-74     _scatter_mean                      (through the blocks)
  model.py:90-108    _graph_softmax                      (through GlobalContextModule, batch None and batched)
  model.py:111-139   EdgeContext(5, D)                   = M2 of the hot path
  model.py:142-162   EdgeInjectionLayer(5, D)            (GCNTrimapNet / GATTrimapNet)
  model.py:165-188   GlobalContextModule(D)              = M6
  model.py:191-213   InputNorm(19), eval mode, non-trivial running statistics   = M1
"""
import importlib.util
import pathlib
import sys
import types

import numpy as np
import torch

HERE = pathlib.Path(__file__).resolve().parent
REF = pathlib.Path("/root/reference/src/gcn_grabcut")


def placeholder(name, *attrs):
    m = types.ModuleType(name)
    for a in attrs:
        setattr(m, a, None)                     # importable names, nothing callable
    sys.modules[name] = m
    return m


placeholder("cv2")
sk = placeholder("skimage")
sk.segmentation = placeholder("skimage.segmentation", "slic", "find_boundaries", "mark_boundaries")
sk.color = placeholder("skimage.color", "rgb2lab", "rgb2hsv")
sk.measure = placeholder("skimage.measure", "regionprops")
pkg = types.ModuleType("refpkg")
pkg.__path__ = [str(REF)]
sys.modules["refpkg"] = pkg


def load(name):
    spec = importlib.util.spec_from_file_location(f"refpkg.{name}", REF / f"{name}.py")
    m = importlib.util.module_from_spec(spec)
    sys.modules[f"refpkg.{name}"] = m
    spec.loader.exec_module(m)
    return m


load("graph_builder")
model = load("model")
assert not model._TORCH_GEOMETRIC, "this script records the PyG-free blocks only"

out = {"torch_version": np.array(torch.__version__)}


def graph(rng, n, e):
    """random directed edges on n nodes; node n-1 has no incoming edge"""
    src = rng.integers(0, n, e)
    dst = rng.integers(0, max(n - 1, 1), e)
    return np.stack([src, dst]).astype(np.int64), rng.random((e, 5)).astype(np.float32)


def seeded(module, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            p.copy_((torch.rand(p.shape, generator=g) - 0.5) * (2.0 if p.dim() > 1 else 1.0) / max(1.0, float(p.shape[-1]) ** 0.5) * 2.0
                    + (1.0 if p.dim() == 1 and p.shape[0] > 1 and "norm" in type(module).__name__.lower() else 0.0))
    return module.eval()


rng = np.random.default_rng(20261005)
cases = {}
# one graph of 50 nodes; a batch of graphs with 37 + 1 + 22 nodes (PyG Batch layout: contiguous node ranges)
ei1, ea1 = graph(rng, 50, 400)
parts, off = [], 0
for n, e in ((37, 260), (1, 0), (22, 150)):
    ei, ea = graph(rng, n, e) if e else (np.zeros((2, 0), np.int64), np.zeros((0, 5), np.float32))
    parts.append((ei + off, ea, np.full(n, len(parts), np.int64)))
    off += n
ei3 = np.concatenate([p[0] for p in parts], 1)
ea3 = np.concatenate([p[1] for p in parts])
batch3 = np.concatenate([p[2] for p in parts])
cases["g1"] = (50, ei1, ea1, None)
cases["g3"] = (off, ei3, ea3, batch3)

with torch.no_grad():
    for D in (32, 128):
        ec = seeded(model.EdgeContext(5, D), 100 + D)
        inj = seeded(model.EdgeInjectionLayer(5, D), 200 + D)
        gcm = seeded(model.GlobalContextModule(D), 300 + D)
        for name, mod in (("edge_ctx", ec), ("edge_inject", inj), ("ctx", gcm)):
            for k, v in mod.state_dict().items():
                out[f"d{D}_{name}.{k}"] = v.numpy().copy()
        for cname, (n, ei, ea, batch) in cases.items():
            tei, tea = torch.from_numpy(ei), torch.from_numpy(ea)
            h = torch.from_numpy(rng.standard_normal((n, D)).astype(np.float32))
            out[f"d{D}_{cname}_h"] = h.numpy().copy()
            out[f"d{D}_{cname}_edge_ctx_gate"] = ec(tea, tei, n).numpy().copy()
            out[f"d{D}_{cname}_edge_inject_out"] = inj(tea, tei, n, h).numpy().copy()
            tb = None if batch is None else torch.from_numpy(batch)
            out[f"d{D}_{cname}_ctx_out"] = gcm(h, tb).numpy().copy()
            scores = gcm.attn(h)
            out[f"d{D}_{cname}_graph_softmax"] = model._graph_softmax(scores, tb).numpy().copy()
    for cname, (n, ei, ea, batch) in cases.items():
        out[f"{cname}_edge_index"], out[f"{cname}_edge_attr"] = ei, ea
        if batch is not None:
            out[f"{cname}_batch"] = batch
    norm = model.InputNorm(19)
    g = torch.Generator().manual_seed(7)
    norm.norm.weight.copy_(torch.rand(19, generator=g) + 0.5)
    norm.norm.bias.copy_(torch.rand(19, generator=g) - 0.5)
    norm.norm.running_mean.copy_(torch.rand(19, generator=g) - 0.3)
    norm.norm.running_var.copy_(torch.rand(19, generator=g) * 2.0 + 0.05)
    norm.eval()
    x = torch.from_numpy(rng.random((50, 19)).astype(np.float32) * 3.0 - 1.0)
    for k, v in norm.state_dict().items():
        if v.dtype.is_floating_point:
            out[f"in_norm.{k}"] = v.numpy().copy()
    out["in_norm_x"] = x.numpy().copy()
    out["in_norm_out"] = norm(x).numpy().copy()
    out["in_norm_out_single_node"] = norm(x[:1]).numpy().copy()      # eval mode: the stored statistics, like model.py:205-211
    # _scatter_mean on its own (model.py:69-74)
    src = torch.from_numpy(rng.standard_normal((400, 16)).astype(np.float32))
    out["scatter_src"] = src.numpy().copy()
    out["scatter_mean_out"] = model._scatter_mean(src, torch.from_numpy(ei1[1]), 50).numpy().copy()

np.savez_compressed(HERE / "reference_modules.npz", **out)
print(f"wrote {HERE / 'reference_modules.npz'}: {len(out)} arrays")
