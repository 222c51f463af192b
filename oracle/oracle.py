"""
ctypes loader for libggc_oracle.so — the CPU restatement of the hot path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product package.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_lib = None

_vp, _i, _f, _d, _i64, _u64 = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_int64, C.c_uint64


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = _DIR / "libggc_oracle.so"
        if not path.exists():
            raise RuntimeError(f"{path} missing: run `make -C {_DIR}`")
        _lib = C.CDLL(str(path))
        for name, res, args in (("ggo_cbrt", _d, [_d]), ("ggo_pow24", _d, [_d]), ("ggo_exp", _d, [_d]), ("ggo_log", _d, [_d]), ("ggo_iou", _d, None),
                                ("ggo_grid_maxflow", _i64, None), ("ggo_graph_build", _vp, None)):
            fn = getattr(_lib, name, None)
            if fn is None:
                continue
            fn.restype = res
            if args is not None:
                fn.argtypes = args
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_vp)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def resgcn_param_order(n_layers: int) -> list[str]:
    """state_dict keys in the order oracle/resgcn.c expects."""
    keys = [
        "in_norm.norm.weight", "in_norm.norm.bias", "in_norm.norm.running_mean", "in_norm.norm.running_var",
        "input_proj.0.weight", "input_proj.0.bias", "input_proj.1.weight", "input_proj.1.bias",
        "prior_booster.0.weight", "prior_booster.0.bias", "prior_booster.2.weight", "prior_booster.2.bias",
        "edge_ctx.encode.0.weight", "edge_ctx.encode.0.bias", "edge_ctx.encode.2.weight", "edge_ctx.encode.2.bias",
        "edge_ctx.to_gate.0.weight", "edge_ctx.to_gate.0.bias", "edge_ctx.to_gate.1.weight", "edge_ctx.to_gate.1.bias",
    ]
    for i in range(n_layers):
        keys += [f"gcn_layers.{i}.bias", f"gcn_layers.{i}.lin.weight", f"norms.{i}.weight", f"norms.{i}.bias"]
    keys += [
        "sage.lin_l.weight", "sage.lin_l.bias", "sage.lin_r.weight", "sage_norm.weight", "sage_norm.bias",
        "jk_logits", "ctx.attn.weight", "ctx.attn.bias", "ctx.compress.weight", "ctx.compress.bias",
        "ctx.expand.weight", "ctx.expand.bias", "fuse.0.weight", "fuse.0.bias", "fuse.1.weight", "fuse.1.bias",
        "head.weight", "head.bias",
    ]
    return keys


def resgcn_forward(state: dict, hidden: int, n_layers: int, x, edge_index, edge_attr, batch=None):
    """state: {key: np.ndarray}. Returns (logits, probs) float32 (N,3)."""
    L = lib()
    keys = resgcn_param_order(n_layers)
    assert L.ggo_resgcn_n_params(n_layers) == len(keys)
    arrs = [f32(np.asarray(state[k])) for k in keys]
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    x = f32(x)
    ei = np.ascontiguousarray(edge_index, dtype=np.int64)
    ea = f32(edge_attr)
    n, e = x.shape[0], ei.shape[1]
    b = None if batch is None else np.ascontiguousarray(batch, dtype=np.int64)
    ng = 1 if b is None else int(b.max()) + 1
    logits = np.empty((n, 3), np.float32)
    probs = np.empty((n, 3), np.float32)
    rc = L.ggo_resgcn_forward(ptrs, _i(hidden), _i(n_layers), _i(n), _i(e), _p(x), _p(ei), _p(ea), _p(b), _i(ng),
                              _p(logits), _p(probs))
    assert rc == 0
    return logits, probs


def _ptr_array(arrs):
    return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def input_norm(x, weight, bias, running_mean, running_var):
    """InputNorm in eval mode (reference model.py:191-213) -> (N,19) float32."""
    x = f32(x)
    out = np.empty_like(x)
    w, b, m, v = f32(weight), f32(bias), f32(running_mean), f32(running_var)
    lib().ggo_input_norm(_i(x.shape[0]), _p(x), _p(w), _p(b), _p(m), _p(v), _p(out))
    return out


def edge_context(state: dict, hidden: int, edge_attr, edge_index, n_nodes: int, prefix: str = ""):
    """EdgeContext.forward (reference model.py:111-139) -> gate (N,D).  state: the module's state_dict as arrays."""
    keys = ["encode.0.weight", "encode.0.bias", "encode.2.weight", "encode.2.bias",
            "to_gate.0.weight", "to_gate.0.bias", "to_gate.1.weight", "to_gate.1.bias"]
    arrs = [f32(np.asarray(state[prefix + k])) for k in keys]
    ei = np.ascontiguousarray(edge_index, dtype=np.int64)
    ea = f32(edge_attr)
    gate = np.empty((n_nodes, hidden), np.float32)
    assert lib().ggo_edge_context(_ptr_array(arrs), _i(hidden), _i(n_nodes), _i(ei.shape[1]), _p(ea), _p(ei), _p(gate)) == 0
    return gate


def global_context(state: dict, hidden: int, h, batch=None, prefix: str = ""):
    """GlobalContextModule.forward (reference model.py:165-188) -> (h * g (N,D), per-graph softmax weights (N,))."""
    keys = ["attn.weight", "attn.bias", "compress.weight", "compress.bias", "expand.weight", "expand.bias"]
    arrs = [f32(np.asarray(state[prefix + k])) for k in keys]
    h = f32(h)
    b = None if batch is None else np.ascontiguousarray(batch, dtype=np.int64)
    ng = 1 if b is None else int(b.max()) + 1
    out, w = np.empty_like(h), np.empty(h.shape[0], np.float32)
    assert lib().ggo_global_context(_ptr_array(arrs), _i(hidden), _i(h.shape[0]), _p(h), _p(b), _i(ng), _p(out), _p(w)) == 0
    return out, w


def edge_injection(state: dict, hidden: int, edge_attr, edge_index, node_updates, prefix: str = ""):
    """EdgeInjectionLayer.forward (reference model.py:142-162) -> node_updates * scatter_mean(gates)."""
    keys = ["proj.0.weight", "proj.0.bias", "proj.2.weight", "proj.2.bias"]
    arrs = [f32(np.asarray(state[prefix + k])) for k in keys]
    ei = np.ascontiguousarray(edge_index, dtype=np.int64)
    ea, nu = f32(edge_attr), f32(node_updates)
    out = np.empty_like(nu)
    assert lib().ggo_edge_injection(_ptr_array(arrs), _i(hidden), _i(nu.shape[0]), _i(ei.shape[1]), _p(ea), _p(ei), _p(nu), _p(out)) == 0
    return out


def gcnnet_param_order(n_layers: int) -> list[str]:
    """GCNTrimapNet state_dict keys in the order oracle/gcnnet.c expects."""
    bn = ("weight", "bias", "running_mean", "running_var")
    keys = [f"in_norm.norm.{k}" for k in bn] + ["input_proj.0.weight", "input_proj.0.bias"] + [f"input_proj.1.{k}" for k in bn]
    for i in range(n_layers):
        keys += [f"blocks.{i}.conv.bias", f"blocks.{i}.conv.lin.weight"] + [f"blocks.{i}.bn.{k}" for k in bn]
        keys += [f"blocks.{i}.edge_inject.proj.0.weight", f"blocks.{i}.edge_inject.proj.0.bias",
                 f"blocks.{i}.edge_inject.proj.2.weight", f"blocks.{i}.edge_inject.proj.2.bias"]
    keys += ["head.0.weight", "head.0.bias"] + [f"head.1.{k}" for k in bn]
    keys += ["head.4.weight", "head.4.bias", "head.6.weight", "head.6.bias"]
    return keys


def gcnnet_forward(state: dict, hidden: int, n_layers: int, x, edge_index, edge_attr):
    """GCNTrimapNet (eval). state: {key: np.ndarray}. Returns (logits, probs) float32 (N,3)."""
    L = lib()
    keys = gcnnet_param_order(n_layers)
    assert L.ggo_gcnnet_n_params(n_layers) == len(keys)
    arrs = [f32(np.asarray(state[k])) for k in keys]
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    x = f32(x)
    ei = np.ascontiguousarray(edge_index, dtype=np.int64)
    ea = f32(edge_attr)
    n, e = x.shape[0], ei.shape[1]
    logits = np.empty((n, 3), np.float32)
    probs = np.empty((n, 3), np.float32)
    rc = L.ggo_gcnnet_forward(ptrs, _i(hidden), _i(n_layers), _i(n), _i(e), _p(x), _p(ei), _p(ea), _p(logits), _p(probs))
    assert rc == 0
    return logits, probs


def gat_param_order(n_layers: int) -> list[str]:
    """GATTrimapNet state_dict keys in the order oracle/gat.c expects."""
    keys = [f"in_norm.norm.{k}" for k in ("weight", "bias", "running_mean", "running_var")]
    keys += ["input_proj.0.weight", "input_proj.0.bias", "input_proj.1.weight", "input_proj.1.bias"]
    for i in range(n_layers):
        keys += [f"convs.{i}.att", f"convs.{i}.lin_l.weight", f"convs.{i}.lin_l.bias", f"convs.{i}.lin_r.weight", f"convs.{i}.lin_r.bias",
                 f"convs.{i}.lin_edge.weight", f"convs.{i}.bias", f"lns.{i}.weight", f"lns.{i}.bias",
                 f"edge_gates.{i}.proj.0.weight", f"edge_gates.{i}.proj.0.bias", f"edge_gates.{i}.proj.2.weight", f"edge_gates.{i}.proj.2.bias"]
    keys += ["skip_proj.weight", "ctx.attn.weight", "ctx.attn.bias", "ctx.compress.weight", "ctx.compress.bias",
             "ctx.expand.weight", "ctx.expand.bias", "head.0.weight", "head.0.bias", "head.3.weight", "head.3.bias"]
    return keys


def gat_forward(state: dict, hidden: int, n_layers: int, x, edge_index, edge_attr, batch=None, heads: int = 8):
    """GATTrimapNet (eval; heads in {1, 2, 4, 8}). state: {key: np.ndarray}. Returns (logits, probs) float32 (N,3)."""
    L = lib()
    keys = gat_param_order(n_layers)
    assert L.ggo_gat_n_params(n_layers) == len(keys)
    arrs = [f32(np.asarray(state[k])) for k in keys]
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    x = f32(x)
    ei = np.ascontiguousarray(edge_index, dtype=np.int64)
    ea = f32(edge_attr)
    n, e = x.shape[0], ei.shape[1]
    b = None if batch is None else np.ascontiguousarray(batch, dtype=np.int64)
    ng = 1 if b is None else int(b.max()) + 1
    logits = np.empty((n, 3), np.float32)
    probs = np.empty((n, 3), np.float32)
    rc = L.ggo_gat_forward(ptrs, _i(hidden), _i(heads), _i(n_layers), _i(n), _i(e), _p(x), _p(ei), _p(ea), _p(b), _i(ng), _p(logits), _p(probs))
    assert rc == 0, rc
    return logits, probs


def gcn_aggregate(xw, edge_index, bias=None, gate=None, h=None):
    L = lib()
    xw = f32(xw)
    n, d = xw.shape
    ei = np.ascontiguousarray(edge_index, dtype=np.int64)
    out = np.empty_like(xw)
    L.ggo_gcn_aggregate(_i(n), _i(ei.shape[1]), _i(d), _p(xw), _p(ei),
                        _p(None if bias is None else f32(bias)),
                        _p(None if gate is None else f32(gate)),
                        _p(None if h is None else f32(h)), _p(out))
    return out


def gcn_conv(x, edge_index, W, bias):
    L = lib()
    x = f32(x)
    n, d = x.shape
    ei = np.ascontiguousarray(edge_index, dtype=np.int64)
    out = np.empty_like(x)
    L.ggo_gcn_conv(_i(n), _i(ei.shape[1]), _i(d), _p(x), _p(ei), _p(f32(W)), _p(f32(bias)), _p(out))
    return out


# ---------------------------------------------------------------- G0 / G1

def preprocess(bgr):
    """-> lab, hsv (H,W,3) f32, gray, grad (H,W) f32."""
    L = lib()
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    h, w = bgr.shape[:2]
    lab = np.empty((h, w, 3), np.float32)
    hsv = np.empty((h, w, 3), np.float32)
    gray = np.empty((h, w), np.float32)
    grad = np.empty((h, w), np.float32)
    L.ggo_preprocess(_i(h), _i(w), _p(bgr), _p(lab), _p(hsv), _p(gray), _p(grad))
    return lab, hsv, gray, grad


def slic_rescale_lab(image, rescale_input=True):
    L = lib()
    image = f32(image)
    h, w = image.shape[:2]
    out = np.empty_like(image)
    L.ggo_slic_rescale_lab(_i(h), _i(w), _p(image), _i(int(rescale_input)), _p(out))
    return out


def gaussian(image, sigma=1.0):
    L = lib()
    image = f32(image)
    h, w, c = image.shape
    out = np.empty_like(image)
    L.ggo_gaussian_f32(_i(h), _i(w), _i(c), _p(image), _d(sigma), _p(out))
    return out


def slic_grid(h, w, n_segments):
    L = lib()
    v = [C.c_int() for _ in range(6)]
    k = L.ggo_slic_grid(_i(h), _i(w), _i(n_segments), *[C.byref(x) for x in v])
    sy, sx, y0, x0, ny, nx = [x.value for x in v]
    return dict(K=k, step_y=sy, step_x=sx, start_y=y0, start_x=x0, ny=ny, nx=nx)


def slic_kmeans(image, seeds_yx, step, max_iter=10):
    L = lib()
    image = f32(image)
    h, w = image.shape[:2]
    k = seeds_yx.shape[0]
    centers = np.zeros((k, 5), np.float32)
    centers[:, :2] = seeds_yx
    labels = np.empty((h, w), np.int32)
    L.ggo_slic_kmeans(_i(h), _i(w), _p(image), _i(k), _p(centers), _f(step), _i(max_iter), _p(labels))
    return labels, centers


def slic_connectivity(labels, min_size, max_size):
    L = lib()
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    h, w = labels.shape
    out = np.empty_like(labels)
    n = L.ggo_slic_connectivity(_i(h), _i(w), _p(labels), _i(int(min_size)), _i(int(max_size)), _p(out))
    return out, n


def slic(image, n_segments, compactness=10.0, sigma=1.0, rescale_input=True):
    L = lib()
    image = f32(image)
    h, w = image.shape[:2]
    seg = np.empty((h, w), np.int32)
    n = L.ggo_slic(_i(h), _i(w), _p(image), _i(n_segments), _f(compactness), _f(sigma), _i(int(rescale_input)), _p(seg))
    return seg, n


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def slic_rescale_lab64(image, rescale_input=True):
    """float64 instance (use_lab=False): min-max rescale + rgb2lab in double"""
    image = f64(image)
    h, w = image.shape[:2]
    out = np.empty_like(image)
    lib().ggo_slic_rescale_lab64(_i(h), _i(w), _p(image), _i(int(rescale_input)), _p(out))
    return out


def gaussian64(image, sigma=1.0):
    image = f64(image)
    h, w, c = image.shape
    out = np.empty_like(image)
    lib().ggo_gaussian_f64(_i(h), _i(w), _i(c), _p(image), _d(sigma), _p(out))
    return out


def slic_kmeans64(image, seeds_yx, step, max_iter=10):
    image = f64(image)
    h, w = image.shape[:2]
    k = seeds_yx.shape[0]
    centers = np.zeros((k, 5), np.float64)
    centers[:, :2] = seeds_yx
    labels = np.empty((h, w), np.int32)
    lib().ggo_slic_kmeans64(_i(h), _i(w), _p(image), _i(k), _p(centers), _d(step), _i(max_iter), _p(labels))
    return labels, centers


def slic_rgb(bgr, n_segments, compactness=10.0, sigma=1.0):
    """SuperpixelGraphConfig(use_lab=False): slic(rgb.astype(float), ...) — reference graph_builder.py:177-188"""
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    h, w = bgr.shape[:2]
    seg = np.empty((h, w), np.int32)
    n = lib().ggo_slic_rgb(_i(h), _i(w), _p(bgr), _i(n_segments), _d(compactness), _d(sigma), _p(seg))
    return seg, n


# ---------------------------------------------------------------- G2-G8

def find_boundaries_inner(seg):
    L = lib()
    seg = np.ascontiguousarray(seg, dtype=np.int32)
    h, w = seg.shape
    out = np.empty((h, w), np.uint8)
    L.ggo_find_boundaries_inner(_i(h), _i(w), _p(seg), _p(out))
    return out


def graph_build(segments, lab, hsv, grad, connectivity=4, n_nonlocal=4):
    """-> dict(node_features, prior, centroids, area_ratio, edge_index (2,E) i64, edge_attr (E,5))."""
    L = lib()
    seg = np.ascontiguousarray(segments, dtype=np.int32)
    h, w = seg.shape
    lab, hsv, grad = f32(lab), f32(hsv), f32(grad)
    n, e = C.c_int(), C.c_int()
    g = L.ggo_graph_build(_i(h), _i(w), _p(seg), _p(lab), _p(hsv), _p(grad), _i(connectivity), _i(n_nonlocal),
                          C.byref(n), C.byref(e))
    g = C.c_void_p(g)
    n, e = n.value, e.value
    out = dict(node_features=np.empty((n, 16), np.float32), prior=np.empty((n, 3), np.float32),
               centroids=np.empty((n, 2), np.float32), area_ratio=np.empty(n, np.float32),
               edge_index=np.empty((2, e), np.int64), edge_attr=np.empty((e, 5), np.float32))
    L.ggo_graph_get(g, _p(out["node_features"]), _p(out["prior"]), _p(out["centroids"]), _p(out["area_ratio"]),
                    _p(out["edge_index"]), _p(out["edge_attr"]))
    L.ggo_graph_free(g)
    out["n_nodes"], out["n_edges"] = n, e
    return out


def auto_prior(segments, lab, centre_sigma=0.45, contrast_sigma=0.40):
    L = lib()
    seg = np.ascontiguousarray(segments, dtype=np.int32)
    h, w = seg.shape
    n = int(seg.max()) + 1
    out = np.empty((n, 3), np.float32)
    L.ggo_auto_prior_sigmas(_i(h), _i(w), _p(seg), _p(f32(lab)), _i(n), _d(centre_sigma), _d(contrast_sigma), _p(out))
    return out


# ---------------------------------------------------------------- P0-P3, S0

def box_blur(img, radius):
    L = lib()
    img = f32(img)
    h, w = img.shape
    out = np.empty_like(img)
    L.ggo_box_blur(_i(h), _i(w), _p(img), _i(radius), _p(out))
    return out


def guided_filter(guide, src, radius=8, eps=1e-3):
    L = lib()
    guide, src = f32(guide), f32(src)
    h, w = guide.shape
    out = np.empty_like(guide)
    L.ggo_guided_filter(_i(h), _i(w), _p(guide), _p(src), _i(radius), _f(eps), _p(out))
    return out


def refine_trimap(probs, segments, bgr, thr_fg=0.55, thr_bg=0.55, radius=8, eps=1e-3, edge_aware=True):
    L = lib()
    probs = f32(probs)
    seg = np.ascontiguousarray(segments, dtype=np.int32)
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    h, w = seg.shape
    out = np.empty((h, w), np.uint8)
    L.ggo_refine_trimap(_i(h), _i(w), _p(probs), _i(probs.shape[0]), _p(seg), _p(bgr), _f(thr_fg), _f(thr_bg),
                        _i(radius), _f(eps), _i(int(edge_aware)), _p(out))
    return out


def seed_from_prior(trimap, prior, segments, seed_frac=0.1):
    L = lib()
    out = np.ascontiguousarray(trimap, dtype=np.uint8).copy()
    prior = f32(prior)
    seg = np.ascontiguousarray(segments, dtype=np.int32)
    h, w = seg.shape
    L.ggo_seed_from_prior(_i(h), _i(w), _p(prior), _i(prior.shape[0]), _p(seg), _d(seed_frac), _p(out))
    return out


# ---------------------------------------------------------------- C0-C6, K0, O0, R0

def grid_maxflow(tw, nw):
    """tw (H,W) int32 source-minus-sink; nw (4,H,W) int32 [left, up-left, up, up-right].
    -> (flow value, source_side (H,W) uint8)."""
    L = lib()
    tw = np.ascontiguousarray(tw, dtype=np.int32)
    nw = np.ascontiguousarray(nw, dtype=np.int32)
    h, w = tw.shape
    side = np.empty((h, w), np.uint8)
    L.ggo_grid_maxflow.argtypes = [_i, _i, _vp, _vp, _vp]
    flow = L.ggo_grid_maxflow(_i(h), _i(w), _p(tw), _p(nw), _p(side))
    return int(flow), side


def grabcut(image, mask, n_iter=5, mode=0, rect=None, seed=0, bgd=None, fgd=None):
    """-> (binary (H,W) u8, mask out, bgd_model, fgd_model, rc) ; rc 1 = degenerate trimap."""
    L = lib()
    image = np.ascontiguousarray(image, dtype=np.uint8)
    h, w = image.shape[:2]
    m = np.ascontiguousarray(mask, dtype=np.uint8).copy() if mask is not None else np.zeros((h, w), np.uint8)
    r = None if rect is None else np.ascontiguousarray(rect, dtype=np.int32)
    bgd = np.zeros(65) if bgd is None else np.ascontiguousarray(bgd, dtype=np.float64).copy().ravel()
    fgd = np.zeros(65) if fgd is None else np.ascontiguousarray(fgd, dtype=np.float64).copy().ravel()
    binary = np.empty((h, w), np.uint8)
    rc = L.ggo_grabcut(_i(h), _i(w), _p(image), _p(m), _p(r), _p(bgd), _p(fgd), _i(n_iter), _i(mode), _u64(seed),
                       _p(binary))
    return binary, m, bgd, fgd, rc


def clean_mask(mask, min_area_ratio=0.002, keep_largest=False):
    L = lib()
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    h, w = mask.shape
    out = np.empty_like(mask)
    L.ggo_clean_mask(_i(h), _i(w), _p(mask), _f(min_area_ratio), _i(int(keep_largest)), _p(out))
    return out


def compose(bgr, binary, alpha=0.45, tint_bgr=(100, 220, 0)):
    L = lib()
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    binary = np.ascontiguousarray(binary, dtype=np.uint8)
    h, w = binary.shape
    overlay = np.empty((h, w, 3), np.uint8)
    rgba = np.empty((h, w, 4), np.uint8)
    L.ggo_compose(_i(h), _i(w), _p(bgr), _p(binary), _f(alpha), _i(tint_bgr[0]), _i(tint_bgr[1]), _i(tint_bgr[2]),
                  _p(overlay), _p(rgba))
    return overlay, rgba


def iou(pred, gt):
    L = lib()
    pred = np.ascontiguousarray(pred, dtype=np.uint8)
    gt = np.ascontiguousarray(gt, dtype=np.uint8)
    return float(L.ggo_iou(_i(pred.size), _p(pred), _p(gt)))


def convert_color8(bgr, mode):
    """uint8 HSV (mode "hsv") or Lab ("lab") of a BGR uint8 image (oracle/color.c; parity with OpenCV unpinned)."""
    L = lib()
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    out = np.empty_like(bgr)
    L.ggo_convert_color8(C.c_size_t(bgr.size // 3), _p(bgr), _i({"hsv": 0, "lab": 1}[mode]), _p(out))
    return out


def eval_counts(pred, gt, trimap=None, width=3):
    """int64[14] tallies behind evaluate / boundary_f1 / evaluate_trimap (see oracle/postproc.c)."""
    L = lib()
    pred = np.ascontiguousarray(pred, dtype=np.uint8)
    gt = np.ascontiguousarray(gt, dtype=np.uint8)
    tri = None if trimap is None else np.ascontiguousarray(trimap, dtype=np.uint8)
    out = np.zeros(14, np.int64)
    h, w = pred.shape
    L.ggo_eval_counts(_i(h), _i(w), _p(pred), _p(gt), _p(tri), _i(width), _p(out))
    return out


# ---------------------------------------------------------------- the whole path

def segment(bgr, state, hidden, n_layers, n_segments=300, compactness=10.0, sigma=1.0, connectivity=4,
            n_nonlocal=4, threshold_fg=0.55, threshold_bg=0.55, n_iter=5, refine_iters=0, min_area_ratio=0.002,
            keep_largest=False, edge_aware=True, filter_radius=8, seed=0, timing=None, use_lab=True):
    """CPU restatement of GCNGrabCutPipeline.segment (reference pipeline.py:265-352), stage by stage.
    `state` is the ResGCNNet state_dict as numpy arrays.  Returns a dict of host arrays."""
    import time
    t0 = time.perf_counter()
    lab, hsv, gray, grad = preprocess(bgr)
    seg, n = slic(lab, n_segments, compactness, sigma, True) if use_lab else slic_rgb(bgr, n_segments, compactness, sigma)
    g = graph_build(seg, lab, hsv, grad, connectivity, n_nonlocal)
    t1 = time.perf_counter()
    x = np.concatenate([g["node_features"], g["prior"]], 1)
    logits, probs = resgcn_forward(state, hidden, n_layers, x, g["edge_index"], g["edge_attr"])
    trimap = refine_trimap(probs, seg, bgr, threshold_fg, threshold_bg, filter_radius, 1e-3, edge_aware)
    t2 = time.perf_counter()
    trimap = seed_from_prior(trimap, g["prior"], seg, 0.1)
    binary, mask, bgd, fgd, rc = grabcut(bgr, trimap, n_iter, 0, None, seed)
    if refine_iters > 0 and rc == 0:
        binary, mask, bgd, fgd, _ = grabcut(bgr, mask, refine_iters, 2, None, seed, bgd, fgd)
    t3 = time.perf_counter()
    cleaned = clean_mask(binary, min_area_ratio, keep_largest)
    overlay, rgba = compose(bgr, cleaned)
    t4 = time.perf_counter()
    if timing is not None:
        timing.update(graph_build=t1 - t0, gcn_inference=t2 - t1, grabcut=t3 - t2, postprocess=t4 - t3)
    return dict(segments=seg, n_nodes=n, graph=g, x=x, logits=logits, probs=probs, trimap=trimap,
                binary_mask=cleaned, gc_mask=mask, overlay=overlay, rgba=rgba)
