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
        for name, res, args in (("ggo_cbrt", _d, [_d]), ("ggo_pow24", _d, [_d]), ("ggo_iou", _d, None),
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
