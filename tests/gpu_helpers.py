"""Thin wrappers that drive the C ABI stage by stage for the GPU parity tests."""
import ctypes as C

import numpy as np
import torch


def stream():
    from gcn_grabcut import _native
    return _native.current_stream(0)


def preprocess(ctx, bgr):
    bgr = torch.as_tensor(np.ascontiguousarray(bgr)).cuda().contiguous()
    b, h, w, _ = bgr.shape
    lab = torch.empty(b, h, w, 3, device="cuda")
    hsv = torch.empty(b, h, w, 3, device="cuda")
    gray = torch.empty(b, h, w, device="cuda")
    grad = torch.empty(b, h, w, device="cuda")
    ctx.call("ggc_preprocess", stream(), b, h, w, bgr.data_ptr(), lab.data_ptr(), hsv.data_ptr(), gray.data_ptr(),
             grad.data_ptr())
    return bgr, lab, hsv, gray, grad


def slic(ctx, lab, n_segments, compactness=10.0, sigma=1.0, rescale=1):
    lab = lab.contiguous()
    b, h, w, _ = lab.shape
    seg = torch.empty(b, h, w, dtype=torch.int32, device="cuda")
    n = torch.empty(b, dtype=torch.int32, device="cuda")
    ctx.call("ggc_slic", stream(), b, h, w, lab.data_ptr(), n_segments, compactness, sigma, rescale, seg.data_ptr(),
             n.data_ptr())
    return seg, n


def graph(ctx, seg, n_nodes, lab, hsv, grad, connectivity=4, n_nonlocal=4, global_ids=False):
    """-> dict with packed device tensors + host node_ptr / edge_ptr."""
    b, h, w = seg.shape
    node_ptr = np.zeros(b + 1, np.int64)
    edge_ptr = np.zeros(b + 1, np.int64)
    ctx.call("ggc_graph_count", stream(), b, h, w, seg.data_ptr(), n_nodes.data_ptr(), lab.data_ptr(), hsv.data_ptr(),
             grad.data_ptr(), connectivity, n_nonlocal, node_ptr.ctypes.data, edge_ptr.ctypes.data)
    n, e = int(node_ptr[-1]), int(edge_ptr[-1])
    out = dict(node_ptr=node_ptr, edge_ptr=edge_ptr,
               x=torch.empty(n, 19, device="cuda"), centroids=torch.empty(n, 2, device="cuda"),
               area=torch.empty(n, device="cuda"), src=torch.empty(max(e, 1), dtype=torch.int32, device="cuda"),
               dst=torch.empty(max(e, 1), dtype=torch.int32, device="cuda"), attr=torch.empty(max(e, 1), 5, device="cuda"))
    ctx.call("ggc_graph_fill", stream(), out["x"].data_ptr(), out["centroids"].data_ptr(), out["area"].data_ptr(),
             out["src"].data_ptr(), out["dst"].data_ptr(), out["attr"].data_ptr(), int(global_ids))
    return out
