"""
Trimap network — host mirror of reference src/gcn_grabcut/model.py.

`ResGCNNet` keeps the reference's constructor, state_dict keys (model.py:449-499,
SURVEY section 8 row M0) and methods (`forward`, `predict_probs`, `predict_trimap`,
`layer_weights`, `param_groups`), but owns no arithmetic: the torch submodules
below are parameter containers only, and `forward` hands raw device pointers
to libggc_hip.so (ggc_resgcn_forward), whose kernels implement model.py:508-536
on the MI355X.  Inference only (eval mode); there is no CPU path.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

try:
    import torch
    import torch.nn as nn
    _TORCH = True
except ImportError:  # pragma: no cover
    _TORCH = False

import itertools

from ._constants import N_NODE_FEATS, N_EDGE_FEATS, N_PRIOR_FEATS

_model_uid = itertools.count(1)

TRIMAP_BG = 0        # cv2.GC_BGD
TRIMAP_FG = 1        # cv2.GC_FGD
TRIMAP_PROB_BG = 2   # cv2.GC_PR_BGD
TRIMAP_PROB_FG = 3   # cv2.GC_PR_FGD

CLASS_BG = 0
CLASS_UNK = 1
CLASS_FG = 2


if _TORCH:
    from . import _native

    class _InputNorm(nn.Module):
        """Parameter holder for reference InputNorm (model.py:191-213)."""
        def __init__(self, n_features: int, momentum: float = 0.05):
            super().__init__()
            self.norm = nn.BatchNorm1d(n_features, momentum=momentum, affine=True)

    class _EdgeContext(nn.Module):
        """Parameter holder for reference EdgeContext (model.py:111-139)."""
        def __init__(self, edge_dim: int, hidden_dim: int, ctx_dim: Optional[int] = None):
            super().__init__()
            ctx_dim = ctx_dim or max(hidden_dim // 2, 8)
            self.encode = nn.Sequential(nn.Linear(edge_dim, ctx_dim), nn.GELU(), nn.Linear(ctx_dim, ctx_dim))
            self.to_gate = nn.Sequential(nn.LayerNorm(ctx_dim), nn.Linear(ctx_dim, hidden_dim), nn.Sigmoid())

    class _GlobalContext(nn.Module):
        """Parameter holder for reference GlobalContextModule (model.py:165-188)."""
        def __init__(self, hidden_dim: int):
            super().__init__()
            self.attn = nn.Linear(hidden_dim, 1)
            self.compress = nn.Linear(hidden_dim, hidden_dim // 2)
            self.expand = nn.Linear(hidden_dim // 2, hidden_dim)

    class _GCNConvParams(nn.Module):
        """state_dict layout of PyG GCNConv(D, D): `lin.weight` [D,D], `bias` [D]."""
        def __init__(self, dim: int):
            super().__init__()
            self.lin = nn.Linear(dim, dim, bias=False)
            self.bias = nn.Parameter(torch.zeros(dim))

    class _SAGEConvParams(nn.Module):
        """state_dict layout of PyG SAGEConv(D, D): `lin_l.{weight,bias}`, `lin_r.weight`."""
        def __init__(self, dim: int):
            super().__init__()
            self.lin_l = nn.Linear(dim, dim, bias=True)
            self.lin_r = nn.Linear(dim, dim, bias=False)

    class ResGCNNet(nn.Module):
        """
        Residual GCN with jumping-knowledge fusion (reference model.py:421-590),
        executed by hand-written gfx950 kernels.

        InputNorm -> InputProj -> PriorBooster -> [ResBlock x n_layers] ->
        SAGEConv -> JK fusion -> GlobalContext -> Head
        """

        def __init__(
            self,
            in_channels: int = N_NODE_FEATS,
            edge_channels: int = N_EDGE_FEATS,
            hidden_channels: int = 128,
            n_layers: int = 6,
            n_classes: int = 3,
            dropout: float = 0.15,
        ):
            super().__init__()
            if in_channels != N_NODE_FEATS or edge_channels != N_EDGE_FEATS or n_classes != 3:
                raise ValueError(
                    "the MI355X kernels are specialised for the reference's fixed widths: "
                    f"in_channels={N_NODE_FEATS}, edge_channels={N_EDGE_FEATS}, n_classes=3"
                )
            if not 8 <= int(hidden_channels) <= 128:
                raise ValueError("hidden_channels must lie in [8, 128] (the HIP kernels are built for widths up to 128; widths that "
                                 "are not a multiple of 32 run zero-padded inside the library)")
            self.n_classes = n_classes
            self.n_layers = n_layers
            self.hidden_channels = hidden_channels
            D = hidden_channels

            self.in_norm = _InputNorm(in_channels)
            self.input_proj = nn.Sequential(nn.Linear(in_channels, D), nn.LayerNorm(D), nn.GELU())
            self.prior_booster = nn.Sequential(
                nn.Linear(N_PRIOR_FEATS, max(D // 4, 8)), nn.GELU(),
                nn.Linear(max(D // 4, 8), D), nn.Sigmoid(),
            )
            self.edge_ctx = _EdgeContext(edge_channels, D)
            self.gcn_layers = nn.ModuleList(_GCNConvParams(D) for _ in range(n_layers))
            self.norms = nn.ModuleList(nn.LayerNorm(D) for _ in range(n_layers))
            self.sage = _SAGEConvParams(D)
            self.sage_norm = nn.LayerNorm(D)
            self.jk_logits = nn.Parameter(torch.zeros(n_layers + 2))
            self.ctx = _GlobalContext(D)
            self.fuse = nn.Sequential(nn.LayerNorm(D), nn.Linear(D, D), nn.GELU(), nn.Dropout(dropout))
            self.head = nn.Linear(D, n_classes)
            self.dropout = dropout
            self._init_weights()
            self._uid = next(_model_uid)   # identifies this model in a context's record of resident weights

        def _init_weights(self):
            # reference model.py:501-506
            for m in self.modules():
                if isinstance(m, nn.Linear):
                    nn.init.kaiming_normal_(m.weight, nonlinearity="relu")
                    if m.bias is not None:
                        nn.init.zeros_(m.bias)

        # ---------------------------------------------------------------- native

        def _device_index(self) -> int:
            dev = self.jk_logits.device
            if dev.type != "cuda":
                raise RuntimeError(
                    "ResGCNNet runs on an MI355X through libggc_hip.so only; "
                    f"the model is on '{dev}'. Move it with .to('cuda') — there is no CPU fallback."
                )
            return dev.index if dev.index is not None else torch.cuda.current_device()

        def _sync_weights(self, ctx: "_native.Context") -> None:
            sd = self.state_dict()
            # the record lives on the CONTEXT: another model may have replaced this one's weights there since
            fp = (self._uid, tuple((k, v.data_ptr(), v._version) for k, v in sd.items()))
            if ctx.resident.get("resgcn") == fp:
                return
            ctx.call("ggc_resgcn_configure", self.hidden_channels, self.n_layers)
            for k, v in sd.items():
                if not v.dtype.is_floating_point:
                    continue   # num_batches_tracked
                a = v.detach().to(device="cpu", dtype=torch.float32).contiguous().numpy()
                ctx.call("ggc_resgcn_load_weight", k.encode(), a.ctypes.data, a.size)
            ctx.call("ggc_resgcn_ready")
            ctx.resident["resgcn"] = fp

        def _run(self, data, want_logits: bool, want_probs: bool, ctx=None):
            if self.training:
                raise RuntimeError("ResGCNNet on MI355X is inference-only: call .eval() first "
                                   "(training lives in the reference and is out of scope here)")
            dev_index = self._device_index()
            if ctx is None:                      # a pipeline replica passes its private context (own scratch arena)
                ctx = _native.get_context(dev_index)
            self._sync_weights(ctx)
            dev = torch.device("cuda", dev_index)

            x = data.x
            if x.device != dev:
                raise RuntimeError(f"data.x is on {x.device}, model on {dev}")
            x = x.to(torch.float32).contiguous()
            n = x.size(0)
            if x.dim() != 2 or x.size(1) != N_NODE_FEATS:
                raise ValueError(f"data.x must be (N, {N_NODE_FEATS}), got {tuple(x.shape)}")
            ei = data.edge_index
            e = ei.size(1)
            edge_attr = getattr(data, "edge_attr", None)
            if edge_attr is None:                      # reference model.py:511-512
                edge_attr = torch.zeros(e, N_EDGE_FEATS, device=dev)
            edge_attr = edge_attr.to(torch.float32).contiguous()
            src = ei[0].to(torch.int32).contiguous()
            dst = ei[1].to(torch.int32).contiguous()

            batch = getattr(data, "batch", None)
            node_ptr = getattr(data, "node_ptr32", None)
            if node_ptr is None:
                if batch is None:
                    node_ptr = torch.tensor([0, n], dtype=torch.int32, device=dev)
                else:
                    n_graphs = getattr(data, "num_graphs", None)
                    if n_graphs is None:
                        n_graphs = int(batch.max().item()) + 1   # reference model.py:86
                    counts = torch.bincount(batch, minlength=n_graphs)
                    node_ptr = torch.zeros(n_graphs + 1, dtype=torch.int32, device=dev)
                    node_ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
            g = node_ptr.numel() - 1

            logits = torch.empty(n, 3, dtype=torch.float32, device=dev) if want_logits else None
            probs = torch.empty(n, 3, dtype=torch.float32, device=dev) if want_probs else None
            ctx.call(
                "ggc_resgcn_forward", _native.current_stream(dev_index), g, n, e,
                x.data_ptr(), src.data_ptr(), dst.data_ptr(), edge_attr.data_ptr(), node_ptr.data_ptr(),
                _native.ptr(logits), _native.ptr(probs),
            )
            return logits, probs

        def forward(self, data) -> "torch.Tensor":
            """logits (N, 3) on the model's device — reference model.py:508-536."""
            with torch.no_grad():
                return self._run(data, True, False)[0]

        @torch.no_grad()
        def layer_weights(self) -> np.ndarray:
            """Fusion weights over [input, block 1..n, SAGE branch] (model.py:538-541)."""
            return torch.softmax(self.jk_logits.detach(), dim=0).cpu().numpy()

        @torch.no_grad()
        def predict_probs(self, data) -> np.ndarray:
            """softmax(logits) as a host array (model.py:543-546)."""
            self.eval()
            return self._run(data, False, True)[1].float().cpu().numpy()

        @torch.no_grad()
        def predict_probs_device(self, data, ctx=None) -> "torch.Tensor":
            """Additive: like predict_probs but the result stays in HBM (ctx: library context to run in)."""
            if self.training:
                self.eval()
            return self._run(data, False, True, ctx)[1]

        @torch.no_grad()
        def predict_trimap(self, data, segments: np.ndarray,
                           threshold_fg: float = 0.55, threshold_bg: float = 0.55) -> np.ndarray:
            return _probs_to_trimap(self.predict_probs(data), segments, threshold_fg, threshold_bg)

        def param_groups(self, base_lr: float) -> list[dict]:
            """Layer-wise learning-rate decay groups (model.py:559-590); kept for API parity."""
            groups = []
            n = self.n_layers
            for i, (gcn, norm) in enumerate(zip(self.gcn_layers, self.norms)):
                groups.append({"params": list(gcn.parameters()) + list(norm.parameters()),
                               "lr": base_lr * (0.8 ** (n - i))})
            groups.append({"params": (list(self.in_norm.parameters()) + list(self.input_proj.parameters()) +
                                      list(self.prior_booster.parameters())), "lr": base_lr * 0.5})
            groups.append({"params": (list(self.edge_ctx.parameters()) + list(self.sage.parameters()) +
                                      list(self.sage_norm.parameters()) + list(self.ctx.parameters())),
                           "lr": base_lr * 0.9})
            groups.append({"params": ([self.jk_logits] + list(self.fuse.parameters()) +
                                      list(self.head.parameters())), "lr": base_lr})
            return groups

    class _EdgeInjection(nn.Module):
        """Parameter holder for reference EdgeInjectionLayer (model.py:142-162)."""
        def __init__(self, edge_dim: int, hidden_dim: int):
            super().__init__()
            self.proj = nn.Sequential(nn.Linear(edge_dim, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, hidden_dim), nn.Sigmoid())

    class _ResGCNBlock(nn.Module):
        """Parameter holder for reference ResGCNBlock with in_dim == out_dim (model.py:216-232): `skip` is the identity."""
        def __init__(self, dim: int, edge_dim: int):
            super().__init__()
            self.conv = _GCNConvParams(dim)
            self.bn = nn.BatchNorm1d(dim)
            self.edge_inject = _EdgeInjection(edge_dim, dim)

    class GCNTrimapNet(nn.Module):
        """
        Baseline GCN with residual blocks, per-block edge injection and a dense-concat head (reference
        model.py:239-316).  Same `state_dict` keys as the reference module, so a
        reference checkpoint loads with `load_state_dict`.  Inference only: the forward pass runs in libggc_hip.so
        (`ggc_gcnnet_forward`) on an MI355X — there is no CPU fallback.
        """

        def __init__(self, in_channels: int = N_NODE_FEATS, edge_channels: int = N_EDGE_FEATS, hidden_channels: int = 128,
                     n_layers: int = 6, n_classes: int = 3, dropout: float = 0.2):
            super().__init__()
            if in_channels != N_NODE_FEATS or edge_channels != N_EDGE_FEATS or n_classes != 3:
                raise ValueError("the MI355X kernels are built for 19 node features, 5 edge features and 3 classes")
            if hidden_channels % 2 or not 2 <= hidden_channels <= 128:
                raise ValueError("hidden_channels must be even and at most 128 (the HIP kernels are built for widths up to 128)")
            self.n_classes, self.hidden_channels, self.n_layers = n_classes, hidden_channels, n_layers
            # The MFMA tiling needs a multiple of 32: other widths (the reference's tests use 16) run zero-padded to
            # the next multiple.  Padded channels carry exact zeros through every layer (zero weight rows / columns,
            # identity BatchNorm statistics), so the result is the unpadded model's, bit for bit.
            self._kernel_width = -(-hidden_channels // 32) * 32
            self.in_norm = _InputNorm(in_channels)
            self.input_proj = nn.Sequential(nn.Linear(in_channels, hidden_channels), nn.BatchNorm1d(hidden_channels), nn.ReLU())
            self.blocks = nn.ModuleList([_ResGCNBlock(hidden_channels, edge_channels) for _ in range(n_layers)])
            self.head = nn.Sequential(
                nn.Linear(hidden_channels * (n_layers + 1), hidden_channels), nn.BatchNorm1d(hidden_channels), nn.ReLU(),
                nn.Dropout(dropout), nn.Linear(hidden_channels, hidden_channels // 2), nn.ReLU(),
                nn.Linear(hidden_channels // 2, n_classes))
            self._uid = next(_model_uid)

        def _device_index(self) -> int:
            dev = self.head[0].weight.device
            if dev.type != "cuda":
                raise RuntimeError("GCNTrimapNet runs on an MI355X through libggc_hip.so only; "
                                   f"the model is on '{dev}'. Move it with .to('cuda') — there is no CPU fallback.")
            return dev.index if dev.index is not None else torch.cuda.current_device()

        def _sync_weights(self, ctx: "_native.Context") -> None:
            sd = self.state_dict()
            fp = (self._uid, tuple((k, v.data_ptr(), v._version) for k, v in sd.items()))
            if ctx.resident.get("gcnnet") == fp:
                return
            ctx.call("ggc_gcnnet_configure", self._kernel_width, self.n_layers)
            for k, v in sd.items():
                if not v.dtype.is_floating_point:
                    continue   # num_batches_tracked
                a = self._padded(k, v.detach().to(device="cpu", dtype=torch.float32).contiguous().numpy())
                ctx.call("ggc_gcnnet_load_weight", k.encode(), a.ctypes.data, a.size)
            ctx.call("ggc_gcnnet_ready")
            ctx.resident["gcnnet"] = fp

        def _padded(self, key: str, a: np.ndarray) -> np.ndarray:
            """Tensor `key` at the kernels' width: zero rows / columns, and BatchNorm statistics (mean 0, var 1, weight 1,
            bias 0) that map the padded zeros to zeros."""
            d, w, n = self.hidden_channels, self._kernel_width, self.n_layers
            if d == w or key.startswith("in_norm."):
                return a

            def pad(x, shape, fill=0.0):
                out = np.full(shape, fill, np.float32)
                out[tuple(slice(0, s) for s in x.shape)] = x
                return np.ascontiguousarray(out)

            if key == "head.0.weight":                         # [d, d (n+1)]: one block per concatenated state
                out = np.zeros((w, w * (n + 1)), np.float32)
                for s in range(n + 1):
                    out[:d, s * w:s * w + d] = a[:, s * d:(s + 1) * d]
                return out
            if key == "head.4.weight": return pad(a, (w // 2, w))
            if key == "head.4.bias": return pad(a, (w // 2,))
            if key == "head.6.weight": return pad(a, (3, w // 2))
            if key == "head.6.bias": return a
            if key.endswith("running_var") or (a.ndim == 1 and key.endswith(".weight")):   # BatchNorm var / weight
                return pad(a, (w,), 1.0)
            if a.ndim == 1: return pad(a, (w,))
            if key.endswith("edge_inject.proj.0.weight") or key == "input_proj.0.weight": return pad(a, (w, a.shape[1]))
            return pad(a, (w, w))

        def _run(self, data, want_logits: bool, want_probs: bool, ctx=None):
            if self.training:
                raise RuntimeError("GCNTrimapNet on MI355X is inference-only: call .eval() first")
            dev_index = self._device_index()
            if ctx is None:
                ctx = _native.get_context(dev_index)
            self._sync_weights(ctx)
            dev = torch.device("cuda", dev_index)
            x = data.x
            if x.device != dev:
                raise RuntimeError(f"data.x is on {x.device}, model on {dev}")
            x = x.to(torch.float32).contiguous()
            n = x.size(0)
            if x.dim() != 2 or x.size(1) != N_NODE_FEATS:
                raise ValueError(f"data.x must be (N, {N_NODE_FEATS}), got {tuple(x.shape)}")
            ei = data.edge_index
            e = ei.size(1)
            edge_attr = getattr(data, "edge_attr", None)
            if edge_attr is None:                      # reference model.py:294-295
                edge_attr = torch.zeros(e, N_EDGE_FEATS, device=dev)
            edge_attr = edge_attr.to(torch.float32).contiguous()
            src, dst = ei[0].to(torch.int32).contiguous(), ei[1].to(torch.int32).contiguous()
            logits = torch.empty(n, 3, dtype=torch.float32, device=dev) if want_logits else None
            probs = torch.empty(n, 3, dtype=torch.float32, device=dev) if want_probs else None
            ctx.call("ggc_gcnnet_forward", _native.current_stream(dev_index), n, e, x.data_ptr(), src.data_ptr(), dst.data_ptr(),
                     edge_attr.data_ptr(), _native.ptr(logits), _native.ptr(probs))
            return logits, probs

        def forward(self, data) -> "torch.Tensor":
            """logits (N, 3) on the model's device — reference model.py:292-304."""
            with torch.no_grad():
                return self._run(data, True, False)[0]

        @torch.no_grad()
        def predict_probs(self, data) -> np.ndarray:
            self.eval()
            return self._run(data, False, True)[1].float().cpu().numpy()

        @torch.no_grad()
        def predict_probs_device(self, data, ctx=None) -> "torch.Tensor":
            if self.training:
                self.eval()
            return self._run(data, False, True, ctx)[1]

        @torch.no_grad()
        def predict_trimap(self, data, segments: np.ndarray, threshold_fg: float = 0.55, threshold_bg: float = 0.55) -> np.ndarray:
            return _probs_to_trimap(self.predict_probs(data), segments, threshold_fg, threshold_bg)


    def _node_ptr_of(data, n: int, dev) -> "torch.Tensor":
        """int32 prefix sums of the graphs' node counts from data.node_ptr32 / data.batch (contiguous graphs, PyG Batch)."""
        node_ptr = getattr(data, "node_ptr32", None)
        if node_ptr is not None:
            return node_ptr
        batch = getattr(data, "batch", None)
        if batch is None:
            return torch.tensor([0, n], dtype=torch.int32, device=dev)
        n_graphs = getattr(data, "num_graphs", None)
        if n_graphs is None:
            n_graphs = int(batch.max().item()) + 1                   # reference model.py:86
        counts = torch.bincount(batch, minlength=n_graphs)
        node_ptr = torch.zeros(n_graphs + 1, dtype=torch.int32, device=dev)
        node_ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
        return node_ptr

    class _GATv2Params(nn.Module):
        """Parameter holder with the state_dict keys of torch_geometric.nn.GATv2Conv(D, D // H, heads=H, concat=True,
        edge_dim=5, share_weights=False): att [1,H,C], lin_l / lin_r (with bias), lin_edge (no bias), bias [D]."""
        def __init__(self, dim: int, heads: int, edge_dim: int):
            super().__init__()
            c = dim // heads
            self.att = nn.Parameter(torch.empty(1, heads, c))
            self.lin_l = nn.Linear(dim, heads * c)
            self.lin_r = nn.Linear(dim, heads * c)
            self.lin_edge = nn.Linear(edge_dim, heads * c, bias=False)
            self.bias = nn.Parameter(torch.zeros(heads * c))
            nn.init.xavier_uniform_(self.att)                       # PyG's glorot

    class GATTrimapNet(nn.Module):
        """
        GATv2 attention variant with edge features (reference model.py:323-414; SURVEY.md section 8(f), last rank).  Same
        `state_dict` keys as the reference module.  Inference only: the forward pass runs in libggc_hip.so
        (`ggc_gat_forward`) on an MI355X — there is no CPU fallback.  Widths 32, 64 or 128 with 1, 2, 4 or 8 heads.
        """

        def __init__(self, in_channels: int = N_NODE_FEATS, edge_channels: int = N_EDGE_FEATS, hidden_channels: int = 128,
                     n_heads: int = 8, n_layers: int = 5, n_classes: int = 3, dropout: float = 0.2):
            super().__init__()
            if in_channels != N_NODE_FEATS or edge_channels != N_EDGE_FEATS or n_classes != 3:
                raise ValueError("the MI355X kernels are built for 19 node features, 5 edge features and 3 classes")
            if hidden_channels not in (32, 64, 128) or n_heads not in (1, 2, 4, 8):
                raise ValueError("GATTrimapNet on MI355X: hidden_channels in {32, 64, 128} with n_heads in {1, 2, 4, 8} "
                                 "(a head must span a power-of-two number of lanes; the reference's default is 128 x 8)")
            self.n_classes, self.n_heads, self.hidden_channels, self.n_layers = n_classes, n_heads, hidden_channels, n_layers
            self.in_norm = _InputNorm(in_channels)
            self.input_proj = nn.Sequential(nn.Linear(in_channels, hidden_channels), nn.LayerNorm(hidden_channels), nn.GELU())
            self.convs = nn.ModuleList([_GATv2Params(hidden_channels, n_heads, edge_channels) for _ in range(n_layers)])
            self.lns = nn.ModuleList([nn.LayerNorm(hidden_channels) for _ in range(n_layers)])
            self.edge_gates = nn.ModuleList([_EdgeInjection(edge_channels, hidden_channels) for _ in range(n_layers)])
            self.dropout = dropout
            self.skip_proj = nn.Linear(hidden_channels, hidden_channels, bias=False)
            self.ctx = _GlobalContext(hidden_channels)
            self.head = nn.Sequential(nn.Linear(hidden_channels, hidden_channels), nn.GELU(), nn.Dropout(dropout),
                                      nn.Linear(hidden_channels, n_classes))
            self._uid = next(_model_uid)

        def _device_index(self) -> int:
            dev = self.head[0].weight.device
            if dev.type != "cuda":
                raise RuntimeError("GATTrimapNet runs on an MI355X through libggc_hip.so only; "
                                   f"the model is on '{dev}'. Move it with .to('cuda') — there is no CPU fallback.")
            return dev.index if dev.index is not None else torch.cuda.current_device()

        def _sync_weights(self, ctx: "_native.Context") -> None:
            sd = self.state_dict()
            fp = (self._uid, tuple((k, v.data_ptr(), v._version) for k, v in sd.items()))
            if ctx.resident.get("gat") == fp:
                return
            ctx.call("ggc_gat_configure", self.hidden_channels, self.n_heads, self.n_layers)
            for k, v in sd.items():
                if not v.dtype.is_floating_point:
                    continue   # num_batches_tracked
                a = v.detach().to(device="cpu", dtype=torch.float32).contiguous().numpy()
                ctx.call("ggc_gat_load_weight", k.encode(), a.ctypes.data, a.size)
            ctx.call("ggc_gat_ready")
            ctx.resident["gat"] = fp

        def _run(self, data, want_logits: bool, want_probs: bool, ctx=None, check_loops: bool = True):
            if self.training:
                raise RuntimeError("GATTrimapNet on MI355X is inference-only: call .eval() first")
            # PyG's GATv2Conv(add_self_loops=True) REMOVES i -> i edges (and their attributes) before it adds its own
            # mean-filled loops; the kernels keep every input edge.  GraphBuilder never emits a loop, so the pipeline is
            # unaffected; a user-supplied edge_index that holds one is refused rather than answered differently.
            if check_loops and data.edge_index.numel() and bool((data.edge_index[0] == data.edge_index[1]).any()):
                raise ValueError("GATTrimapNet: edge_index holds self-loops (i -> i); PyG's GATv2Conv drops them before adding "
                                 "its own, which this build does not do — remove them (edge_attr rows included) first")
            dev_index = self._device_index()
            if ctx is None:
                ctx = _native.get_context(dev_index)
            self._sync_weights(ctx)
            dev = torch.device("cuda", dev_index)
            x = data.x
            if x.device != dev:
                raise RuntimeError(f"data.x is on {x.device}, model on {dev}")
            x = x.to(torch.float32).contiguous()
            n = x.size(0)
            if x.dim() != 2 or x.size(1) != N_NODE_FEATS:
                raise ValueError(f"data.x must be (N, {N_NODE_FEATS}), got {tuple(x.shape)}")
            ei = data.edge_index
            e = ei.size(1)
            edge_attr = getattr(data, "edge_attr", None)
            if edge_attr is None:                      # reference model.py:383-384
                edge_attr = torch.zeros(e, N_EDGE_FEATS, device=dev)
            edge_attr = edge_attr.to(torch.float32).contiguous()
            src, dst = ei[0].to(torch.int32).contiguous(), ei[1].to(torch.int32).contiguous()
            node_ptr = _node_ptr_of(data, n, dev)
            logits = torch.empty(n, 3, dtype=torch.float32, device=dev) if want_logits else None
            probs = torch.empty(n, 3, dtype=torch.float32, device=dev) if want_probs else None
            ctx.call("ggc_gat_forward", _native.current_stream(dev_index), node_ptr.numel() - 1, n, e, x.data_ptr(), src.data_ptr(),
                     dst.data_ptr(), edge_attr.data_ptr(), node_ptr.data_ptr(), _native.ptr(logits), _native.ptr(probs))
            return logits, probs

        def forward(self, data) -> "torch.Tensor":
            """logits (N, 3) on the model's device — reference model.py:380-404."""
            with torch.no_grad():
                return self._run(data, True, False)[0]

        @torch.no_grad()
        def predict_probs(self, data) -> np.ndarray:
            self.eval()
            return self._run(data, False, True)[1].float().cpu().numpy()

        @torch.no_grad()
        def predict_probs_device(self, data, ctx=None) -> "torch.Tensor":
            """the pipeline's entry: graphs straight from the graph stage (no loops by construction, no host sync to check)"""
            if self.training:
                self.eval()
            return self._run(data, False, True, ctx, check_loops=False)[1]

        @torch.no_grad()
        def predict_trimap(self, data, segments: np.ndarray, threshold_fg: float = 0.55, threshold_bg: float = 0.55) -> np.ndarray:
            return _probs_to_trimap(self.predict_probs(data), segments, threshold_fg, threshold_bg)

    def build_model(
        variant: str = "resgcn",
        in_channels: int = N_NODE_FEATS,
        edge_channels: int = N_EDGE_FEATS,
        hidden_channels: int = 128,
        n_layers: int = 6,
        n_classes: int = 3,
        dropout: float = 0.2,
    ) -> "nn.Module":
        """Factory by name (reference model.py:593-620): "resgcn" (hot path), "gcn" and "gat" all run on the MI355X."""
        if variant == "resgcn":
            return ResGCNNet(in_channels=in_channels, edge_channels=edge_channels,
                             hidden_channels=hidden_channels, n_layers=n_layers,
                             n_classes=n_classes, dropout=dropout)
        if variant == "gcn":
            return GCNTrimapNet(in_channels=in_channels, edge_channels=edge_channels, hidden_channels=hidden_channels,
                                n_layers=n_layers, n_classes=n_classes, dropout=dropout)
        if variant == "gat":
            # the reference's factory passes n_layers through unchanged (its class default is 5, the factory's 6)
            return GATTrimapNet(in_channels=in_channels, edge_channels=edge_channels, hidden_channels=hidden_channels,
                                n_layers=n_layers, n_classes=n_classes, dropout=dropout)
        raise ValueError(f"Unknown variant '{variant}'. Choose: resgcn | gcn | gat")


def probs_to_node_trimap(probs: np.ndarray, threshold_fg: float = 0.55,
                         threshold_bg: float = 0.55) -> np.ndarray:
    """Per-region GrabCut labels from class probabilities (reference model.py:623-645)."""
    bg_p, fg_p = probs[:, CLASS_BG], probs[:, CLASS_FG]
    labels = np.where(fg_p > bg_p, TRIMAP_PROB_FG, TRIMAP_PROB_BG).astype(np.uint8)
    labels[bg_p >= threshold_bg] = TRIMAP_BG
    labels[fg_p >= threshold_fg] = TRIMAP_FG
    return labels


def project_to_pixels(node_values: np.ndarray, segments: np.ndarray) -> np.ndarray:
    """values[segments] with zero padding for missing regions (reference model.py:648-661)."""
    n_needed = int(segments.max()) + 1
    values = node_values
    if values.shape[0] < n_needed:
        pad = np.zeros((n_needed - values.shape[0], *values.shape[1:]), dtype=values.dtype)
        values = np.concatenate([values, pad], axis=0)
    return values[segments]


def _probs_to_trimap(probs: np.ndarray, segments: np.ndarray,
                     threshold_fg: float, threshold_bg: float) -> np.ndarray:
    """Pixel trimap without the guided filter (reference model.py:664-678)."""
    node_labels = probs_to_node_trimap(probs, threshold_fg, threshold_bg)
    n_needed = int(segments.max()) + 1
    if node_labels.shape[0] < n_needed:
        node_labels = np.concatenate([
            node_labels, np.full(n_needed - node_labels.shape[0], TRIMAP_PROB_BG, dtype=np.uint8)])
    return node_labels[segments].astype(np.uint8)
