"""
Batched device engine: the hot path of GCNGrabCutPipeline.segment expressed as
calls into libggc_hip.so on tensors that stay in HBM.

Every public class of this package (GraphBuilder, GrabCut, refine_trimap, ...)
is a thin view over these methods with batch size 1; `segment_batch` uses them
with the whole batch.  PyTorch provides device memory and the stream only.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np
import torch

from . import _native


@dataclass
class DeviceGraphs:
    """Superpixel graphs of a batch, packed PyG-Batch style, resident in HBM."""
    segments: torch.Tensor      # (B,H,W) int32
    n_nodes: torch.Tensor       # (B,)   int32
    node_ptr_host: np.ndarray   # (B+1,) int64
    edge_ptr_host: np.ndarray   # (B+1,) int64
    node_ptr: torch.Tensor      # (B+1,) int32 on device
    x: torch.Tensor             # (N,19) float32 = 16 image features || 3 prior
    centroids: torch.Tensor     # (N,2)
    area_ratio: torch.Tensor    # (N,)
    edge_src: torch.Tensor      # (E,) int32, global ids
    edge_dst: torch.Tensor      # (E,) int32, global ids
    edge_attr: torch.Tensor     # (E,5)

    @property
    def batch_size(self) -> int:
        return self.segments.size(0)


class Engine:
    def __init__(self, device_index: int = 0, private_context: bool = False):
        if not torch.cuda.is_available():
            raise RuntimeError("gcn_grabcut needs an MI355X: no HIP device is visible and there is no CPU fallback")
        self.index = int(device_index)
        self.device = torch.device("cuda", self.index)
        # a private context owns its scratch arena, so it can run concurrently with the shared one on another stream
        self.ctx = _native.Context(self.index) if private_context else _native.get_context(self.index)
        self._lanes = None

    # ------------------------------------------------------------------ concurrent lanes
    def lanes(self, n: int):
        """n (engine, stream) pairs with private contexts for work that is independent per image.  GrabCut's
        max-flow ends in rounds with a handful of open images that are pure launch latency; running sub-batches
        on separate streams lets one sub-batch's tail overlap another's bandwidth-bound rounds.  The list only
        grows: streams keep their hardware queue, and two lanes that share a queue serialise."""
        if self._lanes is None:
            self._lanes = []
        while len(self._lanes) < n:
            self._lanes.append((Engine(self.index, private_context=True), torch.cuda.Stream(self.device)))
        return self._lanes[:n]

    def grabcut_lanes(self, image, mask, n_iter=5, mode=0, seed=0, n_lanes=4, bgd=None, fgd=None, post=None):
        """grabcut() on n_lanes contiguous sub-batches at once; same results (image b keeps seed + b).

        Sub-batch 0 runs on the caller's stream, the others on n_lanes - 1 streams of the engine.  HIP maps streams onto 4
        hardware queues by default (GPU_MAX_HW_QUEUES) and two lanes that share a queue serialise: the caller's stream plus
        three created ones are the four queues of a fresh process (53 ms per stage at batch 256); all four lanes on created
        streams are five streams on four queues (measured in round 3: 77 ms)."""
        from concurrent.futures import ThreadPoolExecutor
        b = image.size(0)
        n_lanes = max(1, min(int(n_lanes), b))
        # post(engine, lo, hi, binary[lo:hi]): per-image follow-up work (clean-up, composition) that a lane runs on its own
        # stream as soon as ITS sub-batch is cut, under the other lanes' tails, instead of after the slowest lane
        if n_lanes == 1:
            out = self.grabcut(image, mask, n_iter, mode, None, seed, bgd, fgd)
            if post is not None:
                post(self, 0, b, out[0])
            return out
        bounds = [b * i // n_lanes for i in range(n_lanes + 1)]
        if bgd is None:
            bgd = torch.zeros(b, 65, dtype=torch.float64, device=self.device)
        if fgd is None:
            fgd = torch.zeros(b, 65, dtype=torch.float64, device=self.device)
        binary = self.empty(b, *image.shape[1:3], dtype=torch.uint8)
        caller = torch.cuda.current_stream(self.device)
        lanes = self.lanes(n_lanes - 1)          # sub-batch 0 runs here, on the caller's stream and context

        def run(i):
            eng, stream = lanes[i - 1]
            lo, hi = bounds[i], bounds[i + 1]
            torch.cuda.set_device(self.device)
            stream.wait_stream(caller)
            with torch.cuda.stream(stream):
                out = eng.grabcut(image[lo:hi], mask[lo:hi], n_iter, mode, None, seed + lo, bgd[lo:hi], fgd[lo:hi])
                binary[lo:hi].copy_(out[0])
                if post is not None:
                    post(eng, lo, hi, binary[lo:hi])
            stream.synchronize()

        if getattr(self, "_pool", None) is None or self._pool._max_workers < n_lanes - 1:
            if getattr(self, "_pool", None) is not None:
                self._pool.shutdown(wait=True)
            self._pool = ThreadPoolExecutor(max_workers=n_lanes - 1, thread_name_prefix="ggc-lane")
        futures = [self._pool.submit(run, i) for i in range(1, n_lanes)]
        out = self.grabcut(image[:bounds[1]], mask[:bounds[1]], n_iter, mode, None, seed, bgd[:bounds[1]], fgd[:bounds[1]])
        binary[:bounds[1]].copy_(out[0])
        if post is not None:
            post(self, 0, bounds[1], binary[:bounds[1]])
        for f in futures:
            f.result()
        return binary, mask, bgd, fgd

    def all_contexts(self):
        return [self.ctx] + [e.ctx for e, _ in (self._lanes or [])]

    # ------------------------------------------------------------------ helpers
    def _stream(self) -> int:
        return _native.current_stream(self.index)

    def to_device(self, a, dtype=None) -> torch.Tensor:
        if torch.is_tensor(a):
            t = a
        else:
            t = torch.from_numpy(np.ascontiguousarray(a))
        if dtype is not None:
            t = t.to(dtype)
        return t.to(self.device, non_blocking=True).contiguous()

    def empty(self, *shape, dtype=torch.float32) -> torch.Tensor:
        return torch.empty(*shape, dtype=dtype, device=self.device)

    # ------------------------------------------------------------------ G0 / G1
    def preprocess(self, bgr: torch.Tensor):
        """bgr (B,H,W,3) uint8 -> lab, hsv (B,H,W,3) f32, gray, grad (B,H,W) f32."""
        b, h, w, _ = bgr.shape
        lab, hsv = self.empty(b, h, w, 3), self.empty(b, h, w, 3)
        gray, grad = self.empty(b, h, w), self.empty(b, h, w)
        self.ctx.call("ggc_preprocess", self._stream(), b, h, w, bgr.data_ptr(), lab.data_ptr(), hsv.data_ptr(),
                      gray.data_ptr(), grad.data_ptr())
        return lab, hsv, gray, grad

    def slic(self, image: torch.Tensor, n_segments: int, compactness: float = 10.0, sigma: float = 1.0,
             rescale_input: bool = True):
        b, h, w, _ = image.shape
        seg = self.empty(b, h, w, dtype=torch.int32)
        n = self.empty(b, dtype=torch.int32)
        self.ctx.call("ggc_slic", self._stream(), b, h, w, image.data_ptr(), int(n_segments), float(compactness),
                      float(sigma), int(bool(rescale_input)), seg.data_ptr(), n.data_ptr())
        return seg, n

    def slic_rgb(self, bgr: torch.Tensor, n_segments: int, compactness: float = 10.0, sigma: float = 1.0):
        """SuperpixelGraphConfig(use_lab=False): SLIC on rgb.astype(float), skimage's float64 path (reference graph_builder.py:177-179)."""
        b, h, w, _ = bgr.shape
        seg = self.empty(b, h, w, dtype=torch.int32)
        n = self.empty(b, dtype=torch.int32)
        self.ctx.call("ggc_slic_rgb", self._stream(), b, h, w, bgr.data_ptr(), int(n_segments), float(compactness), float(sigma),
                      seg.data_ptr(), n.data_ptr())
        return seg, n

    # ------------------------------------------------------------------ G2-G8
    def build_graphs(self, seg, n_nodes, lab, hsv, grad, connectivity: int = 4, n_nonlocal: int = 4) -> DeviceGraphs:
        b, h, w = seg.shape
        node_ptr = np.zeros(b + 1, np.int64)
        edge_ptr = np.zeros(b + 1, np.int64)
        self.ctx.call("ggc_graph_count", self._stream(), b, h, w, seg.data_ptr(), n_nodes.data_ptr(), lab.data_ptr(),
                      hsv.data_ptr(), grad.data_ptr(), int(connectivity), int(n_nonlocal), node_ptr.ctypes.data,
                      edge_ptr.ctypes.data)
        n, e = int(node_ptr[-1]), int(edge_ptr[-1])
        x, cen, area = self.empty(n, 19), self.empty(n, 2), self.empty(n)
        src = self.empty(max(e, 1), dtype=torch.int32)
        dst = self.empty(max(e, 1), dtype=torch.int32)
        attr = self.empty(max(e, 1), 5)
        self.ctx.call("ggc_graph_fill", self._stream(), x.data_ptr(), cen.data_ptr(), area.data_ptr(), src.data_ptr(),
                      dst.data_ptr(), attr.data_ptr(), 1)
        return DeviceGraphs(seg, n_nodes, node_ptr, edge_ptr, self.to_device(node_ptr.astype(np.int32)), x, cen, area,
                            src[:e], dst[:e], attr[:e])

    # ------------------------------------------------------------------ M0-M7
    def predict_probs(self, model, graphs: DeviceGraphs) -> torch.Tensor:
        from .data import Data
        d = Data(x=graphs.x, edge_attr=graphs.edge_attr)
        d.edge_index = torch.stack([graphs.edge_src, graphs.edge_dst]) if graphs.edge_src.numel() else \
            torch.zeros(2, 0, dtype=torch.int32, device=self.device)
        d.node_ptr32 = graphs.node_ptr
        return model.predict_probs_device(d, ctx=self.ctx)

    # ------------------------------------------------------------------ P0-P3, S0
    def refine_trimap(self, probs, node_ptr, seg, bgr, thr_fg=0.55, thr_bg=0.55, radius=8, eps=1e-3,
                      edge_aware=True) -> torch.Tensor:
        b, h, w = seg.shape
        tri = self.empty(b, h, w, dtype=torch.uint8)
        self.ctx.call("ggc_refine_trimap", self._stream(), b, h, w, probs.data_ptr(), node_ptr.data_ptr(),
                      seg.data_ptr(), bgr.data_ptr(), float(thr_fg), float(thr_bg), int(radius), float(eps),
                      int(bool(edge_aware)), tri.data_ptr())
        return tri

    def guided_filter(self, guide, src, radius=8, eps=1e-3) -> torch.Tensor:
        b, h, w = guide.shape
        out = self.empty(b, h, w)
        self.ctx.call("ggc_guided_filter", self._stream(), b, h, w, guide.data_ptr(), src.data_ptr(), int(radius),
                      float(eps), out.data_ptr())
        return out

    def seed_from_prior(self, trimap, prior, node_ptr, seg, seed_frac=0.1) -> torch.Tensor:
        b, h, w = seg.shape
        prior = prior.contiguous()
        self.ctx.call("ggc_seed_from_prior", self._stream(), b, h, w, prior.data_ptr(), node_ptr.data_ptr(),
                      seg.data_ptr(), float(seed_frac), trimap.data_ptr())
        return trimap

    # ------------------------------------------------------------------ C0-C6, K0, O0, R0
    def grabcut(self, image, mask, n_iter=5, mode=0, rects=None, seed=0, bgd=None, fgd=None):
        """In place on mask; returns (binary, mask, bgd_model, fgd_model)."""
        b, h, w, _ = image.shape
        if bgd is None:
            bgd = torch.zeros(b, 65, dtype=torch.float64, device=self.device)
        if fgd is None:
            fgd = torch.zeros(b, 65, dtype=torch.float64, device=self.device)
        binary = self.empty(b, h, w, dtype=torch.uint8)
        r = None if rects is None else np.ascontiguousarray(rects, dtype=np.int32).reshape(b, 4)
        self.ctx.call("ggc_grabcut", self._stream(), b, h, w, image.data_ptr(), mask.data_ptr(),
                      None if r is None else r.ctypes.data, bgd.data_ptr(), fgd.data_ptr(), int(n_iter), int(mode),
                      int(seed), binary.data_ptr())
        return binary, mask, bgd, fgd

    def convert_color8(self, bgr: torch.Tensor, space: str) -> torch.Tensor:
        """(…,3) uint8 BGR -> 8-bit HSV / Lab in the style of cv2.cvtColor (ggc_convert_color8)."""
        out = torch.empty_like(bgr)
        self.ctx.call("ggc_convert_color8", self._stream(), int(bgr.numel() // 3), bgr.data_ptr(), {"hsv": 0, "lab": 1}[space],
                      out.data_ptr())
        return out

    def clean_mask(self, mask, min_area_ratio=0.002, keep_largest=False, out=None) -> torch.Tensor:
        b, h, w = mask.shape
        if out is None:
            out = torch.empty_like(mask)
        self.ctx.call("ggc_clean_mask", self._stream(), b, h, w, mask.data_ptr(), float(min_area_ratio),
                      int(bool(keep_largest)), out.data_ptr())
        return out

    def compose(self, bgr, binary, alpha=0.45, tint_bgr=(100, 220, 0), out=None):
        b, h, w, _ = bgr.shape
        overlay, rgba = out if out is not None else (self.empty(b, h, w, 3, dtype=torch.uint8), self.empty(b, h, w, 4, dtype=torch.uint8))
        self.ctx.call("ggc_compose_outputs", self._stream(), b, h, w, bgr.data_ptr(), binary.data_ptr(), float(alpha),
                      int(tint_bgr[0]), int(tint_bgr[1]), int(tint_bgr[2]), overlay.data_ptr(), rgba.data_ptr())
        return overlay, rgba

    def iou(self, pred, gt):
        """-> (iou (B,) float64, counts (B,3) int64 = tp, fp, fn), on device."""
        b, h, w = pred.shape
        iou = self.empty(b, dtype=torch.float64)
        cnt = self.empty(b, 3, dtype=torch.int64)
        self.ctx.call("ggc_mask_iou", self._stream(), b, h, w, pred.data_ptr(), gt.data_ptr(), iou.data_ptr(),
                      cnt.data_ptr())
        return iou, cnt


def merge_graphs(parts: Sequence[DeviceGraphs], segments: torch.Tensor) -> DeviceGraphs:
    """The graphs of consecutive chunks of a batch as ONE packed batch (what build_graphs returns for the whole batch):
    node-indexed arrays are concatenated, edge endpoints shifted by the nodes that precede their chunk."""
    if len(parts) == 1:
        return parts[0]
    node_ptr, edge_ptr, src, dst = [np.zeros(1, np.int64)], [np.zeros(1, np.int64)], [], []
    n0 = e0 = 0
    for g in parts:
        node_ptr.append(g.node_ptr_host[1:] + n0)
        edge_ptr.append(g.edge_ptr_host[1:] + e0)
        src.append(g.edge_src + n0 if n0 else g.edge_src)
        dst.append(g.edge_dst + n0 if n0 else g.edge_dst)
        n0 += int(g.node_ptr_host[-1])
        e0 += int(g.edge_ptr_host[-1])
    node_ptr, edge_ptr = np.concatenate(node_ptr), np.concatenate(edge_ptr)
    dev = segments.device
    return DeviceGraphs(segments, torch.cat([g.n_nodes for g in parts]), node_ptr, edge_ptr,
                        torch.from_numpy(node_ptr.astype(np.int32)).to(dev, non_blocking=True),
                        torch.cat([g.x for g in parts]), torch.cat([g.centroids for g in parts]),
                        torch.cat([g.area_ratio for g in parts]), torch.cat(src), torch.cat(dst),
                        torch.cat([g.edge_attr for g in parts]))


_engines: dict[int, Engine] = {}


def get_engine(device="cuda") -> Engine:
    """Engine for a device spec ("cuda", "cuda:1", torch.device, int)."""
    if isinstance(device, int):
        idx = device
    else:
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError(f"device '{device}': gcn_grabcut runs on MI355X (device 'cuda') only; "
                               "there is no CPU fallback")
        idx = dev.index if dev.index is not None else (torch.cuda.current_device() if torch.cuda.is_available() else 0)
    eng = _engines.get(idx)
    if eng is None:
        eng = Engine(idx)
        _engines[idx] = eng
    return eng
