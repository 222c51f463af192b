"""
GCN-GrabCut end-to-end pipeline — host mirror of reference src/gcn_grabcut/pipeline.py.

  1. superpixel graph (+ automatic FG/BG prior)      ggc_preprocess, ggc_slic, ggc_graph_*
  2. ResGCNNet -> region probabilities               ggc_resgcn_forward
  3. guided-filter projection -> pixel trimap        ggc_refine_trimap (+ ggc_seed_from_prior)
  4. GrabCut refinement -> binary mask               ggc_grabcut
  5. clean-up and output composition                 ggc_clean_mask, ggc_compose_outputs

`segment(image)` keeps the reference signature and result type; `segment_batch`
(additive) runs the same stages over a batch of equally sized images with every
intermediate resident in HBM.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

from .grabcut import GrabCut, GrabCutConfig, Label
from .graph_builder import GraphBuilder, SuperpixelGraphConfig, graphs_to_host, _check_image
from .metrics import evaluate, evaluate_trimap, SegmentationMetrics, TrimapMetrics
from .model import CLASS_BG, CLASS_FG, project_to_pixels  # noqa: F401


def _write_png(path: str, array: np.ndarray) -> None:
    """cv2.imwrite stand-in (OpenCV is not a dependency): BGR(A) array -> PNG via Pillow."""
    from PIL import Image
    a = np.asarray(array)
    if a.ndim == 3 and a.shape[2] == 3:
        a = a[:, :, ::-1]
    elif a.ndim == 3 and a.shape[2] == 4:
        a = a[:, :, [2, 1, 0, 3]]
    Image.fromarray(np.ascontiguousarray(a)).save(path)


@dataclass
class SegmentationResult:
    """All outputs from one pipeline run (reference pipeline.py:32-68)."""
    image: np.ndarray          # original BGR
    binary_mask: np.ndarray    # (H, W) uint8 {0, 1}
    trimap: np.ndarray         # (H, W) uint8 {0,1,2,3}
    segments: np.ndarray       # (H, W) superpixel map
    overlay: np.ndarray        # BGR with coloured overlay
    rgba: np.ndarray           # BGRA transparent background
    timing: dict = field(default_factory=dict)

    def save(self, prefix: str = "result") -> None:
        _write_png(f"{prefix}_overlay.png", self.overlay)
        _write_png(f"{prefix}_rgba.png", self.rgba)
        _write_png(f"{prefix}_trimap_colour.png", _colour_trimap(self.trimap))
        _write_png(f"{prefix}_mask.png", self.binary_mask * 255)
        print(f"Saved outputs with prefix: {prefix}")

    def evaluate_against(self, gt_mask: np.ndarray) -> "tuple[SegmentationMetrics, TrimapMetrics]":
        """Segmentation and trimap metrics against a ground-truth mask (reference pipeline.py:62-68)."""
        return evaluate(self.binary_mask, gt_mask), evaluate_trimap(self.trimap, gt_mask)


def guided_filter(guide: np.ndarray, src: np.ndarray, radius: int = 8, eps: float = 1e-3, device="cuda") -> np.ndarray:
    """Edge-preserving filter of `src` under `guide` (He et al.) — reference pipeline.py:71-100."""
    from ._engine import get_engine
    eng = get_engine(device)
    g = eng.to_device(np.ascontiguousarray(guide, dtype=np.float32)[None])
    s = eng.to_device(np.ascontiguousarray(src, dtype=np.float32)[None])
    return eng.guided_filter(g, s, radius, eps)[0].cpu().numpy()


def refine_trimap(probs: np.ndarray, segments: np.ndarray, image: np.ndarray, threshold_fg: float = 0.55,
                  threshold_bg: float = 0.55, radius: int = 8, eps: float = 1e-3, device="cuda") -> np.ndarray:
    """Region probabilities -> pixel trimap whose boundaries follow image edges
    (reference pipeline.py:103-146)."""
    from ._engine import get_engine
    eng = get_engine(device)
    p = eng.to_device(np.ascontiguousarray(probs, dtype=np.float32))
    node_ptr = eng.to_device(np.array([0, probs.shape[0]], np.int32))
    seg = eng.to_device(np.ascontiguousarray(segments, dtype=np.int32)[None])
    bgr = eng.to_device(_check_image(image)[None])
    return eng.refine_trimap(p, node_ptr, seg, bgr, threshold_fg, threshold_bg, radius, eps, True)[0].cpu().numpy()


def _seed_from_prior(trimap: np.ndarray, graph, seed_frac: float = 0.1, device="cuda") -> np.ndarray:
    """Guarantee a foreground and a background seed (reference pipeline.py:149-186)."""
    from ._engine import get_engine
    prior = graph.prior_features
    if prior is None or prior.size == 0:
        return trimap
    eng = get_engine(device)
    t = eng.to_device(np.ascontiguousarray(trimap, dtype=np.uint8)[None]).clone()
    p = eng.to_device(np.ascontiguousarray(prior, dtype=np.float32))
    node_ptr = eng.to_device(np.array([0, graph.n_nodes], np.int32))
    seg = eng.to_device(np.ascontiguousarray(graph.segments, dtype=np.int32)[None])
    return eng.seed_from_prior(t, p, node_ptr, seg, seed_frac)[0].cpu().numpy()


def clean_mask(mask: np.ndarray, min_area_ratio: float = 0.002, keep_largest: bool = False, device="cuda") -> np.ndarray:
    """Remove spurious connected components (reference pipeline.py:189-227)."""
    from ._engine import get_engine
    eng = get_engine(device)
    m = eng.to_device(np.ascontiguousarray(mask, dtype=np.uint8)[None])
    return eng.clean_mask(m, min_area_ratio, keep_largest)[0].cpu().numpy()


def eroded_box(H: int, W: int, bbox: tuple[int, int, int, int], ksize: int = 30) -> tuple[int, int, int, int]:
    """(y0, y1, x0, x1) of `cv2.erode(box_mask, np.ones((ksize, ksize)))` for the filled box `bbox` = (x, y, w, h) in an H x W
    image (reference pipeline.py:366-370), in closed form: the kernel's anchor is its centre (ksize // 2), so a pixel survives
    when the ksize pixels from -ksize // 2 to ksize - 1 - ksize // 2 around it are all inside the box; cv2.erode's default
    border counts pixels OUTSIDE the image as set, so a side of the box that lies on the image edge is not eroded."""
    x, y, w, h = bbox
    a, b = ksize // 2, ksize - 1 - ksize // 2
    # the reference fills `inner[y:y+h, x:x+w] = 1` (pipeline.py:368): numpy slice semantics, so a negative start counts from
    # the far edge (and usually selects nothing) and a stop past the frame is cut — the same rows and columns the
    # FG_PROBABLE write of segment_bbox gets
    by0, by1, _ = slice(y, y + h).indices(H)
    bx0, bx1, _ = slice(x, x + w).indices(W)
    if by1 <= by0 or bx1 <= bx0:
        return 0, 0, 0, 0
    y0 = by0 if by0 == 0 else by0 + a
    y1 = by1 if by1 == H else by1 - b
    x0 = bx0 if bx0 == 0 else bx0 + a
    x1 = bx1 if bx1 == W else bx1 - b
    return y0, y1, x0, x1


def _colour_trimap(trimap: np.ndarray) -> np.ndarray:
    """reference pipeline.py:230-236"""
    vis = np.zeros((*trimap.shape, 3), dtype=np.uint8)
    vis[trimap == Label.BG_DEFINITE] = [0, 0, 0]
    vis[trimap == Label.FG_DEFINITE] = [255, 255, 255]
    vis[trimap == Label.BG_PROBABLE] = [60, 20, 20]
    vis[trimap == Label.FG_PROBABLE] = [0, 200, 200]
    return vis


class GCNGrabCutPipeline:
    """
    Full GCN-GrabCut segmentation pipeline on the MI355X (reference pipeline.py:239-380).

    model     : trimap predictor (ResGCNNet)
    sp_config : SuperpixelGraphConfig (default 300 segments)
    gc_config : GrabCutConfig (default 5 iterations)
    device    : "cuda" / "cuda:<i>"
    """

    def __init__(self, model, sp_config: Optional[SuperpixelGraphConfig] = None,
                 gc_config: Optional[GrabCutConfig] = None, device: str = "cuda", grabcut_lanes: int = 4, engine=None,
                 chunks: int = 1, chunk_ratio: float = 0.8):
        from ._engine import get_engine
        self._eng = engine if engine is not None else get_engine(device)
        self.grabcut_lanes = int(grabcut_lanes)   # additive: concurrent sub-batches of the GrabCut stage (batched calls only)
        self.chunks = int(chunks)                 # additive: chunks of the software pipeline of segment_batch_device (1 = off, the default: measured no faster)
        self.chunk_ratio = float(chunk_ratio)
        self.model = model.to(self._eng.device)
        self.device = device
        self.sp_config = sp_config or SuperpixelGraphConfig()
        self.gc_config = gc_config or GrabCutConfig()
        self._replicas = None

    def replica(self, grabcut_lanes: int = 1) -> "GCNGrabCutPipeline":
        """Additive: a pipeline over the same model with a private library context (own scratch arena), so that it can
        run CONCURRENTLY with this one from another host thread on another HIP stream — batch k+1's SLIC / graph / GCN
        stages then run under batch k's GrabCut, whose max-flow leaves most of the GPU idle."""
        from ._engine import Engine
        return GCNGrabCutPipeline(self.model, self.sp_config, self.gc_config, self.device, grabcut_lanes,
                                  engine=Engine(self._eng.index, private_context=True))

    def segment_batches_overlapped(self, batches: Sequence, n_pipelines: int = 3, **kwargs) -> list[dict]:
        """Additive: segment_batch_device() over several (B,H,W,3) uint8 device tensors with up to n_pipelines of them
        in flight at once (this pipeline + replicas, one host thread and HIP stream each, one GrabCut lane each).
        Results are those of one-at-a-time calls, in input order; on one MI355X three pipelines move ~25 % more
        images per second than one pipeline with four GrabCut lanes."""
        import threading
        import torch
        batches = list(batches)
        n = max(1, min(int(n_pipelines), len(batches)))
        if n == 1:
            return [self.segment_batch_device(b, **kwargs) for b in batches]
        if getattr(self, "_replicas", None) is None or len(self._replicas) < n - 1:
            self._replicas = [self.replica(grabcut_lanes=1) for _ in range(n - 1)]
            self._streams = [torch.cuda.Stream(self._eng.device) for _ in range(n)]
        pipes = [self] + self._replicas[:n - 1]
        caller = torch.cuda.current_stream(self._eng.device)
        out: list = [None] * len(batches)
        errors: list = []

        def worker(i):
            try:
                torch.cuda.set_device(self._eng.device)
                self._streams[i].wait_stream(caller)               # the batches were produced on the caller's stream
                with torch.cuda.stream(self._streams[i]):
                    for k in range(i, len(batches), n):
                        out[k] = pipes[i].segment_batch_device(batches[k], grabcut_lanes=1, **kwargs)
                self._streams[i].synchronize()
            except Exception as exc:                               # re-raised in the caller's thread
                errors.append(exc)

        threads = [threading.Thread(target=worker, args=(i,), name=f"ggc-pipe{i}") for i in range(n)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        for o in out:                                              # the results were allocated on the side streams
            for v in o.values():
                for t_ in (vars(v).values() if hasattr(v, "__dict__") and not torch.is_tensor(v) else (v,)):
                    if torch.is_tensor(t_) and t_.is_cuda:
                        t_.record_stream(caller)
        return out

    # ------------------------------------------------------------ batched, device resident
    def _front(self, eng, bgr, threshold_fg, threshold_bg, edge_aware, filter_radius, tick=None, timing=None):
        """Stages 1-3 on the current stream: colour prep, SLIC, graph, network, trimap, seeding.  -> (seg, graphs, probs, trimap)"""
        cfg = self.sp_config
        t = tick() if tick else 0.0
        lab, hsv, gray, grad = eng.preprocess(bgr)
        seg, n_nodes = eng.slic(lab, cfg.n_segments, cfg.compactness, cfg.sigma) if cfg.use_lab else \
            eng.slic_rgb(bgr, cfg.n_segments, cfg.compactness, cfg.sigma)          # reference graph_builder.py:177-179
        graphs = eng.build_graphs(seg, n_nodes, lab, hsv, grad, cfg.connectivity, cfg.n_nonlocal)
        if timing is not None:
            timing["graph_build"] = tick() - t
            timing["data_prep"] = 0.0          # the graph is already in HBM: nothing to copy
        t = tick() if tick else 0.0
        if self.model.training:
            self.model.eval()
        probs = eng.predict_probs(self.model, graphs)
        trimap = eng.refine_trimap(probs, graphs.node_ptr, seg, bgr, threshold_fg, threshold_bg, filter_radius,
                                   1e-3, edge_aware)
        if timing is not None:
            timing["gcn_inference"] = tick() - t
        trimap = eng.seed_from_prior(trimap, graphs.x[:, 16:19], graphs.node_ptr, seg, 0.1)
        return seg, graphs, probs, trimap

    @staticmethod
    def chunk_plan(b: int, n_chunks: int, ratio: float = 0.8) -> list[tuple[int, int]]:
        """Contiguous chunks of a batch for the software pipeline, each `ratio` times the size of the one before it: the
        GrabCut of a chunk starts when its trimaps exist, so later chunks start later and get fewer images to end together."""
        n_chunks = max(1, min(int(n_chunks), b))
        w = [ratio ** k for k in range(n_chunks)]
        cuts, acc = [0], 0.0
        for k in range(n_chunks - 1):
            acc += w[k]
            cuts.append(min(b - (n_chunks - 1 - k), max(cuts[-1] + 1, int(round(b * acc / sum(w))))))
        cuts.append(b)
        return [(cuts[k], cuts[k + 1]) for k in range(n_chunks)]

    def segment_batch_device(self, bgr, threshold_fg: float = 0.55, threshold_bg: float = 0.55,
                             refine_iters: int = 0, min_area_ratio: float = 0.002, keep_largest: bool = False,
                             edge_aware: bool = True, filter_radius: int = 8, compose: bool = True,
                             timing: Optional[dict] = None, grabcut_lanes: Optional[int] = None,
                             chunks: Optional[int] = None) -> dict:
        """bgr: (B,H,W,3) uint8 tensor on the pipeline's device.  Returns device tensors.

        Large batches run as a software pipeline (additive, same results): the batch is cut into `chunks` contiguous
        chunks; the front stages of chunk k+1 run on the caller's stream while the GrabCut / clean-up of chunk k runs on a
        lane of its own (private context, stream and host thread).  Images are independent and image b keeps seed + b, so
        every output equals the one-chunk run bit for bit."""
        import torch
        eng, cfg = self._eng, self.sp_config
        cs = self.gc_config.color_space.lower()
        if cs not in ("rgb", "hsv", "lab"):
            raise ValueError(f"unknown color_space '{cs}': rgb | hsv | lab")
        b = bgr.size(0)
        want = self.grabcut_lanes if grabcut_lanes is None else int(grabcut_lanes)   # (an argument, so that concurrent callers do not mutate the pipeline)
        n_chunks = self.chunks if chunks is None else int(chunks)
        if n_chunks <= 0:                          # 0: one chunk per GrabCut lane once every chunk gets a lane's worth of images
            n_chunks = max(want, 1) if b >= 16 * max(want, 1) else 1
        if n_chunks > 1 and b >= 2 * n_chunks:
            return self._segment_pipelined(bgr, self.chunk_plan(b, n_chunks, self.chunk_ratio), cs, threshold_fg, threshold_bg,
                                           refine_iters, min_area_ratio, keep_largest, edge_aware, filter_radius, compose, timing)

        def tick():
            if timing is not None:
                torch.cuda.synchronize(eng.device)
            return time.perf_counter()

        seg, graphs, probs, trimap = self._front(eng, bgr, threshold_fg, threshold_bg, edge_aware, filter_radius, tick, timing)

        t = tick()
        mask = trimap.clone()
        lanes = want if b >= 8 * max(want, 1) else 1
        gc_img = bgr if cs == "rgb" else eng.convert_color8(bgr, cs)      # reference grabcut.py:73-79
        # clean-up and composition are per image: without stage timing each GrabCut lane runs them for its own sub-batch as
        # soon as it is cut (on its stream, under the other lanes' tails); with timing they stay a stage of their own
        cleaned = eng.empty(*bgr.shape[:3], dtype=torch.uint8)
        overlay = eng.empty(*bgr.shape[:3], 3, dtype=torch.uint8) if compose else None
        rgba = eng.empty(*bgr.shape[:3], 4, dtype=torch.uint8) if compose else None

        def post(leng, lo, hi, binary_part):
            leng.clean_mask(binary_part, min_area_ratio, keep_largest, out=cleaned[lo:hi])
            if compose:
                leng.compose(bgr[lo:hi], cleaned[lo:hi], out=(overlay[lo:hi], rgba[lo:hi]))

        fused_post = timing is None
        binary, mask, bgd, fgd = eng.grabcut_lanes(gc_img, mask, self.gc_config.n_iter, 0, self.gc_config.seed, lanes,
                                                   post=post if (fused_post and refine_iters <= 0) else None)
        if refine_iters > 0:
            binary, mask, bgd, fgd = eng.grabcut_lanes(gc_img, mask, refine_iters, 2, self.gc_config.seed, lanes, bgd, fgd,
                                                       post=post if fused_post else None)
        if timing is not None:
            timing["grabcut"] = tick() - t

        t = tick()
        if not fused_post:
            post(eng, 0, bgr.size(0), binary)
        out = {"binary_mask": cleaned, "trimap": trimap, "segments": seg, "graphs": graphs, "probs": probs,
               "gc_mask": mask}
        if compose:
            out["overlay"], out["rgba"] = overlay, rgba
        if timing is not None:
            timing["postprocess"] = tick() - t
        return out

    def _segment_pipelined(self, bgr, plan, cs, threshold_fg, threshold_bg, refine_iters, min_area_ratio, keep_largest,
                           edge_aware, filter_radius, compose, timing) -> dict:
        """The software pipeline behind segment_batch_device: chunk k's GrabCut lane starts as soon as chunk k's trimaps are
        on the device; the caller's stream goes on with chunk k+1's SLIC / graph / network / trimap."""
        import torch
        from concurrent.futures import ThreadPoolExecutor
        from ._engine import merge_graphs
        eng = self._eng
        dev = eng.device
        b, h, w, _ = bgr.shape
        n = len(plan)
        caller = torch.cuda.current_stream(dev)
        lanes = eng.lanes(n)
        if getattr(self, "_chunk_pool", None) is None or self._chunk_pool._max_workers != n:
            if getattr(self, "_chunk_pool", None) is not None:
                self._chunk_pool.shutdown(wait=True)
            self._chunk_pool = ThreadPoolExecutor(max_workers=n, thread_name_prefix="ggc-chunk")
        seg = eng.empty(b, h, w, dtype=torch.int32)
        trimap = eng.empty(b, h, w, dtype=torch.uint8)
        mask = eng.empty(b, h, w, dtype=torch.uint8)
        cleaned = eng.empty(b, h, w, dtype=torch.uint8)
        overlay = eng.empty(b, h, w, 3, dtype=torch.uint8) if compose else None
        rgba = eng.empty(b, h, w, 4, dtype=torch.uint8) if compose else None
        n_iter, seed = self.gc_config.n_iter, self.gc_config.seed
        t_host = time.perf_counter()
        stamps = []                                   # per chunk: events around its front stages / its lane's work

        def lane_work(k, lo, hi, ready, ev):
            leng, stream = lanes[k]
            torch.cuda.set_device(dev)
            with torch.cuda.stream(stream):
                stream.wait_event(ready)              # the chunk's trimap, written on the caller's stream
                if ev is not None:
                    ev[0].record(stream)
                img = bgr[lo:hi]
                gc_img = img if cs == "rgb" else leng.convert_color8(img, cs)
                m = mask[lo:hi]
                binary, _, bgd, fgd = leng.grabcut(gc_img, m, n_iter, 0, None, seed + lo)
                if refine_iters > 0:
                    binary, _, bgd, fgd = leng.grabcut(gc_img, m, refine_iters, 2, None, seed + lo, bgd, fgd)
                if ev is not None:
                    ev[1].record(stream)
                leng.clean_mask(binary, min_area_ratio, keep_largest, out=cleaned[lo:hi])
                if compose:
                    leng.compose(img, cleaned[lo:hi], out=(overlay[lo:hi], rgba[lo:hi]))
                if ev is not None:
                    ev[2].record(stream)
                done = torch.cuda.Event()
                done.record(stream)
            return done

        futures, parts = [], []
        for k, (lo, hi) in enumerate(plan):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if timing is not None else None
            if ev is not None:
                ev[3].record(caller)
            s_k, g_k, p_k, t_k = self._front(eng, bgr[lo:hi], threshold_fg, threshold_bg, edge_aware, filter_radius)
            seg[lo:hi].copy_(s_k)
            trimap[lo:hi].copy_(t_k)
            mask[lo:hi].copy_(t_k)
            if ev is not None:
                ev[4].record(caller)
            ready = torch.cuda.Event()
            ready.record(caller)
            parts.append((g_k, p_k))
            stamps.append(ev)
            futures.append(self._chunk_pool.submit(lane_work, k, lo, hi, ready, ev))
        graphs = merge_graphs([g for g, _ in parts], seg)            # under the lanes' GrabCut
        probs = torch.cat([p for _, p in parts])
        for f in futures:
            caller.wait_event(f.result())             # whatever the caller enqueues next sees the lanes' outputs
        out = {"binary_mask": cleaned, "trimap": trimap, "segments": seg, "graphs": graphs, "probs": probs, "gc_mask": mask}
        if compose:
            out["overlay"], out["rgba"] = overlay, rgba
        if timing is not None:                        # stage times from stream events (the stages overlap: they add up to more than the wall time)
            torch.cuda.synchronize(dev)
            front = sum(e[3].elapsed_time(e[4]) for e in stamps) / 1e3
            timing["graph_build"], timing["data_prep"], timing["gcn_inference"] = front, 0.0, 0.0
            timing["grabcut"] = sum(e[0].elapsed_time(e[1]) for e in stamps) / 1e3
            timing["postprocess"] = sum(e[1].elapsed_time(e[2]) for e in stamps) / 1e3
            timing["wall"] = time.perf_counter() - t_host
        return out

    def segment_batch(self, images: Sequence[np.ndarray], **kwargs) -> list[SegmentationResult]:
        """Segment equally sized BGR images as one batch (additive API)."""
        imgs = [_check_image(im) for im in images]
        if not imgs:
            return []
        if any(im.shape != imgs[0].shape for im in imgs):
            raise ValueError("segment_batch needs images of one size; group them by shape")
        timing: dict[str, float] = {}
        bgr = self._eng.to_device(np.stack(imgs))
        out = self.segment_batch_device(bgr, timing=timing, **kwargs)
        host = {k: out[k].cpu().numpy() for k in ("binary_mask", "trimap", "segments", "overlay", "rgba")}
        per_image = {k: v / len(imgs) for k, v in timing.items()}
        return [SegmentationResult(image=imgs[i], binary_mask=host["binary_mask"][i], trimap=host["trimap"][i],
                                   segments=host["segments"][i], overlay=host["overlay"][i], rgba=host["rgba"][i],
                                   timing=dict(per_image)) for i in range(len(imgs))]

    # ------------------------------------------------------------ reference API
    def segment(self, image: np.ndarray, threshold_fg: float = 0.55, threshold_bg: float = 0.55,
                refine_iters: int = 0, min_area_ratio: float = 0.002, keep_largest: bool = False,
                edge_aware: bool = True, filter_radius: int = 8) -> SegmentationResult:
        """Full pipeline on one BGR image (reference pipeline.py:265-352)."""
        image = _check_image(image)
        timing: dict[str, float] = {}
        out = self.segment_batch_device(self._eng.to_device(image[None]), threshold_fg, threshold_bg, refine_iters,
                                        min_area_ratio, keep_largest, edge_aware, filter_radius, timing=timing)
        return SegmentationResult(
            image=image, binary_mask=out["binary_mask"][0].cpu().numpy(), trimap=out["trimap"][0].cpu().numpy(),
            segments=out["segments"][0].cpu().numpy(), overlay=out["overlay"][0].cpu().numpy(),
            rgba=out["rgba"][0].cpu().numpy(), timing=timing)

    def segment_bbox(self, image: np.ndarray, bbox: tuple[int, int, int, int]) -> SegmentationResult:
        """Classical GrabCut with a bounding box (reference pipeline.py:354-380)."""
        image = _check_image(image)
        gc = GrabCut(image, self.gc_config, device=self.device)
        binary_mask = gc.run_with_bbox(bbox)
        x, y, w, h = bbox
        H, W = image.shape[:2]
        trimap = np.full((H, W), Label.BG_PROBABLE, dtype=np.uint8)
        trimap[y:y + h, x:x + w] = Label.FG_PROBABLE
        y0, y1, x0, x1 = eroded_box(H, W, bbox)
        if y1 > y0 and x1 > x0:
            trimap[y0:y1, x0:x1] = Label.FG_DEFINITE
        return SegmentationResult(image=image, binary_mask=binary_mask, trimap=trimap,
                                  segments=np.zeros((H, W), dtype=np.int32), overlay=gc.overlay_mask(),
                                  rgba=gc.crop_foreground())
