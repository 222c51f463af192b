"""
Graph-cache writer on the MI355X graph builder.

Host mirror of the part of the reference's `dataset.py` that `tools/prepare_graphs.py`
drives: `list_image_mask_pairs` (dataset.py:263-313), `materialise` (:316-361),
`derive_trimap_labels` (:175-206), `prepare_sample` (:213-260) and `prepare_dataset`
(:444-540).  Where the reference fans single images out to CPU worker processes, this
version decodes on a few host threads and pushes whole batches of equally sized images
through the device stages G0-G8 (colour prep, SLIC, graph) in one go; the per-region
ground-truth coverage is an integer reduction on the device (`ggc_region_label_stats`).

Differences that are deliberate and documented:

* Cache entries are a plain dictionary of tensors (`format: "ggc-graph-v1"`), readable with
  `torch.load(..., weights_only=True)`.  The reference pickles a torch_geometric `Data`
  object (dataset.py:437), which cannot be produced or read without PyG; file NAMES follow the
  reference's `_cache_key` recipe (:364-378) so a cache directory is laid out the same way.
* Images are decoded with Pillow (OpenCV is not installed) and resized with a plain
  half-pixel-centre bilinear / nearest filter.  `cv2.imread` / `cv2.resize` parity is
  UNPINNED: neither library nor golden vectors are available here.
* Seeded augmentation (`aug_seed`, reference :107-168) draws its decisions from Python's `random` in the reference's
  order, so a seed selects the same flips / angles / jitters / crops; the warps and the 8-bit HSV round trip are plain
  numpy here instead of OpenCV (pixel parity UNPINNED for the same reason).

Training itself (Dataset classes, loaders, samplers) stays out of scope.
"""
from __future__ import annotations

import hashlib
import logging
import os
import time
import random
import zlib
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Optional

import numpy as np
import torch

from .data import Data
from .graph_builder import SuperpixelGraphConfig
from .model import CLASS_BG, CLASS_FG, CLASS_UNK

logger = logging.getLogger(__name__)

CACHE_FORMAT = "ggc-graph-v1"
_IMAGE_EXTS = {".jpg", ".jpeg", ".png", ".bmp", ".tif", ".tiff"}


# ----------------------------------------------------------------------- descriptors
def list_image_mask_pairs(images_dir, masks_dir, max_size: int = 512, augment_copies: int = 0,
                          seed: int = 0) -> list[dict]:
    """Image/mask pairs as descriptors (nothing is decoded) — reference dataset.py:263-313.
    Keys: image_path, mask_path, max_size, name, aug_seed."""
    images_dir, masks_dir = Path(images_dir), Path(masks_dir)
    out, missing = [], 0
    for img_path in sorted(f for f in images_dir.iterdir() if f.suffix.lower() in _IMAGE_EXTS):
        mask_path = None
        for ext in (".png", ".jpg", ".bmp", ".tif"):
            cand = masks_dir / (img_path.stem + ext)
            if cand.exists():
                mask_path = cand
                break
        if mask_path is None:
            missing += 1
            continue
        base = dict(image_path=str(img_path), mask_path=str(mask_path), max_size=max_size)
        out.append({**base, "name": img_path.stem, "aug_seed": None})
        for k in range(augment_copies):
            stem_id = zlib.crc32(img_path.stem.encode()) % 100003       # stable across interpreters
            out.append({**base, "name": f"{img_path.stem}_aug{k}", "aug_seed": seed + 1000003 * k + stem_id})
    print(f"[Dataset] {len(out)} descriptors from {images_dir.name} ({missing} without a mask)")
    return out


def _resize_bilinear_u8(img: np.ndarray, nh: int, nw: int) -> np.ndarray:
    """Half-pixel-centre bilinear resize of a uint8 (H,W,C) image, edge-clamped (float64 weights, round half up)."""
    h, w = img.shape[:2]
    ys = (np.arange(nh, dtype=np.float64) + 0.5) * (h / nh) - 0.5
    xs = (np.arange(nw, dtype=np.float64) + 0.5) * (w / nw) - 0.5
    y0 = np.floor(ys).astype(np.int64); x0 = np.floor(xs).astype(np.int64)
    fy = (ys - y0)[:, None, None]; fx = (xs - x0)[None, :, None]
    y0c, y1c = np.clip(y0, 0, h - 1), np.clip(y0 + 1, 0, h - 1)
    x0c, x1c = np.clip(x0, 0, w - 1), np.clip(x0 + 1, 0, w - 1)
    f = img.astype(np.float64)
    top = f[y0c][:, x0c] * (1 - fx) + f[y0c][:, x1c] * fx
    bot = f[y1c][:, x0c] * (1 - fx) + f[y1c][:, x1c] * fx
    return np.clip(np.floor(top * (1 - fy) + bot * fy + 0.5), 0, 255).astype(np.uint8)


def _resize_nearest(img: np.ndarray, nh: int, nw: int) -> np.ndarray:
    h, w = img.shape[:2]
    ys = np.minimum((np.arange(nh) * (h / nh)).astype(np.int64), h - 1)
    xs = np.minimum((np.arange(nw) * (w / nw)).astype(np.int64), w - 1)
    return img[ys][:, xs]


def _resize_pair(image: np.ndarray, mask: np.ndarray, max_size: int):
    """Shrink so that the longer side is max_size (never enlarges) — reference dataset.py:772-780."""
    h, w = image.shape[:2]
    scale = max_size / max(h, w)
    if scale < 1.0:
        nw, nh = int(w * scale), int(h * scale)
        image = _resize_bilinear_u8(image, nh, nw)
        mask = _resize_nearest(mask, nh, nw)
    return image, mask


# ----------------------------------------------------------------------- augmentation
# reference dataset.py:107-168: flip, rotation in [-15, 15] degrees (bilinear image / nearest mask, reflected border),
# brightness / contrast / saturation jitter, crop + resize back.  The random DECISIONS are Python's `random` module drawn in
# the reference's order, so a descriptor's aug_seed selects the same transforms here as there; the pixel arithmetic of the
# warps and of the 8-bit HSV round trip is OpenCV's in the reference and plain numpy here (parity with OpenCV unpinned:
# the library is absent; the formulas follow its documented conventions: pixel-centre bilinear taps, H in [0, 180)).
def _reflect(i: np.ndarray, n: int) -> np.ndarray:
    """cv2.BORDER_REFLECT: ... c b a | a b c ... | c b a ..."""
    if n == 1:
        return np.zeros_like(i)
    p = np.mod(i, 2 * n)
    return np.where(p < n, p, 2 * n - 1 - p)


def _warp_affine(img: np.ndarray, m: np.ndarray, linear: bool) -> np.ndarray:
    """dst(x, y) = src(M^-1 (x, y)) like cv2.warpAffine(img, M, (W, H)) with BORDER_REFLECT."""
    h, w = img.shape[:2]
    a = np.vstack([m, [0.0, 0.0, 1.0]])
    inv = np.linalg.inv(a)
    xx, yy = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    sx = inv[0, 0] * xx + inv[0, 1] * yy + inv[0, 2]
    sy = inv[1, 0] * xx + inv[1, 1] * yy + inv[1, 2]
    if not linear:
        xi, yi = _reflect(np.rint(sx).astype(np.int64), w), _reflect(np.rint(sy).astype(np.int64), h)
        return img[yi, xi]
    x0, y0 = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
    fx, fy = (sx - x0), (sy - y0)
    if img.ndim == 3:
        fx, fy = fx[..., None], fy[..., None]
    x0r, x1r, y0r, y1r = _reflect(x0, w), _reflect(x0 + 1, w), _reflect(y0, h), _reflect(y0 + 1, h)
    f = img.astype(np.float64)
    top = f[y0r, x0r] * (1 - fx) + f[y0r, x1r] * fx
    bot = f[y1r, x0r] * (1 - fx) + f[y1r, x1r] * fx
    return np.clip(np.floor(top * (1 - fy) + bot * fy + 0.5), 0, 255).astype(img.dtype)


def _rotation_matrix(cx: float, cy: float, angle_deg: float) -> np.ndarray:
    """cv2.getRotationMatrix2D(centre, angle, 1.0)"""
    al, be = np.cos(np.deg2rad(angle_deg)), np.sin(np.deg2rad(angle_deg))
    return np.array([[al, be, (1 - al) * cx - be * cy], [-be, al, be * cx + (1 - al) * cy]], np.float64)


def _bgr_to_hsv8(bgr: np.ndarray) -> np.ndarray:
    """8-bit HSV in OpenCV's convention: H in [0, 180), S and V in [0, 255]"""
    f = bgr.astype(np.float64)
    b, g, r = f[..., 0], f[..., 1], f[..., 2]
    v = np.maximum(np.maximum(r, g), b)
    d = v - np.minimum(np.minimum(r, g), b)
    s = np.where(v > 0, d / np.maximum(v, 1e-12) * 255.0, 0.0)
    dd = np.maximum(d, 1e-12)
    hdeg = np.where(v == r, 60.0 * (g - b) / dd, np.where(v == g, 120.0 + 60.0 * (b - r) / dd, 240.0 + 60.0 * (r - g) / dd))
    hdeg = np.where(d > 0, np.mod(hdeg, 360.0), 0.0)
    out = np.stack([np.rint(hdeg / 2.0) % 180, np.rint(s), v], -1)
    return np.clip(out, 0, 255).astype(np.uint8)


def _hsv8_to_bgr(hsv: np.ndarray) -> np.ndarray:
    f = hsv.astype(np.float64)
    hh, s, v = f[..., 0] * 2.0 / 60.0, f[..., 1] / 255.0, f[..., 2]
    i = np.floor(hh).astype(np.int64) % 6
    ff = hh - np.floor(hh)
    p, q, t = v * (1 - s), v * (1 - s * ff), v * (1 - s * (1 - ff))
    r = np.choose(i, [v, q, p, p, t, v]); g = np.choose(i, [t, v, v, q, p, p]); b = np.choose(i, [p, p, t, v, v, q])
    return np.clip(np.rint(np.stack([b, g, r], -1)), 0, 255).astype(np.uint8)


def _color_jitter(image: np.ndarray, rng=random) -> np.ndarray:
    """reference dataset.py:154-168 (three draws: brightness, contrast, saturation)"""
    img = image.astype(np.float32)
    img = np.clip(img + np.float32(rng.uniform(-40, 40)), 0, 255)
    img = np.clip(np.float32(128) + np.float32(rng.uniform(0.7, 1.3)) * (img - np.float32(128)), 0, 255)
    hsv = _bgr_to_hsv8(img.astype(np.uint8)).astype(np.float32)
    hsv[:, :, 1] = np.clip(hsv[:, :, 1] * np.float32(rng.uniform(0.7, 1.3)), 0, 255)
    return _hsv8_to_bgr(hsv.astype(np.uint8))


def augment_sample(image: np.ndarray, mask: np.ndarray, prob_flip: float = 0.5, prob_rotate: float = 0.3,
                   prob_color: float = 0.5, prob_crop: float = 0.3, rng=random):
    """Stochastic augmentation of an image / mask pair — reference dataset.py:107-151; draws in its order from `rng`
    (the `random` module like the reference, or a private random.Random — same Mersenne stream, no global state)."""
    h, w = image.shape[:2]
    if rng.random() < prob_flip:
        image, mask = image[:, ::-1], mask[:, ::-1]
    if rng.random() < prob_rotate:
        m = _rotation_matrix(w / 2, h / 2, rng.uniform(-15, 15))
        image = _warp_affine(image, m, linear=True)
        mask = _warp_affine(mask.astype(np.uint8), m, linear=False)
    if rng.random() < prob_color:
        image = _color_jitter(image, rng)
    if rng.random() < prob_crop:
        scale = rng.uniform(0.75, 1.0)
        ch, cw = int(h * scale), int(w * scale)
        y0 = rng.randint(0, h - ch)
        x0 = rng.randint(0, w - cw)
        image = _resize_bilinear_u8(image[y0:y0 + ch, x0:x0 + cw], h, w)
        mask = _resize_nearest(mask[y0:y0 + ch, x0:x0 + cw], h, w)
    return np.ascontiguousarray(image), np.ascontiguousarray(mask)


def materialise(sample: dict) -> Optional[dict]:
    """Descriptor -> {"image": BGR uint8, "gt_mask": uint8 {0,1}, "name"} — reference dataset.py:316-361.
    In-memory samples pass through; unreadable or degenerate pairs give None."""
    if "image" in sample and "gt_mask" in sample:
        return sample
    from PIL import Image
    try:
        with Image.open(sample["image_path"]) as im:
            image = np.ascontiguousarray(np.asarray(im.convert("RGB"))[:, :, ::-1])      # BGR like cv2.imread
        with Image.open(sample["mask_path"]) as im:
            mask = np.asarray(im.convert("L"))
    except (OSError, ValueError) as e:
        logger.warning("unreadable pair: %s (%s)", sample.get("image_path"), e)
        return None
    if image.shape[:2] != mask.shape[:2]:
        logger.warning("image and mask differ in size: %s", sample.get("image_path"))
        return None
    image, mask = _resize_pair(image, mask, sample.get("max_size", 512))
    gt_mask = (mask > 127).astype(np.uint8)
    if sample.get("aug_seed") is not None:                 # seeded: the copy is the same every run (dataset.py:343-353)
        # The reference seeds the process-global `random` inside a worker PROCESS; prepare_dataset decodes on worker THREADS,
        # which would interleave their draws on one global generator.  A private random.Random(seed) is the same Mersenne
        # stream as random.seed(seed), so a seed still gives the reference's parameter sequence — on any thread.
        image, gt_mask = augment_sample(image, gt_mask, prob_flip=0.5, prob_rotate=0.4, prob_color=0.6, prob_crop=0.4,
                                        rng=random.Random(sample["aug_seed"]))
    if gt_mask.sum() < 200 or (1 - gt_mask).sum() < 200:
        return None
    return {"image": np.ascontiguousarray(image), "gt_mask": np.ascontiguousarray(gt_mask), "name": sample.get("name", "")}


# ----------------------------------------------------------------------- labels
def _labels_from_counts(counts: np.ndarray, fg_sum: np.ndarray, fg_threshold: float, bg_threshold: float):
    """(labels int64, fg_ratio float32) from the integer region statistics, with the reference's arithmetic:
    float64 ratio for the labels (dataset.py:197-205), float32-count denominator for fg_ratio (:246-248)."""
    c64 = counts.astype(np.float64)
    ratio = fg_sum.astype(np.float64) / np.maximum(c64, 1.0)
    labels = np.full(counts.shape[0], CLASS_UNK, dtype=np.int64)
    labels[ratio >= fg_threshold] = CLASS_FG
    labels[ratio <= 1 - bg_threshold] = CLASS_BG
    labels[counts == 0] = CLASS_UNK
    c32 = counts.astype(np.float32)
    fg_ratio = (fg_sum.astype(np.float64) / np.maximum(c32, 1.0)).astype(np.float32)
    return labels, fg_ratio


def derive_trimap_labels(segments: np.ndarray, gt_mask: np.ndarray, fg_threshold: float = 0.75,
                         bg_threshold: float = 0.75) -> np.ndarray:
    """Per-superpixel trimap class by foreground coverage — reference dataset.py:175-206.  Host form (integer
    bincounts, identical result); prepare_dataset uses the device reduction instead."""
    n_nodes = int(segments.max()) + 1
    flat = segments.ravel()
    counts = np.bincount(flat, minlength=n_nodes)
    fg_sum = np.bincount(flat, weights=(gt_mask.ravel() > 0).astype(np.float64), minlength=n_nodes).astype(np.int64)
    return _labels_from_counts(counts, fg_sum, fg_threshold, bg_threshold)[0]


# ----------------------------------------------------------------------- cache
def _cache_key(sample: dict, sp_config: Optional[SuperpixelGraphConfig], fg_threshold: float, bg_threshold: float) -> str:
    """File name of a cache entry — the reference's recipe (dataset.py:364-378)."""
    cfg = sp_config or SuperpixelGraphConfig()
    h = hashlib.sha1()
    if "image" in sample:
        h.update(np.ascontiguousarray(sample["image"]))
        h.update(np.ascontiguousarray(sample["gt_mask"]))
    else:
        h.update(repr((sample["image_path"], sample["mask_path"], sample.get("max_size"), sample.get("aug_seed"))).encode())
    h.update(repr((cfg.n_segments, cfg.compactness, cfg.sigma, cfg.use_lab, cfg.connectivity, cfg.n_nonlocal,
                   fg_threshold, bg_threshold)).encode())
    return h.hexdigest()[:20]


def _to_data(blob: dict) -> Data:
    d = blob["data"]
    return Data(x=d["x"], edge_index=d["edge_index"], edge_attr=d["edge_attr"], node_area=d["node_area"],
                fg_ratio=d["fg_ratio"], y=d["y"])


def load_cache_entry(path) -> Optional[tuple]:
    """(Data, labels, segments-or-None) of one cache file, or None when it is missing, stale or not ours."""
    try:
        blob = torch.load(path, map_location="cpu", weights_only=True)
        if blob.get("format") != CACHE_FORMAT:
            return None
        data = _to_data(blob)
        return data, data.y, (blob["segments"].numpy() if blob.get("segments") is not None else None)
    except Exception:            # corrupt or foreign entry: rebuild it
        return None


def _write_cache_entry(path: Path, data: Data, segments: Optional[np.ndarray]) -> None:
    path.parent.mkdir(parents=True, exist_ok=True)
    tmp = path.with_suffix(f".{os.getpid()}.tmp")
    blob = {"format": CACHE_FORMAT,
            "data": {k: getattr(data, k) for k in ("x", "edge_index", "edge_attr", "node_area", "fg_ratio", "y")},
            "segments": None if segments is None else torch.from_numpy(np.ascontiguousarray(segments))}
    try:
        torch.save(blob, tmp)
        os.replace(tmp, path)              # never leaves a truncated entry behind
    except Exception:
        tmp.unlink(missing_ok=True)


# ----------------------------------------------------------------------- graph factory
def _build_batch(eng, images: np.ndarray, gt_masks: np.ndarray, cfg: SuperpixelGraphConfig, fg_t: float, bg_t: float):
    """G0-G8 + label statistics for a batch of equally sized images; returns host records."""
    b, h, w, _ = images.shape
    bgr = eng.to_device(images)
    lab, hsv, _gray, grad = eng.preprocess(bgr)
    seg, n_nodes = eng.slic(lab, cfg.n_segments, cfg.compactness, cfg.sigma) if cfg.use_lab else \
        eng.slic_rgb(bgr, cfg.n_segments, cfg.compactness, cfg.sigma)              # reference graph_builder.py:177-179
    g = eng.build_graphs(seg, n_nodes, lab, hsv, grad, cfg.connectivity, cfg.n_nonlocal)
    n_total = int(g.node_ptr_host[-1])
    counts = eng.empty(max(n_total, 1), dtype=torch.int32)
    fg = eng.empty(max(n_total, 1), dtype=torch.int32)
    gt = eng.to_device(gt_masks)
    eng.ctx.call("ggc_region_label_stats", eng._stream(), b, h, w, seg.data_ptr(), gt.data_ptr(), g.node_ptr.data_ptr(),
                 counts.data_ptr(), fg.data_ptr())
    x, ea, area = g.x.cpu(), g.edge_attr.cpu(), g.area_ratio.cpu()
    src, dst = g.edge_src.cpu().to(torch.int64), g.edge_dst.cpu().to(torch.int64)
    counts_h, fg_h, seg_h = counts.cpu().numpy(), fg.cpu().numpy(), seg.cpu().numpy()
    out = []
    for i in range(b):
        n0, n1 = int(g.node_ptr_host[i]), int(g.node_ptr_host[i + 1])
        e0, e1 = int(g.edge_ptr_host[i]), int(g.edge_ptr_host[i + 1])
        labels, fg_ratio = _labels_from_counts(counts_h[n0:n1], fg_h[n0:n1], fg_t, bg_t)
        data = Data(x=x[n0:n1].clone(), edge_index=torch.stack([src[e0:e1] - n0, dst[e0:e1] - n0]),
                    edge_attr=ea[e0:e1].clone().reshape(-1, 5), node_area=area[n0:n1].clone(),
                    fg_ratio=torch.from_numpy(fg_ratio), y=torch.from_numpy(labels))
        out.append((data, data.y, seg_h[i]))
    return out


def prepare_sample(sample: dict, sp_config: Optional[SuperpixelGraphConfig] = None, fg_threshold: float = 0.70,
                   bg_threshold: float = 0.70, device: str = "cuda") -> tuple:
    """One sample -> (Data, labels, segments) — reference dataset.py:213-260."""
    from ._engine import get_engine
    cfg = sp_config or SuperpixelGraphConfig()
    rec = _build_batch(get_engine(device), sample["image"][None], sample["gt_mask"][None], cfg, fg_threshold, bg_threshold)
    return rec[0]


def prepare_dataset(samples: list[dict], sp_config: Optional[SuperpixelGraphConfig] = None, fg_threshold: float = 0.70,
                    bg_threshold: float = 0.70, cache_dir=None, workers: int = 0, desc: str = "",
                    keep_segments: bool = True, batch_size: int = 64, device: str = "cuda") -> list[tuple]:
    """
    Build (or fetch from the cache) the graph of every sample — reference dataset.py:444-540.

    `workers` decode threads feed batches of up to `batch_size` equally sized images to the device graph builder.
    Returns [(Data, labels, segments-or-None)] in sample order; samples that fail to decode are dropped, like in
    the reference.
    """
    from ._engine import get_engine
    cfg = sp_config or SuperpixelGraphConfig()
    eng = get_engine(device)
    cache_dir = Path(cache_dir) if cache_dir else None
    t0 = time.perf_counter()
    records: dict[int, tuple] = {}
    todo: list[int] = []
    paths: dict[int, Path] = {}
    for i, s in enumerate(samples):
        if cache_dir is not None:
            paths[i] = cache_dir / f"{_cache_key(s, cfg, fg_threshold, bg_threshold)}.pt"
            hit = load_cache_entry(paths[i]) if paths[i].exists() else None
            if hit is not None:
                records[i] = (hit[0], hit[1], hit[2] if keep_segments else None)
                continue
        todo.append(i)
    n_hits, failures = len(records), 0

    def flush(shape, idx, imgs, gts):
        recs = _build_batch(eng, np.stack(imgs), np.stack(gts), cfg, fg_threshold, bg_threshold)
        for i, (data, labels, seg) in zip(idx, recs):
            if i in paths:
                _write_cache_entry(paths[i], data, seg)
            records[i] = (data, labels, seg if keep_segments else None)

    pending: dict[tuple, tuple[list, list, list]] = {}
    pool = ThreadPoolExecutor(max_workers=max(1, int(workers or 1)))
    try:
        for i, m in zip(todo, pool.map(lambda j: materialise(samples[j]), todo)):
            if m is None:
                failures += 1
                continue
            key = m["image"].shape[:2]
            idx, imgs, gts = pending.setdefault(key, ([], [], []))
            idx.append(i); imgs.append(m["image"]); gts.append(m["gt_mask"])
            if len(idx) >= batch_size:
                flush(key, *pending.pop(key))
        for key in list(pending):
            flush(key, *pending.pop(key))
    finally:
        pool.shutdown()
    if desc or failures:
        dt = time.perf_counter() - t0
        print(f"{desc}{len(records)} graphs ({n_hits} from cache, {failures} dropped) in {dt:.1f} s")
    return [records[i] for i in sorted(records)]
