"""
inference.py — automatic segmentation with GCN-GrabCut on MI355X.

Same command line as the reference's inference.py (flags :26-56, outputs :146-155):

    python3 inference.py --image path/to/image.jpg
    python3 inference.py --input path/to/folder --output results/
    python3 inference.py --image cat.jpg --keep-largest --save mask overlay

Images are decoded / written with Pillow (OpenCV is not a dependency of this build); folders are processed in
batches of equally sized images so that the whole batch stays resident in HBM.
"""
import argparse
import time
from pathlib import Path

import numpy as np

IMAGE_EXTS = {".jpg", ".jpeg", ".png", ".bmp", ".tif", ".tiff", ".webp"}


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description="Automatic GCN-GrabCut segmentation (MI355X)")
    src = parser.add_mutually_exclusive_group(required=True)
    src.add_argument("--image", help="Path to a single input image")
    src.add_argument("--input", help="Directory of images to segment")
    parser.add_argument("--output", default="results", help="Output directory")
    parser.add_argument("--checkpoint", default="checkpoints/best_model.pt")
    parser.add_argument("--model", default="resgcn", choices=["resgcn", "gcn", "gat"])
    parser.add_argument("--device", default="cuda")
    parser.add_argument("--threshold", type=float, default=0.55, help="Softmax threshold for definite FG/BG superpixels")
    parser.add_argument("--max-size", type=int, default=800, help="Resize longest edge to this before segmenting (0 = off)")
    parser.add_argument("--refine", type=int, default=0, help="Extra GrabCut refinement iterations")
    parser.add_argument("--superpixels", type=int, default=300)
    parser.add_argument("--hidden", type=int, default=128)
    parser.add_argument("--layers", type=int, default=6)
    parser.add_argument("--no-edge-aware", action="store_true",
                        help="Threshold region probabilities directly instead of projecting them through a guided filter")
    parser.add_argument("--filter-radius", type=int, default=8, help="Guided-filter radius for the edge-aware trimap")
    parser.add_argument("--min-area", type=float, default=0.002,
                        help="Drop mask components smaller than this fraction of the image")
    parser.add_argument("--keep-largest", action="store_true", help="Keep only the largest connected component")
    parser.add_argument("--save", nargs="+", default=["mask", "overlay"], choices=["mask", "overlay", "rgba", "trimap"],
                        help="Which outputs to write")
    parser.add_argument("--batch", type=int, default=64, help="Images per device batch (additive flag)")
    return parser


def read_bgr(path: Path, max_size: int):
    from PIL import Image
    try:
        im = Image.open(path).convert("RGB")
    except Exception:
        return None
    if max_size > 0:
        w, h = im.size
        scale = min(max_size / max(h, w), 1.0)
        if scale < 1.0:
            im = im.resize((int(w * scale), int(h * scale)), Image.BOX)   # area averaging, as cv2.INTER_AREA
    return np.ascontiguousarray(np.asarray(im)[:, :, ::-1])


def main() -> None:
    args = build_parser().parse_args()
    import torch
    from src.gcn_grabcut import GCNGrabCutPipeline
    from src.gcn_grabcut.graph_builder import SuperpixelGraphConfig
    from src.gcn_grabcut.model import GATTrimapNet, GCNTrimapNet, ResGCNNet
    from src.gcn_grabcut.pipeline import _colour_trimap, _write_png

    if not torch.cuda.is_available():
        raise SystemExit("[inference] no MI355X visible: this build has no CPU path")
    model_cls = {"resgcn": ResGCNNet, "gcn": GCNTrimapNet, "gat": GATTrimapNet}[args.model]

    ckpt_path = Path(args.checkpoint)
    if not ckpt_path.exists():
        fallback = Path("checkpoints/final_model.pt")
        if not fallback.exists():
            raise FileNotFoundError(f"No checkpoint at {ckpt_path} (or {fallback}). Train one with the reference's train.py.")
        print(f"[inference] {ckpt_path} not found, using {fallback}")
        ckpt_path = fallback
    state = torch.load(ckpt_path, map_location="cpu", weights_only=True)["model"]
    # width and depth are recovered from the checkpoint (reference inference.py:81-86)
    hidden = state["input_proj.0.weight"].shape[0] if "input_proj.0.weight" in state else args.hidden
    layers = (sum(1 for k in state if k.startswith("gcn_layers.") and k.endswith(".bias"))
              or sum(1 for k in state if k.startswith("blocks.") and k.endswith(".conv.bias"))
              or sum(1 for k in state if k.startswith("convs.") and k.endswith(".att")) or args.layers)
    model = model_cls(hidden_channels=hidden, n_layers=layers)
    model.load_state_dict(state)
    model.eval()
    print(f"[inference] loaded {model_cls.__name__} (D={hidden}, n={layers}) from {ckpt_path} on {args.device}")

    pipeline = GCNGrabCutPipeline(model, sp_config=SuperpixelGraphConfig(n_segments=args.superpixels), device=args.device)

    if args.image:
        paths = [Path(args.image)]
    else:
        in_dir = Path(args.input)
        paths = sorted(p for p in in_dir.iterdir() if p.suffix.lower() in IMAGE_EXTS)
        if not paths:
            raise FileNotFoundError(f"No images found in {in_dir}")
    out_dir = Path(args.output)
    out_dir.mkdir(parents=True, exist_ok=True)

    by_shape: dict = {}
    for path in paths:
        image = read_bgr(path, args.max_size)
        if image is None:
            print(f"[inference] skipping unreadable file: {path}")
            continue
        by_shape.setdefault(image.shape, []).append((path, image))

    total_t, n_done = 0.0, 0
    for items in by_shape.values():
        for i in range(0, len(items), args.batch):
            chunk = items[i:i + args.batch]
            t0 = time.perf_counter()
            results = pipeline.segment_batch(
                [im for _, im in chunk], threshold_fg=args.threshold, threshold_bg=args.threshold,
                refine_iters=args.refine, min_area_ratio=args.min_area, keep_largest=args.keep_largest,
                edge_aware=not args.no_edge_aware, filter_radius=args.filter_radius)
            elapsed = (time.perf_counter() - t0) / len(chunk)
            for (path, _), result in zip(chunk, results):
                total_t += elapsed
                n_done += 1
                stem = out_dir / path.stem
                if "mask" in args.save:
                    _write_png(f"{stem}_mask.png", result.binary_mask * 255)
                if "overlay" in args.save:
                    _write_png(f"{stem}_overlay.png", result.overlay)
                if "rgba" in args.save:
                    _write_png(f"{stem}_rgba.png", result.rgba)
                if "trimap" in args.save:
                    _write_png(f"{stem}_trimap.png", _colour_trimap(result.trimap))
                t = result.timing
                print(f"[{n_done}/{len(paths)}] {path.name}  fg={result.binary_mask.mean():.1%}  "
                      f"graph={t.get('graph_build', 0):.4f}s gcn={t.get('gcn_inference', 0):.4f}s "
                      f"grabcut={t.get('grabcut', 0):.4f}s  total={elapsed:.4f}s")
    if n_done:
        print(f"\n[inference] {n_done} image(s) → {out_dir}/  ({total_t / n_done:.4f}s per image)")
    else:
        print("[inference] nothing to do.")


if __name__ == "__main__":
    main()
