"""
Build the superpixel graphs of a dataset on the MI355X and store them in the cache
(the reference's tools/prepare_graphs.py, same flags).

    python3 tools/prepare_graphs.py --images DIR --masks DIR --cache DIR --workers 6 --max-size 384

`--workers` are decode threads; the graphs themselves are built in device batches of `--batch`
equally sized images.  `--augment N` adds N seeded augmented copies per image (host-side, see gcn_grabcut/dataset.py).
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

_ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(_ROOT))
sys.path.insert(0, str(_ROOT / "gcn-grabcut_amd"))

from src.gcn_grabcut.dataset import list_image_mask_pairs, prepare_dataset   # noqa: E402
from src.gcn_grabcut.graph_builder import SuperpixelGraphConfig                # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser(description="Warm the graph cache")
    ap.add_argument("--images", required=True)
    ap.add_argument("--masks", required=True)
    ap.add_argument("--cache", required=True)
    ap.add_argument("--workers", type=int, default=4)
    ap.add_argument("--max-size", type=int, default=384)
    ap.add_argument("--superpixels", type=int, default=300)
    ap.add_argument("--augment", type=int, default=0, help="Seeded augmented copies per image")
    ap.add_argument("--limit", type=int, default=0)
    ap.add_argument("--stride", type=int, default=1, help="Take every n-th sample, to cover a split sparsely")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--batch", type=int, default=64, help="images of one size per device batch (additive flag)")
    args = ap.parse_args()

    samples = list_image_mask_pairs(args.images, args.masks, max_size=args.max_size, augment_copies=args.augment,
                                    seed=args.seed)
    if args.stride > 1:
        samples = samples[::args.stride]
    if args.limit:
        samples = samples[:args.limit]
    # nothing is returned to the caller: the run exists to fill the cache directory
    prepare_dataset(samples, SuperpixelGraphConfig(n_segments=args.superpixels), cache_dir=args.cache,
                    workers=args.workers, keep_segments=False, desc="cache: ", batch_size=args.batch)


if __name__ == "__main__":
    main()
