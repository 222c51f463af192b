"""
Multi-GPU layout of the hot path (the reference has no counterpart: it is single-process).

Images are independent from decode to mask, so the path shards with NO data-path collective: one process per GPU, rank r
takes a contiguous block of the image list, the weights (0.75 MB) are replicated.  The only communication is one
all-gather, after the timed region, of a fixed 64-byte record per rank (RCCL over xGMI under torch.distributed's "nccl"
backend; "gloo" in the CPU tests), from which rank 0 forms the job's throughput (all images / slowest rank) and the
parity tallies.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict

RECORD_FIELDS = ("n_images", "seconds", "sum_iou", "n_iou", "n_trimap_exact", "n_label_exact", "n_mask_exact", "n_checked")


def shard_range(n_total: int, rank: int, world: int) -> range:
    """Contiguous block of rank `rank`: sizes differ by at most one, blocks are disjoint and cover range(n_total)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


@dataclass
class RankRecord:
    """What a rank reports after its timed region (8 x f64 = 64 bytes on the wire)."""
    n_images: float = 0.0        # images this rank pushed through the path in the timed region
    seconds: float = 0.0         # its wall time for them
    sum_iou: float = 0.0         # sum of mask IoU against the CPU oracle over the images it checked
    n_iou: float = 0.0
    n_trimap_exact: float = 0.0  # checked images whose trimap equals the oracle's bit for bit
    n_label_exact: float = 0.0   # ... whose SLIC label map does
    n_mask_exact: float = 0.0
    n_checked: float = 0.0

    def as_list(self) -> list[float]:
        return [float(getattr(self, k)) for k in RECORD_FIELDS]


def gather_records(record: RankRecord, device=None) -> list[RankRecord]:
    """All ranks' records, in rank order, on every rank.  A single process (no initialised group) returns [record]."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [record]
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.tensor(record.as_list(), dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [RankRecord(**dict(zip(RECORD_FIELDS, t.cpu().tolist()))) for t in out]


def summarise(records: list[RankRecord]) -> dict:
    """Whole-job figures from the gathered records: throughput = all images / the slowest rank's time."""
    n = sum(r.n_images for r in records)
    t = max((r.seconds for r in records), default=0.0)
    chk = sum(r.n_checked for r in records)
    n_iou = sum(r.n_iou for r in records)
    return {
        "n_images": int(n), "seconds_max_over_ranks": t, "images_per_s": (n / t) if t > 0 else 0.0,
        "per_rank": [asdict(r) for r in records],
        "checked_images": int(chk),
        "mean_mask_iou": (sum(r.sum_iou for r in records) / n_iou) if n_iou else None,
        "trimap_exact_pct": (100.0 * sum(r.n_trimap_exact for r in records) / chk) if chk else None,
        "label_map_exact_pct": (100.0 * sum(r.n_label_exact for r in records) / chk) if chk else None,
        "mask_exact_pct": (100.0 * sum(r.n_mask_exact for r in records) / chk) if chk else None,
    }
