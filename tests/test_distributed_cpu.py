"""CPU test of the N>1 layout with world_size 2 over gloo, driving the PRODUCT's sharding rule and record gather
(gcn_grabcut/distributed.py, used by bench.py): contiguous image shards per rank, no data-path collective, one all-gather of
the 64-byte per-rank record.  The per-rank "work" is the CPU oracle on tiny images so the test needs no GPU."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank: int, world: int, port: int, out_dir: str):
    for p in (ROOT, ROOT / "src", ROOT / "tests"):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gcn_grabcut.distributed import RankRecord, gather_records, shard_range, summarise
    from gcn_grabcut.synthetic import synthetic_image
    from oracle import oracle as orc
    mine = shard_range(9, rank, world)                    # 9 images over 2 ranks: 5 + 4
    n_regions = 0
    for i in mine:
        img = synthetic_image(24, 32, 40_000 + i)
        lab, hsv, gray, grad = orc.preprocess(img)
        seg, n = orc.slic(lab, 12)
        n_regions += int(n)
    dist.barrier()
    rec = RankRecord(n_images=len(mine), seconds=0.25 * (rank + 1), sum_iou=0.9 * len(mine), n_iou=len(mine),
                     n_trimap_exact=len(mine) - rank, n_label_exact=len(mine), n_mask_exact=len(mine), n_checked=len(mine))
    records = gather_records(rec)                         # the call bench.py makes after its timed region
    s = summarise(records)
    if rank == 0:
        np.save(os.path.join(out_dir, "summary.npy"), np.array([s["n_images"], s["seconds_max_over_ranks"], s["images_per_s"],
                                                                 s["mean_mask_iou"], s["trimap_exact_pct"], s["label_map_exact_pct"],
                                                                 len(records), records[1].n_images]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather(tmp_path, oracle):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    n, tmax, ips, iou, tri, lab, n_rec, n1 = np.load(tmp_path / "summary.npy")
    assert n == 9 and n_rec == 2 and n1 == 4              # every image processed exactly once, records in rank order
    assert tmax == pytest.approx(0.5)                     # step time = max over ranks
    assert ips == pytest.approx(18.0)                     # throughput = all images / slowest rank
    assert iou == pytest.approx(0.9) and lab == pytest.approx(100.0)
    assert tri == pytest.approx(100.0 * 8 / 9)            # rank 1 reported one inexact trimap


def test_shard_rule():
    from gcn_grabcut.distributed import RankRecord, gather_records, shard_range, summarise
    for n, w in ((2048, 8), (512, 8), (9, 2), (5, 8), (0, 3)):
        blocks = [list(shard_range(n, r, w)) for r in range(w)]
        assert sum(blocks, []) == list(range(n))          # disjoint, contiguous, in order, complete
        assert max(map(len, blocks)) - min(map(len, blocks)) <= 1
    assert list(shard_range(2048, 3, 8)) == list(range(768, 1024))      # configs[3]: 256 images per GPU
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)
    one = gather_records(RankRecord(n_images=4, seconds=2.0))            # no process group: a job of one
    assert len(one) == 1 and summarise(one)["images_per_s"] == 2.0 and summarise(one)["mean_mask_iou"] is None
