"""CPU test of the N>1 layout with world_size 2 over gloo: contiguous image shards per rank, no data-path
collective, only the timing / parity records are gathered.  The per-rank "work" is the
CPU oracle on tiny images so the test needs no GPU."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def shard(n_images: int, rank: int, world: int) -> range:
    """images [rank*per, (rank+1)*per) — the same rule bench.py uses (first_index = rank * batch)."""
    per = n_images // world
    return range(rank * per, (rank + 1) * per)


def _worker(rank: int, world: int, port: int, out_dir: str):
    for p in (ROOT, ROOT / "src", ROOT / "tests"):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gcn_grabcut.synthetic import synthetic_image
    from oracle import oracle as orc
    mine = shard(8, rank, world)
    fg = []
    for i in mine:
        img = synthetic_image(24, 32, 40_000 + i)
        lab, hsv, gray, grad = orc.preprocess(img)
        seg, n = orc.slic(lab, 12)
        fg.append(float(n))
    dist.barrier()
    rec = torch.tensor([len(mine), 0.25 * (rank + 1), sum(fg)], dtype=torch.float64)   # n_images, seconds, checksum
    gathered = [torch.zeros_like(rec) for _ in range(world)]
    dist.all_gather(gathered, rec)
    t = torch.tensor([rec[1].item()], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        np.save(os.path.join(out_dir, "gathered.npy"), torch.stack(gathered).numpy())
        np.save(os.path.join(out_dir, "tmax.npy"), t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather(tmp_path, oracle):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g = np.load(tmp_path / "gathered.npy")
    assert g.shape == (2, 3)
    assert g[:, 0].sum() == 8                                  # every image processed exactly once
    assert np.load(tmp_path / "tmax.npy")[0] == pytest.approx(0.5)   # step time = max over ranks
    # shards are disjoint, contiguous and cover the list
    assert sorted(list(shard(8, 0, 2)) + list(shard(8, 1, 2))) == list(range(8))
    # throughput = all images / slowest rank
    assert g[:, 0].sum() / 0.5 == pytest.approx(16.0)
