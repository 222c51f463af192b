"""
All-cores leg of bench.py's cpu_baseline: one worker process per host core, one image per task (the reference's own
pattern for bulk CPU work: dataset.py:496-520, a spawn pool with one thread per worker).

TEST / BENCH INFRASTRUCTURE ONLY (like the rest of oracle/).  Workers import numpy + the ctypes oracle only: no torch, no
GPU.  Images are regenerated in the worker from their seed, so a task ships a few integers and returns either a small
timing record or, for the first `n_return` images, the arrays the parity check needs.
"""
from __future__ import annotations

import importlib.util
import sys
import time
from pathlib import Path

_ROOT = Path(__file__).resolve().parent.parent
_state = {}


def _init(state_np: dict, hidden: int, layers: int, n_segments: int, h: int, w: int, config_id: int):
    if str(_ROOT) not in sys.path:
        sys.path.insert(0, str(_ROOT))
    from oracle import oracle as orc
    # the generator is one pure-numpy file of the package: loaded by path, so that the worker does not import torch
    spec = importlib.util.spec_from_file_location("ggc_synthetic", _ROOT / "gcn-grabcut_amd" / "gcn_grabcut" / "synthetic.py")
    syn = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(syn)
    orc.lib()
    _state.update(orc=orc, syn=syn, sd=state_np, hidden=hidden, layers=layers, n_seg=n_segments, h=h, w=w, cfg=config_id)


def _one(task):
    index, want_arrays = task
    s = _state
    img = s["syn"].synthetic_image(s["h"], s["w"], 10_000 * s["cfg"] + index)
    timing = {}
    t0 = time.perf_counter()
    r = s["orc"].segment(img, s["sd"], s["hidden"], s["layers"], n_segments=s["n_seg"], seed=index, timing=timing)
    dt = time.perf_counter() - t0
    out = {"index": index, "seconds": dt, "timing": timing}
    if want_arrays:
        out.update(segments=r["segments"], trimap=r["trimap"], binary_mask=r["binary_mask"], probs=r["probs"],
                   edge_index=r["graph"]["edge_index"])
    return out


def run(indices, n_return: int, n_workers: int, state_np: dict, hidden: int, layers: int, n_segments: int, h: int, w: int,
        config_id: int):
    """-> (wall seconds of the pool's map, results in index order).  n_workers == 1 runs in this process."""
    tasks = [(i, k < n_return) for k, i in enumerate(indices)]
    if n_workers <= 1:
        _init(state_np, hidden, layers, n_segments, h, w, config_id)
        t0 = time.perf_counter()
        res = [_one(t) for t in tasks]
        return time.perf_counter() - t0, res
    import multiprocessing as mp
    ctx = mp.get_context("spawn")                         # the parent holds a GPU context: never fork it
    with ctx.Pool(n_workers, initializer=_init, initargs=(state_np, hidden, layers, n_segments, h, w, config_id)) as pool:
        pool.map(_noop, range(n_workers * 2))             # workers up and the library loaded before the clock starts
        t0 = time.perf_counter()
        res = pool.map(_one, tasks, chunksize=1)
        dt = time.perf_counter() - t0
    return dt, res


def _noop(_):
    return 0
