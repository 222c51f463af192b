#!/usr/bin/env python3
"""
bench.py — throughput of the GCN-GrabCut hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A "step" is one pass of the hot path over one per-GPU batch of synthetic
DUTS-shaped images that are already resident in HBM:

    --workload full  (default)  configs[2]: SLIC -> graph -> ResGCNNet -> guided-filter
                                trimap -> GrabCut -> clean-up, batch 256 of 400x300 per GPU
    --workload gcn              configs[1]: ResGCNNet forward only on 64 pre-built graphs

Rank 0 prints ONE JSON line: BASELINE.json's metric (images/s), the roofline of
the dominant kernel named by the north star (the GCNConv scatter-gather, timed
with HIP events on its launch stream inside the timed region), parity against
the CPU oracle on a sample, and the oracle timed on that bounded sample.
Weights are a seeded random init (the reference ships no checkpoint).  Images
shard across ranks with no data-path collective; RCCL only gathers the timing.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
for _p in (ROOT, ROOT / "src"):
    if str(_p) not in sys.path:
        sys.path.insert(0, str(_p))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
HIDDEN, LAYERS = 128, 6    # ResGCNNet default (reference model.py:453-454)
H, W, N_SEGMENTS = 300, 400, 600


def synthetic_region_graph(n: int, rng: np.random.Generator, k_nl: int = 4):
    """Region graph of DUTS shape without running SLIC (for --workload gcn): jittered grid of
    n regions, 4-neighbour adjacency + most diagonals, k non-local colour neighbours per node,
    mirrored as in reference graph_builder.py:303-306.  n = 600 -> E ~ 6.4k directed edges."""
    gw = int(round(np.sqrt(n * 4 / 3)))
    gh = int(np.ceil(n / gw))
    ids = np.arange(gh * gw).reshape(gh, gw)
    lo, hi = [], []
    for a, b, p in ((ids[:, :-1], ids[:, 1:], 1.0), (ids[:-1, :], ids[1:, :], 1.0),
                    (ids[:-1, :-1], ids[1:, 1:], 0.9)):
        ok = (a < n) & (b < n) & (rng.random(a.shape) < p)
        lo.append(np.minimum(a[ok], b[ok])); hi.append(np.maximum(a[ok], b[ok]))
    lo, hi = np.concatenate(lo), np.concatenate(hi)
    adj = np.unique(lo.astype(np.int64) * n + hi)
    col = rng.random((n, 3), dtype=np.float32)
    d = np.linalg.norm(col[:, None] - col[None], axis=2)
    np.fill_diagonal(d, np.inf)
    d[adj // n, adj % n] = np.inf
    d[adj % n, adj // n] = np.inf
    nb = np.argpartition(d, k_nl - 1, axis=1)[:, :k_nl]
    rows = np.repeat(np.arange(n), k_nl)
    nl = np.unique(np.minimum(rows, nb.ravel()).astype(np.int64) * n + np.maximum(rows, nb.ravel()))
    codes = np.concatenate([adj, nl])
    pr = np.stack([codes // n, codes % n], 1)
    src = np.concatenate([pr[:, 0], pr[:, 1]])
    dst = np.concatenate([pr[:, 1], pr[:, 0]])
    attr = rng.random((len(pr), 5), dtype=np.float32)
    attr[: len(adj), 4] = 0.0
    attr[len(adj):, 4] = 1.0
    attr[len(adj):, 2] = 0.0
    x = rng.random((n, 19), dtype=np.float32)
    return x, np.stack([src, dst]), np.concatenate([attr, attr], 0)


def agg_bytes(n_nodes: int, n_edges: int, d: int, n_graphs: int) -> int:
    """Algorithmic HBM bytes of one fused GCNConv aggregation launch:
    read XW, gate, h and write h' (4 N D f32) + CSR col incl. self loops + row_ptr + dis + bias."""
    return (4 * n_nodes * d * 4 + (n_edges + n_nodes) * 4 + (n_nodes + n_graphs) * 4
            + n_nodes * 4 + d * 4)


def pmc_traffic(workload: str, batch: int):
    """HBM bytes per launch of the graded kernel from the PMC counters.  Counters cannot be read from inside a
    timed run (rocprofv3 --pmc serialises every dispatch), so this is the committed summary of the separate
    FETCH_SIZE / WRITE_SIZE passes over this same command (profiles/r01_pmc_bench_hbm.json, tools/pmc_bench.sh);
    null for any other workload or batch size."""
    if workload != "full" or batch != 256:
        return None
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_bench_hbm.json")) as fh:
            return int(json.load(fh)["kernels"]["k_aggregate_graph<128, 0, 32, true>"]["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["full", "gcn"], default="full")
    ap.add_argument("--lanes", type=int, default=0, help="concurrent sub-batches of the GrabCut stage (full workload; "
                    "default 4 with one pipeline, 1 per pipeline otherwise)")
    ap.add_argument("--overlap-pass", type=int, default=0, help="after the contract's run (N=1, full workload, one pipeline): a second "
                    "timed pass of the same steps over this many overlapping pipelines, reported as 'overlapped' (0 = skip, the "
                    "default: the pass re-runs every kernel under contention, which would blur a rocprofv3 summary of the command)")
    ap.add_argument("--pipelines", type=int, default=1, help="full workload: pipelines (private contexts, own HIP streams and "
                    "host threads) that take the timed steps in turn, so consecutive batches overlap")
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per step (default 256 full / 64 gcn)")
    ap.add_argument("--cpu-sample", type=int, default=40, help="images timed on the CPU oracle, ~12 s of one core (0 = skip)")
    args = ap.parse_args()
    batch_size = args.batch or (256 if args.workload == "full" else 64)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI; gathers timings only

    from gcn_grabcut import _native
    from gcn_grabcut.data import Batch, Data
    from gcn_grabcut.graph_builder import SuperpixelGraphConfig
    from gcn_grabcut.model import ResGCNNet
    from gcn_grabcut.pipeline import GCNGrabCutPipeline
    from gcn_grabcut.synthetic import synthetic_batch

    # ---- model: seeded random init (reference model.py:501-506); no checkpoint ships
    torch.manual_seed(0)
    model = ResGCNNet(hidden_channels=HIDDEN, n_layers=LAYERS).to(dev).eval()
    ctx = _native.get_context(local_rank)

    host_imgs = host_graphs = None
    if args.workload == "full":
        # this rank's shard: images rank*B .. rank*B + B - 1 of config 3 (seeds 30000 + index)
        host_imgs = synthetic_batch(batch_size, H, W, config_id=3, first_index=rank * batch_size)
        n_pipes = max(1, args.pipelines)
        lanes = args.lanes if args.lanes > 0 else (4 if n_pipes == 1 else 1)
        pipe = GCNGrabCutPipeline(model, sp_config=SuperpixelGraphConfig(n_segments=N_SEGMENTS), device=f"cuda:{local_rank}",
                                  grabcut_lanes=lanes)
        pipes = [pipe] + [pipe.replica(grabcut_lanes=lanes) for _ in range(n_pipes - 1)]
        streams = [torch.cuda.Stream(dev) for _ in range(n_pipes)] if n_pipes > 1 else None
        bgr = torch.from_numpy(host_imgs).to(dev)
        last = {}

        def step():
            last["out"] = pipe.segment_batch_device(bgr, compose=True)

        def run_steps(n):
            """n passes over the batch; with several pipelines, pipeline i takes steps i, i + P, ... on its own stream"""
            if n_pipes == 1:
                for _ in range(n):
                    step()
                return
            import threading

            def worker(i):
                torch.cuda.set_device(dev)
                with torch.cuda.stream(streams[i]):
                    for s in range(i, n, n_pipes):
                        out = pipes[i].segment_batch_device(bgr, compose=True)
                        if s == n - 1:
                            last["out"] = out
                streams[i].synchronize()

            threads = [threading.Thread(target=worker, args=(i,), name=f"ggc-pipe{i}") for i in range(n_pipes)]
            for th in threads:
                th.start()
            for th in threads:
                th.join()
    else:
        rng = np.random.default_rng(20_000 + rank)
        host_graphs = [synthetic_region_graph(int(rng.integers(585, 618)), rng) for _ in range(batch_size)]
        batch = Batch.from_data_list([Data(x=torch.from_numpy(x), edge_index=torch.from_numpy(ei),
                                           edge_attr=torch.from_numpy(ea)) for x, ei, ea in host_graphs]).to(dev)
        batch.node_ptr32 = batch.ptr.to(torch.int32)
        last = {}

        n_pipes, lanes = 1, 0

        def step():
            last["out"] = model.predict_probs_device(batch)

        def run_steps(n):
            for _ in range(n):
                step()

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    warmup = max(args.warmup, n_pipes) if n_pipes > 1 else args.warmup     # every pipeline allocates its arena once
    run_steps(warmup)
    # the GrabCut stage runs on concurrent lanes with private contexts (created during warm-up): profile them all
    ctxs = [c for p_ in pipes for c in p_._eng.all_contexts()] if args.workload == "full" else [ctx]
    for c in ctxs:
        c.profile_enable(True)
    sync_all()
    t0 = time.perf_counter()
    run_steps(args.steps)
    sync_all()
    elapsed = time.perf_counter() - t0
    prof = {}
    for k in ("gcn_aggregate", "gcn_gemm", "slic_assign", "slic_update", "slic_connectivity", "graph_stats",
              "graph_knn", "graph_prior", "refine_trimap", "grabcut_init_gmm", "grabcut_gmm", "maxflow_relabel",
              "maxflow_push"):
        q = [c.profile_query(k) for c in ctxs]       # (launch scopes, ms); lanes overlap, so stage times can exceed the step
        prof[k] = (sum(v[0] for v in q), sum(v[1] for v in q))
    for c in ctxs:
        c.profile_enable(False)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- informational second pass: consecutive batches overlapped (batch k+1's SLIC / graph / GCN under batch k's GrabCut).
    # Not the contract's number: per-kernel event timing is meaningless under overlap, so `value` and `roofline` stay serial.
    overlapped = None
    if world == 1 and args.workload == "full" and n_pipes == 1 and args.overlap_pass > 1:
        try:
            serial_out = last["out"]
            serial_mask = serial_out["binary_mask"].clone()
            n_pipes, keep_lanes = args.overlap_pass, pipe.grabcut_lanes
            pipe.grabcut_lanes = 1
            pipes = [pipe] + [pipe.replica(grabcut_lanes=1) for _ in range(n_pipes - 1)]
            streams = [torch.cuda.Stream(dev) for _ in range(n_pipes)]
            k2 = -(-args.steps // n_pipes) * n_pipes
            run_steps(n_pipes)                       # each replica allocates its arena
            sync_all()
            t1 = time.perf_counter()
            run_steps(k2)
            sync_all()
            e2 = time.perf_counter() - t1
            overlapped = {"pipelines": n_pipes, "grabcut_lanes": 1, "steps": k2, "value": round(batch_size * k2 / e2, 2),
                          "unit": "images/s", "ms_per_step": round(e2 / k2 * 1e3, 3),
                          "masks_identical_to_serial_run": bool(torch.equal(last["out"]["binary_mask"], serial_mask))}
            pipe.grabcut_lanes, n_pipes = keep_lanes, 1
            last["out"] = serial_out
        except Exception as exc:                     # never lose the contract's line over the extra pass
            overlapped = {"error": f"{type(exc).__name__}: {exc}"}
            n_pipes = 1

    images = batch_size * world * args.steps
    value = images / elapsed

    if rank == 0:
        if args.workload == "full":
            g = last["out"]["graphs"]
            n_nodes, n_edges = int(g.node_ptr_host[-1]), int(g.edge_ptr_host[-1])
        else:
            n_nodes, n_edges = batch.x.size(0), batch.edge_index.size(1)
        launches, agg_ms = prof["gcn_aggregate"]
        agg_avg_s = agg_ms / 1e3 / max(launches, 1)
        b_launch = agg_bytes(n_nodes, n_edges, HIDDEN, batch_size)
        achieved = b_launch / agg_avg_s / 1e9 if launches else 0.0
        roofline = {
            "kernel": "k_aggregate_graph<128,0,32,gated> (GCNConv scatter-gather, graph slice resident in LDS, fused gate/GELU/residual epilogue)",
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(args.workload, batch_size),
            "bytes_per_launch": b_launch, "avg_launch_us": round(agg_avg_s * 1e6, 2), "launches": launches,
            "event_pair_overhead_us": round(ctx.profile_query("#event_pair_overhead")[1] * 1e3, 2),
            # informational: bytes of all gathered neighbour rows per second; they are served from LDS
            "effective_gather_gbs": round((n_edges + n_nodes) * HIDDEN * 4 / agg_avg_s / 1e9, 1) if launches else 0.0,
        }
        stage_ms = {k: round(v[1] / args.steps, 3) for k, v in prof.items() if v[0]}

        cpu = parity = None
        if args.cpu_sample > 0 and world == 1:          # the CPU baseline is a rank-0, N=1 leg
            from oracle import oracle as orc       # checker / CPU baseline only
            sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items() if v.dtype.is_floating_point}
            n_s = min(args.cpu_sample, batch_size)
            if args.workload == "full":
                out = last["out"]
                seg_g, tri_g = out["segments"][:n_s].cpu().numpy(), out["trimap"][:n_s].cpu().numpy()
                bin_g, probs_g = out["binary_mask"][:n_s].cpu().numpy(), out["probs"].cpu().numpy()
                t1 = time.perf_counter()
                ref = [orc.segment(host_imgs[i], sd, HIDDEN, LAYERS, n_segments=N_SEGMENTS, seed=i) for i in range(n_s)]
                dt = time.perf_counter() - t1
                ious, dl, ei_ok = [], 0.0, []
                esrc, edst = g.edge_src.cpu().numpy().astype(np.int64), g.edge_dst.cpu().numpy().astype(np.int64)
                for i, r in enumerate(ref):
                    n0, n1 = int(g.node_ptr_host[i]), int(g.node_ptr_host[i + 1])
                    e0, e1 = int(g.edge_ptr_host[i]), int(g.edge_ptr_host[i + 1])
                    ei_ok.append(np.array_equal(np.stack([esrc[e0:e1], edst[e0:e1]]) - n0, r["graph"]["edge_index"]))
                    if n1 - n0 == r["probs"].shape[0]:
                        dl = max(dl, float(np.abs(probs_g[n0:n1] - r["probs"]).max()))
                    ious.append(orc.iou(bin_g[i], r["binary_mask"]) if r["binary_mask"].any() or bin_g[i].any() else 1.0)
                parity = {
                    "sample": n_s,
                    "label_map_exact_pct": round(100.0 * float(np.mean([np.array_equal(seg_g[i], ref[i]["segments"]) for i in range(n_s)])), 2),
                    "edge_index_exact_pct": round(100.0 * float(np.mean(ei_ok)), 2),
                    "trimap_pixel_match_pct": round(100.0 * float(np.mean([(tri_g[i] == ref[i]["trimap"]).mean() for i in range(n_s)])), 4),
                    "mask_exact_pct": round(100.0 * float(np.mean([np.array_equal(bin_g[i], ref[i]["binary_mask"]) for i in range(n_s)])), 2),
                    "max_abs_dprob": dl, "mean_mask_iou": round(float(np.mean(ious)), 6), "min_mask_iou": round(float(np.min(ious)), 6),
                }
                what = f"{n_s} of the {batch_size} images, full pipeline"
            else:
                probs_g = last["out"].cpu().numpy()
                off = np.cumsum([0] + [x.shape[0] for x, _, _ in host_graphs])
                orc.resgcn_forward(sd, HIDDEN, LAYERS, *host_graphs[0])          # warm
                t1 = time.perf_counter()
                ref = [orc.resgcn_forward(sd, HIDDEN, LAYERS, x, ei, ea) for x, ei, ea in host_graphs[:n_s]]
                dt = time.perf_counter() - t1
                parity = {"sample": n_s, "max_abs_dprob": max(float(np.abs(probs_g[off[i]:off[i + 1]] - ref[i][1]).max())
                                                              for i in range(n_s))}
                what = f"{n_s} of the {batch_size} graphs, GCN forward only"
            cpu = {"value": round(n_s / dt, 3), "unit": "images/s", "cores": 1, "kind": "port",
                   "sample": f"{what}; C oracle, 1 thread; host has {len(os.sched_getaffinity(0))} cores"}

        cfg_name = ("configs[2]: full pipeline (SLIC->graph->ResGCNNet->guided-filter trimap->GrabCut 5 it->clean-up), "
                    f"batch {batch_size} of {W}x{H}, n_segments={N_SEGMENTS}") if args.workload == "full" else \
                   (f"configs[1]: batch {batch_size} DUTS-shape region graphs (~600 superpixels), "
                    "ResGCNNet(D=128,n=6) forward only")
        out_json = {
            "metric": "images/sec end-to-end mask (DUTS-shape batch)", "value": round(value, 2),
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg_name, "images_per_gpu": batch_size, "nodes": n_nodes,
                       "directed_edges": n_edges, "weights": "seeded random init (torch.manual_seed(0))",
                       "pipelines": n_pipes, "grabcut_lanes": lanes},
            "roofline": roofline, "cpu_baseline": cpu, "parity_vs_cpu_oracle": parity,
            "stage_ms_per_step": stage_ms, "overlapped": overlapped,
        }
        print(json.dumps(out_json), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
