"""Per-queue view of a rocprofv3 kernel trace of bench.py: for the LAST step, per queue: kernels, busy time, span;
   per kernel name: launches, total and mean duration; chip-wide: union busy time and mean concurrency."""
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
ev.sort()
# steps: find k_preprocess launches (one per step)
pre = [e[0] for e in ev if "k_preprocess" in e[2]]
import os
t0 = pre[-int(os.environ.get("NPRE", "1"))]      # NPRE: k_preprocess launches per step (chunks of the software pipeline)
t1 = ev[-1][1]
step = [e for e in ev if e[0] >= t0]
print(f"last step: {len(step)} kernels, span {(t1 - t0) / 1e6:.2f} ms")
byq = collections.defaultdict(list)
for e in step: byq[(e[3], e[4])].append(e)
for q, l in sorted(byq.items()):
    busy = sum(e[1] - e[0] for e in l)
    print(f"queue/stream {q}: {len(l):5d} kernels, busy {busy / 1e6:7.2f} ms, span {(l[-1][1] - l[0][0]) / 1e6:7.2f} ms, first at +{(l[0][0] - t0) / 1e6:.2f} ms")
# busy fraction per queue in 5-ms bins
nb = int((t1 - t0) / 5e6) + 1
for q, l in sorted(byq.items()):
    bins = [0.0] * nb
    for e in l:
        a, b_ = e[0] - t0, e[1] - t0
        for k in range(int(a / 5e6), min(nb - 1, int(b_ / 5e6)) + 1):
            bins[k] += max(0.0, min(b_, (k + 1) * 5e6) - max(a, k * 5e6))
    print(f"  busy % per 5 ms, queue {q}: " + " ".join(f"{100 * v / 5e6:3.0f}" for v in bins))
# union busy + concurrency
pts = []
for e in step: pts.append((e[0], 1)); pts.append((e[1], -1))
pts.sort()
cur = 0; last = pts[0][0]; hist = collections.Counter()
for t, dlt in pts:
    hist[cur] += t - last; last = t; cur += dlt
tot = sum(hist.values())
print("time by number of kernels in flight:", {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
byn = collections.defaultdict(lambda: [0, 0])
for e in step:
    n = e[2].replace("(anonymous namespace)::", "").split("(")[0][-60:]
    byn[n][0] += 1; byn[n][1] += e[1] - e[0]
print("kernel, launches, total ms, mean us")
for n, (c, t) in sorted(byn.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"  {n:60s} {c:6d} {t / 1e6:8.2f} {t / c / 1e3:8.1f}")
# the GrabCut lanes: per queue, time split between max-flow kernels, other kernels and gaps
for q, l in sorted(byq.items()):
    if len(l) < 200: continue
    mf = sum(e[1] - e[0] for e in l if "k_mf" in e[2] or "k_aq" in e[2] or "done_update" in e[2] or "open_" in e[2])
    oth = sum(e[1] - e[0] for e in l) - mf
    gaps = 0; big = 0
    for a, b in zip(l, l[1:]):
        g = b[0] - a[1]
        if g > 0: gaps += g
        if g > 20000: big += 1
    print(f"lane {q}: max-flow kernels {mf / 1e6:.2f} ms, other kernels {oth / 1e6:.2f} ms, gaps {gaps / 1e6:.2f} ms ({big} gaps > 20 us)")
