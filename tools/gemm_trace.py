"""Per-wave phase timeline of k_gemm<128, 0> from the stamps of the GEMM_TRACE experiment build
(tools/build_variant.sh trace ggc_resgcn -DGEMM_TRACE; bench.py --workload gcn with GGC_HIP_LIBRARY / GGC_GEMM_TRACE set)."""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64)
launches, i = [], 0
while i < len(raw):
    n = int(raw[i]); launches.append(raw[i + 1:i + 1 + n * 8].reshape(n, 8)); i += 1 + n * 8
t = launches[-1]
st = t[:, :7].astype(np.int64)
names = ["stage W + barrier", "row loads + sum", "LayerNorm", "MFMA loop", "stores issued", "stores done"]
d = np.diff(st, axis=1)
print(len(launches), "launches; waves in the last one:", len(t))
for k, nm in enumerate(names):
    print(f"{nm:20s} mean {d[:, k].mean():9.0f}  p10 {np.percentile(d[:, k], 10):9.0f}  p90 {np.percentile(d[:, k], 90):9.0f}")
print("wave lifetime mean", round((st[:, 6] - st[:, 0]).mean()))
hw = t[:, 7]
xcc = ((hw >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64)
key = xcc * 100000 + ((hw & np.uint64(0xFFFF)) >> np.uint64(4)).astype(np.int64)      # xcc | se, sh, cu, pipe, simd
u, cnt = np.unique(key, return_counts=True)
print("SIMD slots", len(u), " waves per slot min/mean/max", cnt.min(), round(cnt.mean(), 2), cnt.max())
span = mf_sum = mf_union = both = 0
for k in u:
    rows = st[key == k]
    span += rows[:, 6].max() - rows[:, 0].min()                       # one XCD's clock per slot: comparable
    iv = sorted((int(a), int(b)) for a, b in rows[:, [3, 4]])
    mf_sum += sum(b - a for a, b in iv)
    ca, cb = iv[0]; un = 0
    for a, b in iv[1:]:
        if a > cb: un += cb - ca; ca, cb = a, b
        else: cb = max(cb, b)
    mf_union += un + cb - ca
    lives = sorted((int(a), int(b)) for a, b in rows[:, [0, 6]])
    ev = sorted([(a, 1) for a, _ in lives] + [(b, -1) for _, b in lives]); lvl = 0; last = ev[0][0]
    for x, dlt in ev:
        if lvl >= 2: both += x - last
        lvl += dlt; last = x
print("per SIMD slot: span", round(span / len(u)), " MFMA-loop intervals sum", round(mf_sum / len(u)), " union", round(mf_union / len(u)),
      " time with two waves resident", round(both / len(u)))
