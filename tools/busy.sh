#!/bin/bash
# experiment helper: GPU busy time (union of kernel intervals) vs wall time of the timed steps, default bench (4 lanes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/busy
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/busy -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --overlap-pass 0 > $R/gpurun_out/busy.log 2>&1
python3 - <<PY
import csv, glob, os, json
f=sorted(glob.glob("$R/gpurun_out/busy/*/*kernel_trace.csv"), key=lambda p: -os.path.getsize(p))[0]
iv=[]
for r in csv.DictReader(open(f)):
    iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
iv.sort()
# timed region = last 3/4 of the span roughly: find SLIC minmax kernels as step starts
starts=[a for a,b,n in iv if "k_preprocess" in n]
print("steps found", len(starts))
t0=starts[1]; t1=max(b for a,b,n in iv)
sel=[(a,b) for a,b,n in iv if a>=t0]
busy=0; cur_s,cur_e=sel[0]
for a,b in sel[1:]:
    if a<=cur_e: cur_e=max(cur_e,b)
    else: busy+=cur_e-cur_s; cur_s,cur_e=a,b
busy+=cur_e-cur_s
tot=sum(b-a for a,b in sel)
print(f"wall {(t1-t0)/1e6:.1f} ms, busy(union) {busy/1e6:.1f} ms = {busy/(t1-t0)*100:.1f} %, sum of kernel durations {tot/1e6:.1f} ms, mean concurrency {tot/busy:.2f}")
PY
tail -1 $R/gpurun_out/busy.log | cut -c1-160
