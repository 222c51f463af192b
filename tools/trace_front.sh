#!/bin/bash
# kernel sequence of the FRONT stages (colour prep .. seeding) of one batch-256 step, in launch order -> gpurun_out/front_TAG.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-a}
rm -rf $R/gpurun_out/kfr_$T
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kfr_$T -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --h2d-steps 0 > $R/gpurun_out/kfr_$T.log 2>&1
python3 - <<PY > $R/gpurun_out/front_$T.txt
import csv, glob
f=glob.glob("$R/gpurun_out/kfr_$T/**/*kernel_trace.csv", recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
pre=[i for i,r in enumerate(rows) if "k_preprocess" in r["Kernel_Name"]]
start=pre[-1]
t0=int(rows[start]["Start_Timestamp"]); prev=t0
for r in rows[start:]:
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","").replace("ggc::","")
    if "k_gc_flags" in n: break
    s_,e_=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print(f'{(s_-t0)/1e3:10.1f} us  gap {max(0,s_-prev)/1e3:7.1f}  {(e_-s_)/1e3:8.1f} us  grid {int(r["Grid_Size_X"])//max(int(r["Workgroup_Size_X"]),1):6d} x{r["Grid_Size_Y"]:>4s} x{r["Grid_Size_Z"]:>4s}  {n[:60]}')
    prev=e_
PY
rm -rf $R/gpurun_out/kfr_$T
