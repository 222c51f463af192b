#!/bin/bash
# kernel sequence of the GrabCut stage of ONE lane (tools/mf_time.py, LANES=1): name, duration, grid of every max-flow launch
# of the last repetition, in launch order -> gpurun_out/seq_TAG.txt   (env MF_BATCH)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-a}
rm -rf $R/gpurun_out/kseq_$T
LANES=1 REPS=1 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kseq_$T -- python3 $R/tools/mf_time.py > $R/gpurun_out/kseq_$T.log 2>&1
python3 - <<PY > $R/gpurun_out/seq_$T.txt
import csv, glob
f=glob.glob("$R/gpurun_out/kseq_$T/**/*kernel_trace.csv", recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last GrabCut: from the last k_gc_flags pair backwards
idx=[i for i,r in enumerate(rows) if "k_km_chunks" in r["Kernel_Name"]]
start=idx[-5] if len(idx)>=5 else 0
t0=int(rows[start]["Start_Timestamp"])
for r in rows[start:]:
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","").replace("ggc::","")
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:10.1f} us  {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} us  grid {int(r["Grid_Size_X"])//max(int(r["Workgroup_Size_X"]),1):6d} x{r["Grid_Size_Y"]:>4s}  {n[:40]}')
PY
rm -rf $R/gpurun_out/kseq_$T
