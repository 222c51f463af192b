#!/bin/bash
# rocprofv3 kernel stats of the GrabCut stage alone (tools/mf_time.py): kstats_mf.sh TAG   (env LANES, MF_BATCH, REPS)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-a}
rm -rf $R/gpurun_out/kmf_$T
REPS=${REPS:-4} rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kmf_$T -- python3 $R/tools/mf_time.py > $R/gpurun_out/kmf_$T.log 2>&1
python3 - <<PY > $R/gpurun_out/kmf_$T.txt
import csv, glob, os
f=sorted(glob.glob("$R/gpurun_out/kmf_$T/*/*kernel_stats.csv"), key=lambda p: -os.path.getsize(p))[0]
rows=list(csv.DictReader(open(f)))
print(open("$R/gpurun_out/kmf_$T.log").read().strip().splitlines()[-1])
for r in rows[:40]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {int(r["Calls"]):6d} calls {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:60]}')
PY
rm -rf $R/gpurun_out/kmf_$T
