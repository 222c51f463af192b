#!/bin/bash
# experiment helper: rocprofv3 kernel stats of the default bench (single GrabCut lane so kernel times are unperturbed)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/kstats
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kstats -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --overlap-pass 0 --lanes ${LANES:-1} > $R/gpurun_out/kstats.log 2>&1
python3 - <<PY
import csv, glob
f=sorted(glob.glob("$R/gpurun_out/kstats/*/*kernel_stats.csv"), key=lambda p: -__import__("os").path.getsize(p))[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step", round(tot/1e6/4,2))
for r in rows[:int("${TOP:-30}")]:
    print(f'{float(r["TotalDurationNs"])/4e6:8.2f} ms/step {int(r["Calls"])//4:6d} calls/step {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:70]}')
PY
