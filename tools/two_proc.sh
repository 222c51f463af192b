#!/bin/bash
# experiment helper: NPROC bench processes on one GPU at once (does another pipeline fill the GPU?).  LANES per process.
R=$GRAFT_REPO_ROOT
L=${LANES:-1}; N=${NPROC:-2}
pids=()
for i in $(seq 1 $N); do
  timeout -k 10 300 python $R/bench.py --cpu-sample 0 --lanes $L --steps ${STEPS:-60} --warmup 5 > $R/gpurun_out/multi_$i.log 2>&1 &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
tot=0
for i in $(seq 1 $N); do v=$(tail -1 $R/gpurun_out/multi_$i.log | sed 's/.*"value": \([0-9.]*\),.*/\1/'); echo "  proc $i: $v"; tot=$(python3 -c "print($tot+$v)"); done
echo "NPROC=$N LANES=$L aggregate $tot"
