#!/bin/bash
# kernel trace (timestamps) of a short default bench run: per-queue timeline for tools/trace_lanes.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/ktrace
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ktrace -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 > $R/gpurun_out/ktrace.log 2>&1
python3 $R/tools/trace_lanes.py $R/gpurun_out/ktrace > $R/gpurun_out/trace_lanes.txt 2>&1
rm -rf $R/gpurun_out/ktrace
