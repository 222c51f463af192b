#!/bin/bash
# kernel trace of the software-pipelined step: trace_pipe.sh TAG PLAN(e.g. 4:1.0)   -> gpurun_out/trace_pipe_TAG.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-a}; P=${2:-4:1.0}
rm -rf $R/gpurun_out/ktp_$T
PLANS=$P STEPS=2 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ktp_$T -- python3 $R/tools/pipe_sweep.py > $R/gpurun_out/ktp_$T.log 2>&1
NPRE=${P%%:*} python3 $R/tools/trace_lanes.py $R/gpurun_out/ktp_$T > $R/gpurun_out/trace_pipe_$T.txt 2>&1
rm -rf $R/gpurun_out/ktp_$T
