#!/bin/bash
# experiment helper: default bench under one knob at a time.  MODE="--pipelines 3" sweeps the overlapped mode.
# Usage on the GPU box: MODE="--pipelines 3" bash tools/mf_sweep2.sh > gpurun_out/mf_sweep2.log
R=$GRAFT_REPO_ROOT
M=${MODE:-}
run() { echo "$M $* -> $(env "$@" timeout -k 10 200 python $R/bench.py --cpu-sample 0 --steps ${STEPS:-12} $M 2>/dev/null | tail -1 | cut -c60-78)"; }
for g in ${PUSH_GRIDS:-512 1024 2048}; do run GGC_MF_PUSH_GRID=$g; done
for g in ${RELAX_GRIDS:-1024 2048 4096}; do run GGC_MF_RELAX_GRID=$g; done
for k in ${EXTRA:-GGC_MF_PR_LAUNCHES0=8 GGC_MF_PR_LAUNCHES=16 GGC_MF_TAIL_LAUNCHES=48}; do run $k; done
