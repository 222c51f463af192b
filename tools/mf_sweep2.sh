#!/bin/bash
# experiment helper: default bench under one max-flow knob at a time (work-list grids, schedule, lanes).
# Usage on the GPU box: bash tools/mf_sweep2.sh > gpurun_out/mf_sweep2.log
R=$GRAFT_REPO_ROOT
run() { echo "$* -> $(env "$@" timeout -k 10 200 python $R/bench.py --cpu-sample 0 2>/dev/null | tail -1 | cut -c60-78)"; }
for g in ${PUSH_GRIDS:-512 1024 2048}; do run GGC_MF_PUSH_GRID=$g; done
for g in ${RELAX_GRIDS:-1024 2048 4096}; do run GGC_MF_RELAX_GRID=$g; done
run GGC_MF_PR_LAUNCHES0=8
run GGC_MF_PR_LAUNCHES=16
run GGC_MF_TAIL_LAUNCHES=48
run GGC_MF_PPT=2
for l in ${LANES:-1 2 4}; do echo "lanes=$l -> $(timeout -k 10 200 python $R/bench.py --cpu-sample 0 --lanes $l 2>/dev/null | tail -1 | cut -c60-78)"; done
