#!/bin/bash
# ab_lanes.sh LANES...: GrabCut stage time per lane count (GPU_MAX_HW_QUEUES from the environment)
R=$GRAFT_REPO_ROOT
for l in "$@"; do
  LANES=$l REPS=${REPS:-4} timeout -k 10 200 python3 $R/tools/mf_time.py 2>&1 | tail -1
done
