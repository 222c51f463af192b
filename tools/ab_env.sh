#!/bin/bash
# ab_env.sh "ENV=.. ENV=.." ...: GrabCut stage time (tools/mf_time.py; batch MF_BATCH=256 on LANES=4 lanes) under each
# environment setting ("-" = none).  One process per setting: the library reads its GGC_* switches (include/ggc.h) once.
# This one parametrised helper replaces the per-experiment sweep scripts of round 2 (mf*_sweep.sh, ab.sh, ab_lanes.sh), e.g.
#   tools/ab_env.sh - "GGC_MF_ASYNC_PUSH_ACTIVE=30000" "GGC_MF_DENSE_LAUNCHES=8" "LANES=6 GPU_MAX_HW_QUEUES=8"
#   GGC_HIP_LIBRARY=$R/gcn-grabcut_amd/libggc_hip_VARIANT.so tools/ab_env.sh -       (a library built by tools/build_variant.sh)
R=$GRAFT_REPO_ROOT
for e in "$@"; do
  [ "$e" = "-" ] && e=""
  echo "== ${e:-default}"
  env LANES=${LANES:-4} REPS=${REPS:-4} $e timeout -k 10 200 python3 $R/tools/mf_time.py 2>&1 | tail -1
done
