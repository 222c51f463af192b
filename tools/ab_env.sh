#!/bin/bash
# ab_env.sh "ENV=.. ENV=.." ...: GrabCut stage time (batch 256, 4 lanes) under each environment setting ("-" = none)
R=$GRAFT_REPO_ROOT
for e in "$@"; do
  [ "$e" = "-" ] && e=""
  echo "== ${e:-default}"
  env $e LANES=${LANES:-4} REPS=${REPS:-4} timeout -k 10 200 python3 $R/tools/mf_time.py 2>&1 | tail -1
done
