#!/bin/bash
# ab.sh VARIANT...: GrabCut stage time (batch 256, 4 lanes) with the default library and each libggc_hip_VARIANT.so
R=$GRAFT_REPO_ROOT
for v in "" "$@"; do
  if [ -z "$v" ]; then unset GGC_HIP_LIBRARY; else export GGC_HIP_LIBRARY=$R/gcn-grabcut_amd/libggc_hip_$v.so; fi
  echo "== ${v:-default}"
  LANES=${LANES:-4} REPS=${REPS:-4} timeout -k 10 200 python3 $R/tools/mf_time.py 2>&1 | tail -1
done
