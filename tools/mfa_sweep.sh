#!/bin/bash
# async max-flow knobs (ggc_maxflow_async.hip): GrabCut stage time per setting
for a in "GGC_MF_ASYNC_RELAX=0 GGC_MF_ASYNC_PUSH_ACTIVE=0" "$@"; do
  env $a LANES=${LANES:-4} timeout -k 10 120 python tools/mf_time.py 2>&1 | tail -1
done
