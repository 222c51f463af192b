#!/bin/bash
# ab_nt.sh VARIANT ...: the graded kernel's launch time (bench.py HIP events, configs[2]) for the default library and for
# libraries built by tools/build_variant.sh, alternating, REPS (3) times on one box in one session.
R=$GRAFT_REPO_ROOT
for rep in $(seq ${REPS:-3}); do
for v in default "$@"; do
  if [ $v = default ]; then unset GGC_HIP_LIBRARY; else export GGC_HIP_LIBRARY=$R/gcn-grabcut_amd/libggc_hip_$v.so; fi
  python3 $R/bench.py --cpu-sample 0 --h2d-steps 0 --overlap-pass 0 --steps 10 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', 'full:', d['roofline']['avg_launch_us'], 'us', d['roofline']['frac'], 'step', d['ms_per_step'])"
done; done
