R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in default nt0 nt4 nt3; do
  if [ $v = default ]; then unset GGC_HIP_LIBRARY; else export GGC_HIP_LIBRARY=$R/gcn-grabcut_amd/libggc_hip_$v.so; fi
  python3 $R/bench.py --cpu-sample 0 --h2d-steps 0 --steps 10 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', 'full:', d['roofline']['avg_launch_us'], 'us', d['roofline']['frac'], 'step', d['ms_per_step'])"
done; done
