#!/bin/bash
# experiment helper: number of concurrent GrabCut lanes on the full-pipeline bench
for cfg in ${CFGS:-"4 3" "4 4" "8 4" "8 6" "8 8"}; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$1 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --lanes $2 2>&1 | tail -1 > /tmp/mf.json
  python - "$cfg" <<'PY'
import json,sys
d=json.load(open("/tmp/mf.json")); s=d["stage_ms_per_step"]
print("hwq lanes",sys.argv[1],"img/s",d["value"],"ms_per_step",d["ms_per_step"],"agg_us",d["roofline"]["avg_launch_us"])
PY
done
