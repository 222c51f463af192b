#!/bin/bash
# experiment helper: hardware queues x GrabCut lanes (full-pipeline bench)
for cfg in ${CFGS:-"16 6" "16 8" "2 4" "1 4"}; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$1 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --lanes $2 2>&1 | tail -1 > /tmp/mf.json
  python - "$cfg" <<'PY'
import json,sys
d=json.load(open("/tmp/mf.json"))
print("hwq lanes",sys.argv[1],"img/s",d["value"],"ms_per_step",d["ms_per_step"])
PY
done
