#!/bin/bash
# experiment helper: relabel launches per host read-back, and lanes (full-pipeline bench)
for cfg in ${CFGS:-"4 4" "6 4" "8 4" "12 4" "8 6" "12 8"}; do
  set -- $cfg
  GGC_MF_RELAX_REP=$1 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --lanes $2 2>&1 | tail -1 > /tmp/mf.json
  python - "$cfg" <<'PY'
import json,sys
d=json.load(open("/tmp/mf.json")); s=d["stage_ms_per_step"]
print("relax_rep lanes",sys.argv[1],"img/s",d["value"],"ms_per_step",d["ms_per_step"])
PY
done
