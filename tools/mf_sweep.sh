#!/bin/bash
# experiment helper: push schedule (launches per round x inner sweeps per tile visit) on the full-pipeline bench
for cfg in ${CFGS:-"24 8" "16 16" "12 16" "12 32" "8 32" "24 16"}; do
  set -- $cfg
  GGC_MF_PR_LAUNCHES=$1 GGC_MF_PR_INNER=$2 python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -1 > /tmp/mf.json
  python - "$cfg" <<'PY'
import json,sys
d=json.load(open("/tmp/mf.json")); s=d["stage_ms_per_step"]
print("launches inner",sys.argv[1],"img/s",d["value"],"ms_per_step",d["ms_per_step"],"relabel",s["maxflow_relabel"],"push",s["maxflow_push"])
PY
done
