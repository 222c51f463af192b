#!/bin/bash
# experiment helper: push launches in the first round of every max-flow (full-pipeline bench)
for cfg in ${CFGS:-"24 24" "8 24" "12 24" "16 24" "8 16" "12 32"}; do
  set -- $cfg
  GGC_MF_PR_LAUNCHES0=$1 GGC_MF_PR_LAUNCHES=$2 python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -1 > /tmp/mf.json
  python - "$cfg" <<'PY'
import json,sys
d=json.load(open("/tmp/mf.json")); s=d["stage_ms_per_step"]
print("launches0 launches",sys.argv[1],"img/s",d["value"],"ms_per_step",d["ms_per_step"],"relabel",s["maxflow_relabel"],"push",s["maxflow_push"])
PY
done
