#!/bin/bash
# experiment helper: late-round schedule with 32x32 push tiles (full-pipeline bench)
for cfg in ${CFGS:-"0 24 32" "1 24 32" "1 16 32" "1 32 32" "1 16 64" "1 12 32"}; do
  set -- $cfg
  GGC_MF_TALL=$1 GGC_MF_TALL_LAUNCHES=$2 GGC_MF_TALL_INNER=$3 python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -1 > /tmp/mf.json
  python - "$cfg" <<'PY'
import json,sys
d=json.load(open("/tmp/mf.json")); s=d["stage_ms_per_step"]
print("tall launches inner",sys.argv[1],"img/s",d["value"],"ms_per_step",d["ms_per_step"],"relabel",s["maxflow_relabel"],"push",s["maxflow_push"])
PY
done
