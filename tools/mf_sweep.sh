#!/bin/bash
# experiment helper: sweep push-relabel schedule (first-round launches, max launches, inner sweeps)
for cfg in "4 24 8" "2 24 8" "1 24 8" "2 32 8" "2 16 16" "4 16 16" "1 16 16" "2 48 8"; do
  set -- $cfg
  GGC_MF_PR_FIRST=$1 GGC_MF_PR_LAUNCHES=$2 GGC_MF_PR_INNER=$3 python bench.py --steps 2 --warmup 1 --cpu-sample 0 2>&1 | tail -1 > /tmp/mf.json
  python - "$1" "$2" "$3" <<'PY'
import json,sys
d=json.load(open("/tmp/mf.json")); s=d["stage_ms_per_step"]
print("first",sys.argv[1],"launches",sys.argv[2],"inner",sys.argv[3],"ms_per_step",d["ms_per_step"],"relabel",s["maxflow_relabel"],"push",s["maxflow_push"])
PY
done
