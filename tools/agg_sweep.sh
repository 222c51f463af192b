#!/bin/bash
# experiment helper: time the GCN aggregation kernel variants (gcn workload)
for v in ${XPS:-0 1 2 3 4 5}; do
  GGC_AGG_XP=$v python bench.py --workload gcn --batch ${1:-256} --steps 5 --warmup 2 --cpu-sample 0 2>&1 | tail -1 > /tmp/agg.json
  python - "$v" <<'PY'
import json,sys
d=json.load(open("/tmp/agg.json"))
print("xp",sys.argv[1],"agg_us",d["roofline"]["avg_launch_us"],"frac",d["roofline"]["frac"],"ms_per_step",d["ms_per_step"],d["stage_ms_per_step"])
PY
done
