#!/bin/bash
# experiment helper: time the GCN aggregation kernel on both bench workloads
python bench.py --workload gcn --batch ${1:-256} --steps 5 --warmup 2 --cpu-sample 0 2>&1 | tail -1 > /tmp/agg.json
python - <<'PY'
import json
d=json.load(open("/tmp/agg.json"))
print("gcn workload: agg_us",d["roofline"]["avg_launch_us"],"frac",d["roofline"]["frac"],"ms_per_step",d["ms_per_step"],d["stage_ms_per_step"])
PY
python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -1 > /tmp/agg.json
python - <<'PY'
import json
d=json.load(open("/tmp/agg.json"))
print("full workload: img/s",d["value"],"agg_us",d["roofline"]["avg_launch_us"],"frac",d["roofline"]["frac"])
PY
