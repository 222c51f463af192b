#!/bin/bash
# experiment helper: time the GCN aggregation kernel under debug switches
# GGC_AGG_DBG bits: 1 no gather, 2 no tile fill, 4 gate/h from one row, 8 no store, 16 no CSR loads
for v in ${DBGS:-0 31}; do
  GGC_AGG_DBG=$v python bench.py --workload gcn --batch ${1:-256} --steps 5 --warmup 2 --cpu-sample 0 2>&1 | tail -1 > /tmp/agg.json
  python - "$v" <<'PY'
import json,sys
d=json.load(open("/tmp/agg.json"))
print("dbg",sys.argv[1],"agg_us",d["roofline"]["avg_launch_us"],"ms_per_step",d["ms_per_step"],d["stage_ms_per_step"])
PY
done
GGC_AGG_DIRECT=1 python bench.py --workload gcn --batch ${1:-256} --steps 5 --warmup 2 --cpu-sample 0 2>&1 | tail -1 | cut -c 1000-1400
