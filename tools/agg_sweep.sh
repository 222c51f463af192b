#!/bin/bash
# experiment helper: sweep the aggregate kernel's block size (GGC_AGG_THREADS) on the GCN-only workload
for t in 64 128 256 512 1024; do
  GGC_AGG_THREADS=$t python bench.py --workload gcn --batch ${1:-256} --steps 30 --warmup 5 --cpu-sample 0 2>&1 | tail -1 > /tmp/agg_$t.json
  python - "$t" <<'PY'
import json,sys
t=sys.argv[1]; d=json.load(open(f"/tmp/agg_{t}.json"))
print("threads",t,"ms_per_step",d["ms_per_step"],"agg_us",d["roofline"]["avg_launch_us"],"frac",d["roofline"]["frac"])
PY
done
