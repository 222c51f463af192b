#!/bin/bash
# experiment helper: sweep the work-list grid size of the max-flow kernels on the full-pipeline bench
for g in 16384 32768 131072; do
  GGC_MF_LIST_GRID=$g python bench.py --steps 2 --warmup 1 --cpu-sample 0 2>&1 | tail -1 > /tmp/mf.json
  python - "$g" <<'PY'
import json,sys
d=json.load(open("/tmp/mf.json")); s=d["stage_ms_per_step"]
print("grid",sys.argv[1],"ms_per_step",d["ms_per_step"],"relabel",s["maxflow_relabel"],"push",s["maxflow_push"])
PY
done
