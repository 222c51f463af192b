#!/bin/bash
R=$GRAFT_REPO_ROOT
for l in 1 2 4; do echo "lanes=$l -> $(timeout -k 10 200 python $R/bench.py --cpu-sample 0 --lanes $l 2>/dev/null | tail -1 | cut -c60-78)"; done
