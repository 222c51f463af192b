#!/bin/bash
# PMC passes (HBM bytes of the graded kernel) on the default bench workload: separate passes, kernel-trace + pmc only
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmcb_*
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcb_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 --overlap-pass 0 > $R/gpurun_out/pmcb_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmcb_write -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 --overlap-pass 0 > $R/gpurun_out/pmcb_write.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcb_summary.json $R/gpurun_out/pmcb_fetch $R/gpurun_out/pmcb_write
