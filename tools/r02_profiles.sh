#!/bin/bash
# round-2 evidence: the default bench line, rocprofv3 kernel stats of the same command, then the PMC passes.
# Only the summaries stay under gpurun_out/ (the raw traces exceed what gpurun copies back).
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
python3 $R/bench.py > $O/r02_bench.json 2> $O/r02_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/r02_kstats /tmp/pmcb_fetch /tmp/pmcb_write
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r02_kstats -- python3 $R/bench.py > $O/r02_kstats.log 2>&1
cp "$(ls -S /tmp/r02_kstats/*/*kernel_stats.csv | head -1)" $O/r02_bench_default_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmcb_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 --cpu-all 0 --h2d-steps 0 > $O/pmcb_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmcb_write -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 --cpu-all 0 --h2d-steps 0 > $O/pmcb_write.log 2>&1
python3 $R/tools/pmc_summary.py $O/r02_pmc_bench_hbm.json /tmp/pmcb_fetch /tmp/pmcb_write
# per-lane timeline of the last step of a short run (kernels per lane, busy time, gaps, kernels in flight)
bash $R/tools/trace_lanes.sh
cp $O/trace_lanes.txt $O/r02_trace_lanes.txt
