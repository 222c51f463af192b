#!/bin/bash
# round-3 evidence, one box, one session: the default bench line, rocprofv3 kernel stats of the same command, the PMC passes
# (HBM bytes of every kernel), the per-lane timeline, the overlapped / software-pipelined figures of this build.
# Only the summaries stay under gpurun_out/ (the raw traces exceed what gpurun copies back); copy them to profiles/.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
python3 $R/bench.py > $O/r03_bench_line.json 2> $O/r03_bench.err
echo "bench done"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/r03_kstats /tmp/pmcb_fetch /tmp/pmcb_write
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r03_kstats -- python3 $R/bench.py > $O/r03_kstats.log 2>&1
cp "$(ls -S /tmp/r03_kstats/*/*kernel_stats.csv | head -1)" $O/r03_bench_default_kernel_stats.csv
echo "kernel stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmcb_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 --cpu-all 0 --h2d-steps 0 > $O/pmcb_fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmcb_write -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 --cpu-all 0 --h2d-steps 0 > $O/pmcb_write.log 2>&1
python3 $R/tools/pmc_summary.py $O/r03_pmc_bench_hbm.json /tmp/pmcb_fetch /tmp/pmcb_write
echo "pmc done"
# per-lane timeline of the last step of a short run (kernels per lane, busy time, gaps, kernels in flight)
bash $R/tools/trace_lanes.sh
cp $O/trace_lanes.txt $O/r03_trace_lanes.txt
echo "trace done"
# consecutive batches overlapped on three pipeline replicas (informational, not `value`)
python3 $R/bench.py --cpu-sample 0 --h2d-steps 0 --overlap-pass 3 > $O/r03_bench_overlap3.json 2>> $O/r03_bench.err
# the software pipeline inside one step (segment_batch_device(chunks=...)) against the one-chunk step
PLANS=4:1.0,4:0.7,3:0.7,2:0.8 python3 $R/tools/pipe_sweep.py > $O/r03_pipe_sweep.txt 2>&1
echo "all done"
