#!/bin/bash
# PMC passes for the GCN aggregation kernel (separate passes; kernel-trace + pmc only — no sys/hip traces)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=${1:-256}
run() { # name counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$name -- python3 $R/bench.py --workload gcn --batch $B --steps 3 --warmup 1 --cpu-sample 0 > $R/gpurun_out/pmc_$name.log 2>&1
}
rm -rf $R/gpurun_out/pmc_*
for p in ${PASSES:-fetch write sq sq2 lds}; do
  case $p in
    fetch) run fetch FETCH_SIZE ;;
    write) run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum ;;
    sq)    run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS ;;
    sq2)   run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC ;;
    lds)   run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR SQ_WAVES ;;
    tcp)   run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr ;;
  esac
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_summary.json $R/gpurun_out/pmc_*/
echo done
