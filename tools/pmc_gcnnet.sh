#!/bin/bash
# PMC passes for the fused edge-gate kernel of GCNTrimapNet (kernel-trace + pmc only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmcg_*
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmcg_$name -- python3 $R/tools/gcnnet_rate.py > $R/gpurun_out/pmcg_$name.log 2>&1; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD
run b SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcg_summary.json $R/gpurun_out/pmcg_a $R/gpurun_out/pmcg_b
