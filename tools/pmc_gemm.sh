#!/bin/bash
# PMC passes for the f32-MFMA node-feature GEMM k_gemm<128,0> on configs[1] at batch 256 (kernel-trace + pmc only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmcm_*
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmcm_$name -- python3 $R/bench.py --workload gcn --batch 256 --steps 3 --warmup 1 --cpu-sample 0 > $R/gpurun_out/pmcm_$name.log 2>&1; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD
run b SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run c SQ_WAVES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcm_summary.json $R/gpurun_out/pmcm_a $R/gpurun_out/pmcm_b $R/gpurun_out/pmcm_c
rm -rf $R/gpurun_out/pmcm_a $R/gpurun_out/pmcm_b $R/gpurun_out/pmcm_c
