#!/bin/bash
# schedule sweep of the per-image max-flow (one process per setting: the library reads its env once)
out=gpurun_out/mfi_sweep.log; : > $out
run() { env "$@" timeout -k 10 120 python tools/mf_time.py 2>&1 | grep "grabcut stage" >> $out; }
run LANES=1
run LANES=4
run LANES=4 GGC_MFI_PASSES0=6 GGC_MFI_PASSES=8
run LANES=4 GGC_MFI_PASSES0=4 GGC_MFI_PASSES=4
run LANES=4 GGC_MFI_PASSES0=8 GGC_MFI_PASSES=12 GGC_MFI_INNER=16
run LANES=4 GGC_MFI_PASSES0=4 GGC_MFI_PASSES=6 GGC_MFI_INNER=16
run LANES=4 GGC_MFI_TAIL_ACTIVE=1024
run LANES=4 GGC_MFI_TAIL_ACTIVE=64
run LANES=4 GGC_MFI_TAIL_PASSES=8 GGC_MFI_TAIL_INNER=64
run LANES=4 GGC_MFI_TAIL_PASSES=32 GGC_MFI_TAIL_INNER=16
cat $out
