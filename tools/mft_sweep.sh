#!/bin/bash
out=gpurun_out/mft_sweep.log; : > $out
run() { env "$@" timeout -k 10 120 python tools/mf_time.py 2>&1 | grep "grabcut stage\|Error" >> $out; }
run LANES=4 GGC_MF_IMAGE_TAIL=2
run LANES=4
run LANES=4 GGC_MF_TAIL_ACTIVE=16000
run LANES=4 GGC_MF_TAIL_ACTIVE=64000
run LANES=4 GGC_MF_IMAGE_TAIL_PASSES=8 GGC_MF_IMAGE_TAIL_INNER=64
run LANES=4 GGC_MF_IMAGE_TAIL_PASSES=32 GGC_MF_IMAGE_TAIL_INNER=16
run LANES=4 GGC_MF_TAIL_ACTIVE=16000 GGC_MF_IMAGE_TAIL_PASSES=32 GGC_MF_IMAGE_TAIL_INNER=16
run LANES=1
run LANES=2
cat $out
