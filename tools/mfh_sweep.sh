#!/bin/bash
out=gpurun_out/mfh_sweep.log; : > $out
run() { env "$@" timeout -k 10 120 python tools/mf_time.py 2>&1 | grep "grabcut stage\|Error" >> $out; }
run LANES=4 GGC_MF_DRIVER=host
run LANES=4 GGC_MF_HANDOFF_PER_IMAGE=16
run LANES=4 GGC_MF_HANDOFF_PER_IMAGE=64
run LANES=4 GGC_MF_HANDOFF_PER_IMAGE=256
run LANES=4 GGC_MF_HANDOFF_PER_IMAGE=1024
run LANES=4 GGC_MF_HANDOFF_PER_IMAGE=4096
run LANES=1 GGC_MF_HANDOFF_PER_IMAGE=256
run LANES=2 GGC_MF_HANDOFF_PER_IMAGE=256
run LANES=8 GGC_MF_HANDOFF_PER_IMAGE=256
cat $out
