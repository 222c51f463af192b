#!/bin/bash
out=gpurun_out/mfb_sweep.log; : > $out
run() { env "$@" timeout -k 10 120 python tools/mf_time.py 2>&1 | grep "grabcut stage\|Error" >> $out; }
for b in 1 8 32 64; do
  for drv in host image pool; do run LANES=1 MF_BATCH=$b GGC_MF_DRIVER=$drv REPS=3; done
done
cat $out
