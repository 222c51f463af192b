#!/bin/bash
# kstats_var.sh PATTERN VARIANT...: average duration of the kernels matching PATTERN under each library variant (tools/build_variant.sh)
R=$GRAFT_REPO_ROOT
P=$1; shift
for v in default "$@"; do
  if [ $v = default ]; then unset GGC_HIP_LIBRARY; else export GGC_HIP_LIBRARY=$R/gcn-grabcut_amd/libggc_hip_$v.so; fi
  echo "== $v"; TOP=60 $R/tools/kstats.sh | grep "$P"
done
