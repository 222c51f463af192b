#!/bin/bash
# build_variant.sh NAME FILE.hip "-DFLAGS": libggc_hip_NAME.so = the library with one translation unit rebuilt with extra flags
set -e
cd "$(dirname "$0")/../gcn-grabcut_amd"
make -s
mkdir -p build/var
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-variable"
/opt/rocm/bin/hipcc $F $3 -c csrc/$2.hip -o build/var/$2_$1.o
OBJS=$(ls build/*.o | grep -v "/$2.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libggc_hip_$1.so $OBJS build/var/$2_$1.o
echo built libggc_hip_$1.so
