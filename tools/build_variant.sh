#!/bin/bash
# Build a variant of the library for A/B timing:  bash tools/build_variant.sh NAME "-DFLAG=1 ..."  ->  bhnerf_amd/csrc/libbhnerf_hip_NAME.so
# (objects in /tmp/bhn_NAME; the product build is untouched).  Time the variants on ONE box with tools/ab.sh.
set -euo pipefail
NAME=$1; FLAGS=${2:-}
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/bhnerf_amd/csrc; O=/tmp/bhn_$NAME
mkdir -p $O
for f in simple_kernels fused_fwd fused_bwd fused_bwd128 general_mlp selftest; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed $FLAGS -c $C/$f.hip -o $O/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -Wl,--version-script=$C/export.map $O/*.o -o $C/libbhnerf_hip_$NAME.so
echo built $C/libbhnerf_hip_$NAME.so
