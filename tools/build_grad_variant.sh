#!/bin/bash
# Diagnostic builds of grad.hip: tools/build_grad_variant.sh <name> <-D...>  ->  ffvd_amd/libffvd_hip_<name>.so
# (every other object shared with the product library; load with FFVD_LIB=<path>)
set -e
cd "$(dirname "$0")/.."
python -m ffvd_amd.build > /dev/null
NAME=$1; shift
OUT=ffvd_amd/libffvd_hip_${NAME}.so
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-value "$@" -c ffvd_amd/csrc/grad.hip -o /tmp/grad_${NAME}.o
/opt/rocm/bin/hipcc ffvd_amd/build/kernels.hip.o ffvd_amd/build/kernels_f32.hip.o /tmp/grad_${NAME}.o ffvd_amd/build/optim.hip.o ffvd_amd/build/abi.hip.o -shared -fPIC --offload-arch=gfx950 -ldl -o $OUT
echo $OUT
