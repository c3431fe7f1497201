#!/bin/bash
# Tuning builds of the fp32-contraction kernels: tools/build_variant.sh <KT> <OCC> -> ffvd_amd/libffvd_hip_kt<KT>_occ<OCC>.so
# (same objects as the product library except kernels_f32; load with FFVD_LIB=<path>)
set -e
cd "$(dirname "$0")/.."
python -m ffvd_amd.build > /dev/null
KT=$1; OCC=$2
OUT=ffvd_amd/libffvd_hip_kt${KT}_occ${OCC}.so
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-value -DFFVD_F32_KT=$KT -DFFVD_F32_OCC=$OCC -c ffvd_amd/csrc/kernels_f32.hip -o /tmp/kernels_f32_${KT}_${OCC}.o
/opt/rocm/bin/hipcc ffvd_amd/build/kernels.hip.o /tmp/kernels_f32_${KT}_${OCC}.o ffvd_amd/build/grad.hip.o ffvd_amd/build/optim.hip.o ffvd_amd/build/abi.hip.o -shared -fPIC --offload-arch=gfx950 -ldl -o $OUT
echo $OUT
