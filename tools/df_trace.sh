#!/bin/bash
# Debug build of the library with wall-clock stamps in the dataflow Cholesky (tools/df_trace.py reads them):
#   tools/df_trace.sh  ->  ffvd_amd/libffvd_hip_dftrace.so
set -e
cd "$(dirname "$0")/.."
python -m ffvd_amd.build > /dev/null
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-value -DFFVD_DF_TRACE -c ffvd_amd/csrc/kernels.hip -o /tmp/kernels_dftrace.o
/opt/rocm/bin/hipcc /tmp/kernels_dftrace.o ffvd_amd/build/kernels_f32.hip.o ffvd_amd/build/grad.hip.o ffvd_amd/build/optim.hip.o ffvd_amd/build/abi.hip.o -shared -fPIC --offload-arch=gfx950 -ldl -o ffvd_amd/libffvd_hip_dftrace.so
echo ffvd_amd/libffvd_hip_dftrace.so
