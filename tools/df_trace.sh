#!/bin/bash
# Debug build of the library with wall-clock stamps in the dataflow Cholesky (tools/df_trace.py reads them):
#   tools/df_trace.sh  ->  ffvd_amd/libffvd_hip_dftrace.so
set -e
cd "$(dirname "$0")/.."
python -m ffvd_amd.build --dftrace
