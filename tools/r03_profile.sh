#!/bin/bash
# Round-3 artefacts in one GPU call: bench lines (c2, c4, c2 reference route, c5), kernel stats + PMC passes (traffic.json at HEAD),
# next rows, per-rank step times, actuator step, dataflow Cholesky trace.
set -e
R=$GRAFT_REPO_ROOT
COMMIT=$1
cd $R
tools/profile_round.sh r3final_c2 $COMMIT c2/f64/gram gram_kernel kfu_build --
tools/profile_round.sh r3final_c4 $COMMIT c4/f32c/reference gram_f32_kernel proj_gemm_f32_kernel -- --workload c4
TAG=r3final bash tools/final_prof.sh
python3 tools/actuator_step.py > gpurun_out/r3final/actuator_step.txt 2>&1 || true
cat gpurun_out/r3final/actuator_step.txt
