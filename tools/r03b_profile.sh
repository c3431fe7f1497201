#!/bin/bash
# Round-3 (second half) artefacts in one GPU call: c2 profile with PMC passes (traffic.json at HEAD), bench lines, next rows, training
# kernel stats + timeline, per-rank step times, actuator step.
set -e
R=$GRAFT_REPO_ROOT
COMMIT=$1
cd $R
tools/profile_round.sh r3b_c2 $COMMIT c2/f64/gram gram_kernel kfu_build --
TAG=r3b bash tools/final_prof.sh
python3 tools/actuator_step.py > gpurun_out/r3b/actuator_step.txt 2>&1 || true
FFVD_NO_SMALL_SIDE=1 python3 tools/actuator_step.py forward 2>&1 | sed 's/^/FFVD_NO_SMALL_SIDE=1 /' >> gpurun_out/r3b/actuator_step.txt || true
cat gpurun_out/r3b/actuator_step.txt
tools/prof_train_timeline.sh r3b_tt > gpurun_out/r3b/train_step_timeline.txt 2>&1 || true
