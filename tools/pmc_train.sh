#!/bin/bash
# PMC passes of the training step (tools/prof_grad.py): HBM bytes per launch and matrix-pipe utilisation per kernel (run on the GPU box)
#   tools/pmc_train.sh <tag> <commit>
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1; COMMIT=$2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/prof_grad.py > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/tools/prof_grad.py > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 $R/tools/prof_grad.py > /dev/null 2> $OUT/pmc_sq.err
python3 $R/tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_summary.txt $OUT/traffic_train.json "c2/f64/gram/train" "$COMMIT" gram_kernel bwd_fused > /dev/null
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
sed -i 's/python3 bench.py .../python3 tools\/prof_grad.py/' $OUT/pmc_summary.txt
cat $OUT/pmc_summary.txt | cut -c1-190
