#!/bin/bash
# L2 / memory-side counters of the dominant kernels (which bound is it: matrix pipe, L2, fabric / HBM?).
#   tools/pmc_tcc.sh <tag> -- <bench.py args...>       (on the GPU box; output gpurun_out/<tag>/tcc_summary.txt)
# Separate rocprofv3 --pmc passes (4 TCC slots per pass), bench.py --steps 2 --warmup 1.
set -e
TAG=$1; shift 2
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHORT="$@ --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum --output-format csv -d $OUT/p1 -- python3 $R/bench.py $SHORT > /dev/null 2> $OUT/p1.err
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/p2 -- python3 $R/bench.py $SHORT > /dev/null 2> $OUT/p2.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p3 -- python3 $R/bench.py $SHORT > /dev/null 2> $OUT/p3.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/p4 -- python3 $R/bench.py $SHORT > /dev/null 2> $OUT/p4.err || true
python3 $R/tools/pmc_tcc_summary.py $OUT > $OUT/tcc_summary.txt
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
cat $OUT/tcc_summary.txt
