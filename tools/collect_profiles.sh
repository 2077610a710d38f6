#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + PMC passes of the default bench workload.
# Outputs under gpurun_out/prof_$1/ ; summaries are copied into profiles/ by tools/summarize_profiles.py
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--steps 4 --warmup 1 --cpu-users 0 --recall-users 0 --small 0 --train 0 --dr 0 --other-scorer 0 ${BENCH_EXTRA:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python bench.py $ARGS > $OUT/bench_trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o p -- python bench.py $ARGS > $OUT/bench_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python bench.py $ARGS > $OUT/bench_pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python bench.py $ARGS > $OUT/bench_pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -o p -- python bench.py $ARGS > $OUT/bench_pmc4.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_lds -o p -- python bench.py $ARGS > $OUT/bench_pmc5.log 2>&1
ls -R $OUT | head -40
