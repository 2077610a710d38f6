#!/bin/bash
# Run on the GPU box (via gpurun): round-6 profile set.  Outputs under gpurun_out/prof_<tag>/; tools/summarize_profiles.py condenses
# them into profiles/.
#   (1) kernel-trace stats of the WHOLE default bench line (every kernel of every extra: beam, fp64 beam, rows, train, wgrad, Adam,
#       JTM expand / sum, sampler, Deep-Retrieval)                                                   -> prof_r06_all
#   (2) kernel trace + PMC passes of the headline workload only, default (split-fp16) scorer        -> prof_r06
#   (3) the same with --scorer f32                                                                   -> prof_r06_f32
#   (4) kernel trace + PMC passes of the fp64 OTM beam kernel at depth 24 (tools/otm_f64_bench.py)   -> prof_r06_otm64_d24
#   (5) kernel trace + PMC passes of the Deep-Retrieval search, fp64 and f32 (tools/dr_bench.py)     -> prof_r06_dr_f64 / prof_r06_dr_f32
#   (7) `longhist` (not part of `all`): tools/long_history_bench.py 32768 (the two-key-tile kernels)   -> prof_r06_longhist
#   (6) `otmtrain` (not part of `all`): the fp64 OTM training iteration at 8 192 users                 -> prof_r06_otmtrain
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WHAT=${1:-all}
pmc_set() {   # $1 = out dir, rest = command
  local OUT=$1; shift
  mkdir -p $OUT
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- "$@" > $OUT/bench_trace.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o p -- "$@" > $OUT/bench_pmc1.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- "$@" > $OUT/bench_pmc2.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- "$@" > $OUT/bench_pmc3.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -o p -- "$@" > $OUT/bench_pmc4.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_lds -o p -- "$@" > $OUT/bench_pmc5.log 2>&1
}
HEAD="--steps 4 --warmup 1 --cpu-users 0 --recall-users 0 --small 0 --train 0 --dr 0 --other-scorer 0 --otm64 0 --diverse 0 --long-history 0 --host-buffer-steps 0 --jtm-full 0"
if [ "$WHAT" = all ] || [ "$WHAT" = full ]; then
  mkdir -p gpurun_out/prof_r06_all
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r06_all/trace -o t -- python bench.py --steps 6 --warmup 1 --cpu-users 0 --trained-recall-steps 200 > gpurun_out/prof_r06_all/bench_trace.log 2>&1
fi
if [ "$WHAT" = all ] || [ "$WHAT" = head ]; then pmc_set gpurun_out/prof_r06 python bench.py $HEAD; fi
if [ "$WHAT" = all ] || [ "$WHAT" = f32 ]; then pmc_set gpurun_out/prof_r06_f32 python bench.py $HEAD --scorer f32; fi
if [ "$WHAT" = all ] || [ "$WHAT" = otm64 ]; then pmc_set gpurun_out/prof_r06_otm64_d24 python tools/otm_f64_bench.py 24 16384 f64only; fi
if [ "$WHAT" = all ] || [ "$WHAT" = dr ]; then
  pmc_set gpurun_out/prof_r06_dr_f64 python tools/dr_bench.py --dtype f64 --rerank 0 --steps 4
  pmc_set gpurun_out/prof_r06_dr_f32 python tools/dr_bench.py --dtype f32 --rerank 0 --steps 4
fi
if [ "$WHAT" = drf64 ]; then pmc_set gpurun_out/prof_r06_dr_f64 python tools/dr_bench.py --dtype f64 --rerank 0 --steps 4; fi     # after the conflict-free staging stores
if [ "$WHAT" = drf32 ]; then pmc_set gpurun_out/prof_r06_dr_f32 python tools/dr_bench.py --dtype f32 --rerank 0 --steps 4; fi     # after the 256 x 256 history GEMM
if [ "$WHAT" = jtm ]; then pmc_set gpurun_out/prof_r06_jtm python tools/jtm_bench.py 10000000 24; fi     # JTM.optimize at 10 M items: the general-rows split kernel
if [ "$WHAT" = otmtrain ]; then pmc_set gpurun_out/prof_r06_otmtrain python tools/otm_train_bench.py 24 8192 f64; fi     # fp64 OTM training iteration at train_batch_size 8192
if [ "$WHAT" = diverse ]; then       # the headline search on beams that diverge (tools/diverse_bench.py) beside the shared-beam headline model
  pmc_set gpurun_out/prof_r06_diverse python tools/diverse_bench.py 131072 4 s1.7e32
  pmc_set gpurun_out/prof_r06_diverse_head python tools/diverse_bench.py 131072 4 head
fi
if [ "$WHAT" = longhist ]; then pmc_set gpurun_out/prof_r06_longhist python tools/long_history_bench.py 32768; fi     # histories of 17 / 24 / 32 positions: dm_beam_kernel<128, 4, true, 2> and dm_beam64_kernel<128, 4, 2>
ls gpurun_out/prof_r06*/ | head -40
