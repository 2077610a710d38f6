#!/bin/bash
# Samples socket power / clocks with rocm-smi while the headline bench runs (GPU box): is the beam kernel power-managed?
#   bash tools/power_probe.sh [extra bench.py args]   -> gpurun_out/power_probe.log
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/power_probe.log
: > $OUT
( while true; do rocm-smi --showpower --showclocks --showuse --csv 2>/dev/null | tail -n +2 | head -2 | tr '\n' ' ' >> $OUT; echo >> $OUT; sleep 0.2; done ) &
SAMPLER=$!
python bench.py --steps 60 --warmup 2 --cpu-users 0 --recall-users 0 --small 0 --train 0 --dr 0 --other-scorer 0 --otm64 0 --jtm-full 0 "$@" > gpurun_out/power_probe_bench.json 2> gpurun_out/power_probe_bench.err
kill $SAMPLER
rocm-smi --showmaxpower --showpower --csv 2>/dev/null | tail -3 >> $OUT
wc -l $OUT
