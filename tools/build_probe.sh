#!/bin/bash
# Probe build of the library: tools/build_probe.sh DM_PHASE_TIMERS [more -D switches]  ->  tools/_bin/libdm_probe_<first switch>.so
# (knock-outs and in-kernel timers only compile with -DDM_PROBE_BUILD: beam_kernel.hip.inc; never used by the product build)
set -eu
cd "$(dirname "$0")/.."
mkdir -p tools/_bin
DEFS="-DDM_PROBE_BUILD"; for d in "$@"; do DEFS="$DEFS -D$d"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -pthread -shared -fPIC $DEFS -o tools/_bin/libdm_probe_$1.so \
  dismember_amd/csrc/dm_hip.hip -I/opt/rocm/include -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo tools/_bin/libdm_probe_$1.so
