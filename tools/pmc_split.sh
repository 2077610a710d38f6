#!/bin/bash
# PMC passes of the beam kernel under tools/split_probe.py (DM_SCORER selects the mode); prints per-launch counters
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_split_${DM_SCORER:-f32}
rm -rf $OUT; mkdir -p $OUT
LIB=${1:-dismember_amd/libdismember_hip.so}
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -o p -- python tools/split_probe.py $LIB 32768 > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/p2 -o p -- python tools/split_probe.py $LIB 32768 > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES --output-format csv -d $OUT/p3 -o p -- python tools/split_probe.py $LIB 32768 > $OUT/p3.log 2>&1
python - <<'P'
import csv, glob, collections, os
out = os.environ.get("OUT_DIR", "")
for d in sorted(glob.glob("gpurun_out/pmc_split_%s/p*/" % os.environ.get("DM_SCORER", "f32"))):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float); n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if "dm_beam_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for k in acc: print("%-34s %16.0f per launch (%d launches)" % (k, acc[k] / n[k], n[k]))
P
tail -1 $OUT/p1.log
