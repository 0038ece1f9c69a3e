#!/bin/bash
# SQ counters of the raycast kernel (bench_raycast.py, cloud A launches first, then cloud B): instruction mix, LDS
# conflicts and waits.  Separate --pmc passes; prints per-kernel means for the RaycastKernel dispatches of cloud A.
#   gpurun -- tools/raycast_sq_counters.sh [outdir]
OUT=${1:-gpurun_out/raycast_sq}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
pass() {
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 bench_raycast.py --no-check --steps 2 --warmup 1 > /dev/null 2>&1
}
pass mix SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES
pass busy SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY
pass wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_LDS_UNALIGNED_STALL
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/*/*/*counter_collection.csv") + glob.glob(out + "/*/*counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(f)) if "RaycastKernel" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    first_half = set(ids[:len(ids) // 2])
    per = collections.defaultdict(list)
    for r in rows:
        if int(r["Dispatch_Id"]) in first_half:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in per.items():
        print("%-28s %16.0f" % (k, sum(v) / len(v)))
PY
