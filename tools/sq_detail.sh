#!/bin/bash
# Instruction-fetch, branch and wait counters of the SDF kernels (run through gpurun from the repo root):
#   tools/sq_detail.sh <tag> [extra bench args]   -> gpurun_out/sqd_<tag>/summary.txt
set -u
TAG=${1:-run}; shift 1
OUT=gpurun_out/sqd_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-end-to-end --no-raycast $*"
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 bench.py $ARGS --no-secondary --steps 1 --warmup 1 > /dev/null 2>&1
done
python3 - $OUT > $OUT/summary.txt <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "p*", "*", "*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if "vgt::" not in name: continue
        short = "Records" if ("ClassRecordKernel" in name or "ScanZ" in name) else ("X" if "<int, float" in name else "Y")
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-32s %16.0f" % (c, sum(v) / len(v)))
PY
cat $OUT/summary.txt
