#!/bin/bash
# SQ counters and HBM traffic of the SDF kernels for one EDT variant (run through gpurun from the repo root):
#   tools/profile_variant.sh <variant> <tag> [extra bench args]   -> gpurun_out/pv_<tag>/
set -u
V=${1:-0}; TAG=${2:-run}; shift 2
OUT=gpurun_out/pv_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--variant $V --no-cpu-baseline --no-end-to-end $*"
python3 bench.py $ARGS --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS --steps 10 --warmup 3 > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 bench.py $ARGS --steps 2 --warmup 1 > /dev/null 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py $ARGS --steps 1 --warmup 1 > /dev/null 2>&1
python3 tools/summarize_profiles.py $OUT > $OUT/summary.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
for d in ("pmc_sq", "pmc_sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(out, d, "*", "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            if "vgt::" not in name: continue
            short = "ScanZ" if "ScanZ" in name else ("X" if ("Lb1E" in name or "true" in name.split("Kernel")[-1][:40]) else "Y")
            acc[short + " " + name[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print(d, k, {c: sum(x) / len(x) for c, x in v.items()})
PY
cat $OUT/bench.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'].get('kernel_ms'))"
cat $OUT/pmc_hbm_traffic.json 2>/dev/null | python3 -c "import sys,json; d=json.load(sys.stdin); [print(k, v['fetch_bytes_raw']/1e9, v['write_bytes']/1e9) for k,v in d['kernels'].items()]"
head -8 $OUT/rocprof_kernel_stats.csv 2>/dev/null | cut -c1-200
