#!/bin/bash
# Collects the round's evidence on the GPU box (run through gpurun from the repo root):
#   bench line, rocprofv3 kernel stats, HBM traffic counters (separate --pmc passes).
# Usage: tools/collect_profiles.sh <tag>      -> gpurun_out/profiles_<tag>/
set -u
TAG=${1:-run}
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/bench1024.json 2> $OUT/bench1024.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
python3 tools/summarize_profiles.py $OUT
