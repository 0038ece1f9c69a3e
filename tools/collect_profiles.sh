#!/bin/bash
# Collects the round's evidence on the GPU box (run through gpurun from the repo root):
#   bench lines (headline with raycast / host paths / CPU baseline, 512^3, salt, config 5 on one GPU, config 5 through
#   the slab path on one GPU), rocprofv3 kernel stats, HBM traffic counters (separate --pmc passes), SQ counters of the
#   line passes, and the raycaster's kernel stats + atomic counters.
# Usage: tools/collect_profiles.sh <tag> <commit>      -> gpurun_out/profiles_<tag>/
set -u
TAG=${1:-run}
COMMIT=${2:-unknown}
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo $COMMIT > $OUT/commit.txt
# ONLY=raycast tools/collect_profiles.sh <tag> <commit>: the raycaster's part alone (after a change to voxelizer_kernels.hip)
if [ "${ONLY:-all}" != raycast ]; then
timeout 600 python3 bench.py > $OUT/bench1024.json 2> $OUT/bench1024.err
timeout 300 python3 bench.py --dist salt --no-cpu-baseline --no-end-to-end --no-raycast > $OUT/bench1024_salt.json 2> /dev/null
timeout 300 python3 bench.py --dist unknown_mix --no-cpu-baseline --no-end-to-end --no-raycast > $OUT/bench1024_unknown_mix.json 2> /dev/null
timeout 300 python3 bench.py --size 512 --no-cpu-baseline --no-end-to-end > $OUT/bench512.json 2> /dev/null
timeout 600 python3 tools/slab_rank_geometry.py --out $OUT/slab_rank_geometry.json > /dev/null 2> $OUT/slab_rank_geometry.err
timeout 300 python3 bench.py --workload c5 --no-cpu-baseline --steps 5 --warmup 2 > $OUT/bench_c5_one_gpu.json 2> /dev/null
python3 -c "import json,sys; p=sys.argv[1]; d=json.load(open(p)); d['commit']=sys.argv[2]; open(p,'w').write(json.dumps(d)+'\n')" $OUT/bench_c5_one_gpu.json $COMMIT
timeout 300 python3 bench.py --force-slab --steps 3 --warmup 1 > $OUT/bench_config5_one_gpu_slab_path.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --no-raycast --no-secondary > $OUT/bench_under_rocprof.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-raycast --no-secondary > /dev/null 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end --no-raycast --no-secondary > /dev/null 2>&1
fi
# raycaster: the bench line, the atomics micro-benchmark, kernel stats and the L2's atomic request counters (whatever this
# rocprofv3 calls them)
make -s -C tools/microbench scattered_atomics > /dev/null 2>&1 && tools/microbench/scattered_atomics > $OUT/microbench_scattered_atomics.json 2> /dev/null
timeout 300 python3 bench_raycast.py > $OUT/bench_raycast_config3.json 2> /dev/null
timeout 120 tests/cpp/bench_voxelize > $OUT/bench_voxelize_end_to_end.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raycast_stats -- python3 bench_raycast.py --no-check --steps 5 > /dev/null 2>&1
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*ATOMIC[A-Z0-9_]*" | sort -u > $OUT/available_atomic_counters.txt
ATOMIC=$(grep -E "^TCC_(EA0_)?ATOMIC(_sum)?$|^TCC_ATOMIC_sum$|^TCC_EA0_ATOMIC_sum$" $OUT/available_atomic_counters.txt | head -2 | tr '\n' ' ')
if [ -n "$ATOMIC" ]; then
  rocprofv3 --pmc $ATOMIC --kernel-trace --output-format csv -d $OUT/raycast_pmc_atomic -- python3 bench_raycast.py --no-check --steps 2 --warmup 1 > /dev/null 2>&1
fi
python3 tools/summarize_profiles.py $OUT
