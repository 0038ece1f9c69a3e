#!/bin/bash
# Counters of the dense 1024^3 scenes (D2 salt, D3 unknown mix) on the GPU box, same recipe as tools/collect_profiles.sh:
# kernel stats, HBM traffic (separate --pmc passes for FETCH_SIZE / WRITE_SIZE) and the SQ set, per scene.
# Usage: tools/collect_dense_counters.sh <tag> <commit>     -> gpurun_out/profiles_<tag>_<dist>/
set -u
TAG=${1:-run}
COMMIT=${2:-unknown}
export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-end-to-end --no-raycast --no-secondary"
for DIST in ${DISTS:-salt unknown_mix}; do
  OUT=gpurun_out/profiles_${TAG}_$DIST
  mkdir -p $OUT
  echo $COMMIT > $OUT/commit.txt
  timeout 300 python3 bench.py --dist $DIST $COMMON > $OUT/bench1024_$DIST.json 2> /dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --dist $DIST --steps 10 --warmup 3 $COMMON > /dev/null 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 bench.py --dist $DIST --steps 2 --warmup 1 $COMMON > /dev/null 2>&1
  done
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py --dist $DIST --steps 1 --warmup 1 $COMMON > /dev/null 2>&1
  python3 tools/summarize_profiles.py $OUT > /dev/null
  # (the summaries' "command" fields name the D1 run: say which scene this was)
  python3 - $OUT $DIST <<'EOF'
import json, sys
out, dist = sys.argv[1], sys.argv[2]
for name in ("pmc_hbm_traffic.json", "sq_counters.json"):
    try:
        d = json.load(open(out + "/" + name))
    except OSError:
        continue
    d["scene"] = "1024^3 " + dist
    d["command"] = d["command"].replace("bench.py", "bench.py --dist " + dist).replace("(1024^3 D1 spheres)", "").replace("(1024^3 D1)", "")
    json.dump(d, open(out + "/" + name.replace(".json", "_" + dist + ".json"), "w"), indent=1)
EOF
  rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_sq $OUT/stats
done
