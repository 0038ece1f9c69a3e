#!/bin/bash
# A / B timing of several builds of the library on ONE box, interleaved (box state drifts): tools/ab_libs.sh ROUNDS "bench args" lib...
# prints ms per step and per-kernel ms of each run.
rounds=$1; args=$2; shift 2
for r in $(seq $rounds); do
  for lib in "$@"; do
    VGT_HIP_LIB=$PWD/voxelized_geometry_tools_amd/$lib python bench.py --no-end-to-end --no-cpu-baseline --no-raycast --no-secondary $args 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-24s' % '$lib', '$args', d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
done
