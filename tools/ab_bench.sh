#!/bin/bash
# A/B timing of two builds of libvgt_hip on the same box: tools/ab_bench.sh <libA.so> <libB.so> [bench args]
# prints ms per step and per-kernel ms, alternating A B A B.
a=$1; b=$2; shift 2
for lib in $a $b $a $b; do
  VGT_HIP_LIB=$lib python bench.py --no-end-to-end --no-cpu-baseline "$@" 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['roofline']['kernel_ms'])"
done
