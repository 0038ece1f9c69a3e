#!/bin/bash
# Diagnostic: per-kernel times of an EDT variant for several tile widths (needs the DEBUG build:
# make -C voxelized_geometry_tools_amd/csrc DEBUG=1 OUT=../libvgt_hip_dbg.so OBJDIR=dbg)
# usage: tools/bench_variants.sh <variant> [extra bench.py args]
v=$1; shift
for w in 8 16 32; do
  echo "== variant $v W=$w $*"
  VGT_HIP_LIB=voxelized_geometry_tools_amd/libvgt_hip_dbg.so VGT_HULL_W=$w python3 bench.py --variant $v --steps 5 --warmup 2 --no-cpu-baseline "$@" 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['sdf_min_max'])
    elif l: print(l[:300])
"
done
