#!/bin/bash
# Times several builds of libvgt_hip on one box: tools/ab_variants.sh "<lib1> <lib2> ..." [bench args]
libs=$1; shift
for lib in $libs; do
  VGT_HIP_LIB=$PWD/voxelized_geometry_tools_amd/$lib python bench.py --no-end-to-end --no-cpu-baseline --steps 5 --warmup 2 "$@" 2>/dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['roofline'].get('kernel_ms'))"
done
