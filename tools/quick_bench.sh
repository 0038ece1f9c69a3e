#!/bin/bash
# Quick timing of the SDF kernels on the GPU box: tools/quick_bench.sh [lib.so ...]  (default: the built library)
# prints ms per step and per-kernel ms for 1024^3 D1, 1024^3 salt and 512^3 D1.
libs=${@:-voxelized_geometry_tools_amd/libvgt_hip.so}
for lib in $libs; do
  for w in "" "--dist salt" "--size 512"; do
    VGT_HIP_LIB=$PWD/$lib python bench.py --no-end-to-end --no-cpu-baseline --no-raycast --no-secondary $w 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', '$w', d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
done
