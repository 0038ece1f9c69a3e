#!/bin/bash
# Builds diagnostic variants of the library that differ in voxelizer_kernels.hip's compile flags only, one per
# (name, "flags") pair, into diag_<name>/libvgt_diag.so (the other objects are the product build's); tools/raycast_variants.sh
# then benches every diag_*/ next to the product library on one box.  The -D switches of an experiment belong to
# diagnostic #if blocks patched into the kernel for that experiment and removed again (profiles/r5/experiments.md says
# which ones each table row used); the script itself is generic.
#   tools/raycast_build_variants.sh s16 "-DVGT_RC_SEGMENT=16" s32 "-DVGT_RC_SEGMENT=32" ...
cd "$(dirname "$0")/../voxelized_geometry_tools_amd/csrc" || exit 1
ROOT=$(cd ../.. && pwd)
BASE="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -fvisibility=hidden -fvisibility-inlines-hidden -Wno-unused-parameter"
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  d=$ROOT/diag_$name; mkdir -p $d
  for f in cell_kernels edt_kernels edt_record_kernels edt_short_kernels edt_sweep_kernels vgt_hip_capi vgt_hipx_multi; do cp -u $f.o $d/ 2>/dev/null; done
  ( hipcc $BASE $flags -I../../include -c voxelizer_kernels.hip -o $d/voxelizer_kernels.o && hipcc -shared -fPIC --offload-arch=gfx950 -o $d/libvgt_diag.so $d/*.o -Wl,--version-script=exports.map && echo built $name ) &
done
wait
