#!/bin/bash
# Builds the committed library as libvgt_hip_base.so and the working tree as libvgt_hip_new.so (and libvgt_hip.so),
# for tools/ab_variants.sh "libvgt_hip_base.so libvgt_hip_new.so".  Run from the repo root on the build machine.
set -e
cd voxelized_geometry_tools_amd/csrc
git stash -q
make -j8 > /dev/null
cp ../libvgt_hip.so ../libvgt_hip_base.so
git stash pop -q
make -j8 > /dev/null
cp ../libvgt_hip.so ../libvgt_hip_new.so
ls -la ../libvgt_hip_base.so ../libvgt_hip_new.so
