#!/bin/bash
# The C-ABI library's HOST code and the C++ glue under AddressSanitizer, on a box WITH a device: the library is built
# with -fsanitize=address -fno-gpu-sanitize (host side instrumented, device code as always), the C++ host test is built
# with the same compiler and the shared sanitizer runtime, and runs its whole suite (SDF, batches, tagged maps,
# voxelizer, multi-slab entry point) against the real device.  tests/test_sanitizers.py covers the same host code
# without a device; GPU AddressSanitizer (xnack) is not available on this pool and is not used.
#   build here (no GPU needed):   tools/host_asan_on_device.sh build
#   run on the GPU box:           gpurun -- tools/host_asan_on_device.sh run
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/voxelized_geometry_tools_amd
CL=/opt/rocm/lib/llvm/bin/clang++
RT=$(dirname "$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)")
case "${1:-build}" in
build)
  make -s -j8 -C "$PKG/csrc" product OBJDIR=asan_host OUT=../libvgt_hip_hostasan.so \
    HIPFLAGS="-O1 -g -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -fvisibility=hidden -fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer"
  $CL -std=c++17 -O1 -g -pthread -fsanitize=address -shared-libasan -fno-omit-frame-pointer \
    -o "$ROOT/tests/cpp/test_hip_host_devasan" "$ROOT/tests/cpp/test_hip_host.cc" \
    "$PKG/csrc/host/hip_voxelization_helpers.cc" "$PKG/csrc/host/hip_pointcloud_voxelizer.cc" \
    -L"$PKG" -lvgt_hip_hostasan -Wl,-rpath,"$PKG" -Wl,-rpath,"$RT"
  ;;
run)
  # (the HIP runtime keeps allocations for the life of the process: leak detection off)
  ASAN_OPTIONS=detect_leaks=0 "$ROOT/tests/cpp/test_hip_host_devasan"
  ;;
esac
