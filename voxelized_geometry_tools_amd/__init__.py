"""MI355X-native (gfx950) backend for the SDF/EDT + pointcloud-raycast hot path of
calderpg/voxelized_geometry_tools.  See DESIGN.md.

The compute path lives in ``csrc/`` (hand-written HIP behind the C ABI declared in
``include/vgt_hip.h``); this package is the thin Python host side used by the tests
and by bench.py.  Importing the package never touches the GPU; calling into it
without the built ``libvgt_hip.so`` raises immediately (there is no CPU fallback).
"""

__all__ = ["capi", "synthetic"]
