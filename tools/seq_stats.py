#!/usr/bin/env python3
"""Diagnostic (DEBUG build of the library): per-phase workgroup cycles of the sequential-band
envelope passes on one SDF extraction.  usage: tools/seq_stats.py [size] [dist] [variant]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VGT_HIP_LIB", os.path.join(ROOT, "voxelized_geometry_tools_amd", "libvgt_hip_dbg.so"))
import torch  # noqa: E402
import bench  # noqa: E402
from voxelized_geometry_tools_amd import capi  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dist = sys.argv[2] if len(sys.argv) > 2 else "spheres"
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 2
shape = (size,) * 3
occ = bench.device_occupancy(torch, shape, dist, 42, torch.device("cuda", 0))
sdf = torch.empty(shape, dtype=torch.float32, device="cuda")
nbytes = capi.sdf_workspace_bytes(shape, variant)
ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
ctx = capi.Context(0)
ctx.set_stream(None)
ctx.set_edt_variant(variant)
raw = ctypes.CDLL(capi.LIB_PATH)
stats = (ctypes.c_ulonglong * 16)()
for it in range(2):
    ctx.sdf_dev(occ.data_ptr(), shape, 0.01, sdf.data_ptr(), ws.data_ptr(), nbytes)
    torch.cuda.synchronize()
    raw.vgt_hip_debug_seq_stats(stats, 1)
names = ["load", "H", "C", "J", "E"]
for p, label in ((0, "Y"), (8, "X")):
    v = np.array([stats[p + i] for i in range(8)], dtype=np.float64)
    wgs = max(v[5], 1.0)
    total = v[:5].sum()
    print("%s pass: %d workgroups, %.0f cycles/workgroup, %.2f junction rounds; " % (label, wgs, total / wgs, v[6] / wgs) +
          ", ".join("%s %.1f%%" % (n, 100.0 * c / total) for n, c in zip(names, v[:5])))
