#!/usr/bin/env python3
"""Per-phase cycle shares and evaluation counts of the tiled divide & conquer passes.
Needs the diagnostic build (see tools/hull_stats.py).  Usage: python tools/dc_stats.py [size] [dist]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VGT_HIP_LIB", os.path.join(ROOT, "voxelized_geometry_tools_amd", "libvgt_hip_stats.so"))

import numpy as np
import torch

import bench
from voxelized_geometry_tools_amd import capi

NAMES = ["cyc_load", "cyc_prep", "cyc_queries", "cyc_rows", "workgroups", "cyc_total", "row_evals",
         "query_scorings", "max_evals_thread"]


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    dist = sys.argv[2] if len(sys.argv) > 2 else "spheres"
    shape = (size,) * 3
    dev = torch.device("cuda", 0)
    occ = bench.device_occupancy(torch, shape, dist, 42, dev)
    sdf = torch.empty(shape, dtype=torch.float32, device=dev)
    nbytes = capi.sdf_workspace_bytes(shape)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    ctx = capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    lib = capi.load()
    out = (ctypes.c_ulonglong * 32)()
    lib.vgt_hip_debug_dc_stats(out, 1)
    ctx.sdf_dev(occ.data_ptr(), shape, 0.01, sdf.data_ptr(), ws.data_ptr(), nbytes)
    torch.cuda.synchronize()
    lib.vgt_hip_debug_dc_stats(out, 1)
    vox = float(np.prod(shape))
    for base, name in ((0, "Y pass"), (16, "X pass")):
        vals = [out[base + i] for i in range(9)]
        wg = max(vals[4], 1)
        print("== %s (%s^3 %s): %d workgroups" % (name, size, dist, vals[4]))
        for i in range(4):
            print("  %-12s %10.0f cycles/WG  %5.1f %%" % (NAMES[i], vals[i] / wg, 100.0 * vals[i] / max(vals[5], 1)))
        print("  %-16s %14d  %8.3f per voxel" % (NAMES[6], vals[6], vals[6] / vox))
        print("  %-16s %14d  %8.3f per voxel" % (NAMES[7], vals[7], vals[7] / vox))
        print("  %-16s %14d" % (NAMES[8], vals[8]))


if __name__ == "__main__":
    main()
