#!/usr/bin/env python3
"""Per-phase cycle shares and operation counts of the tiled envelope passes.

Needs the diagnostic build:  make -C voxelized_geometry_tools_amd/csrc testing STATS=1 \
    OUT=../libvgt_hip_stats.so OBJDIR=stats
(the tiled envelope lives in the TESTING library, which the Makefile names after OUT: libvgt_hip_stats_testing.so)
Usage: VGT_HIP_LIB=voxelized_geometry_tools_amd/libvgt_hip_stats_testing.so python tools/hull_stats.py [size] [dist]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VGT_HIP_LIB", os.path.join(ROOT, "voxelized_geometry_tools_amd", "libvgt_hip_stats_testing.so"))

import numpy as np
import torch

import bench
from voxelized_geometry_tools_amd import capi

NAMES = ["cyc_load", "cyc_local", "cyc_merge", "cyc_starts", "cyc_eval", "workgroups", "cyc_total",
         "predicates", "local_pops", "merge_kills", "survivors", "-", "max_merge_walk"]


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
    lib.vgt_hip_debug_hull_stats(out, 1)
    ctx.sdf_dev(occ.data_ptr(), shape, 0.01, sdf.data_ptr(), ws.data_ptr(), nbytes)
    torch.cuda.synchronize()
    lib.vgt_hip_debug_hull_stats(out, 1)
    vox = float(np.prod(shape))
    for base, name in ((0, "Y pass"), (16, "X pass")):
        vals = [out[base + i] for i in range(13)]
        wg = max(vals[5], 1)
        print("== %s (%s^3 %s): %d workgroups" % (name, size, dist, vals[5]))
        for i in range(5):
            print("  %-12s %10.0f cycles/WG  %5.1f %%" % (NAMES[i], vals[i] / wg, 100.0 * vals[i] / max(vals[6], 1)))
        for i in range(7, 11):
            print("  %-14s %14d  %8.3f per voxel" % (NAMES[i], vals[i], vals[i] / vox))
        print("  %-14s %14d" % (NAMES[12], vals[12]))
        sub = [out[base + i] for i in (11, 13, 14, 15)]
        print("  local split: rows->regs %d, carries %d (thread 0); seeds %d, prefilter %d, prefilter done at %d, "
              "stack done at %d cycles after load (thread 64)" % tuple(v / wg for v in sub + [out[base + 12], out[base + 9]]))


if __name__ == "__main__":
    main()
