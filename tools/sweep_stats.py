#!/usr/bin/env python3
"""Event counts of the sweep passes (the default EDT line passes) from a diagnostic build:
   make -C voxelized_geometry_tools_amd/csrc OBJDIR=sstats OUT=../libvgt_hip_sstats.so HIPFLAGS="... -DVGT_SWEEP_STATS"
   (event counts; slow) or -DVGT_SWEEP_TIMING (item durations and wave exit times; near full speed)
   VGT_HIP_LIB=.../libvgt_hip_sstats.so python tools/sweep_stats.py [size] [dist]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from voxelized_geometry_tools_amd import capi

NAMES = ["lane refills sweep1", "wave refill events sweep1", "lane refills sweep2", "wave refill events sweep2",
         "spilled chunks (lanes)", "wave spill events", "exact conversions (wave events)", "wave pop iterations sweep1",
         "wave pop iterations sweep2", "waves with class changes", "waves", "lane pops sweep1", "lane pops sweep2",
         "lane pushes"]


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    dist = sys.argv[2] if len(sys.argv) > 2 else "spheres"
    shape = (size, size, size)
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    occ = bench.device_occupancy(torch, shape, dist, 42, dev)
    sdf = torch.empty(shape, dtype=torch.float32, device=dev)
    nbytes = capi.sdf_workspace_bytes(shape)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    mm = torch.empty(2, dtype=torch.float32, device=dev)
    lib = capi.load()
    buf = (ctypes.c_ulonglong * 32)()
    have_counts = hasattr(lib, "vgt_hip_debug_sweep_stats")
    have_timing = hasattr(lib, "vgt_hip_debug_sweep_items")
    if have_counts:
        lib.vgt_hip_debug_sweep_stats(buf, 1)
    bins = (ctypes.c_ulonglong * 128)()
    exits = (ctypes.c_ulonglong * 8)()
    if have_timing:
        ctx.sdf_dev(occ.data_ptr(), shape, 0.01, sdf.data_ptr(), ws.data_ptr(), nbytes, mm.data_ptr(), True, False)
        lib.vgt_hip_debug_sweep_items(bins, exits, 1)
    ctx.sdf_dev(occ.data_ptr(), shape, 0.01, sdf.data_ptr(), ws.data_ptr(), nbytes, mm.data_ptr(), True, False)
    torch.cuda.synchronize()
    if have_counts:
        lib.vgt_hip_debug_sweep_stats(buf, 1)
    if have_timing:
        lib.vgt_hip_debug_sweep_items(bins, exits, 1)
    for p, name in ((0, "Y"), (1, "X")):
        first, last, total, waves = [exits[p * 4 + i] for i in range(4)]
        if waves:
            span = (last - first) / 100.0
            mean_exit = (total / waves - first) / 100.0
            print("%s pass: %d waves, first item start to last exit %.1f us, mean exit at %.1f us (%.1f %% of the span idle at the end)"
                  % (name, waves, span, mean_exit, 100.0 * (span - mean_exit) / span))
        for c in (0, 1):
            row = [bins[(p * 2 + c) * 32 + b] for b in range(32)]
            if sum(row):
                print("   item durations, %s class changes, 50-us bins from 0: %s" % ("with" if c else "no", " ".join(str(v) for v in row)))
    rows = size * size * size / 64.0
    for base, name in ((0, "Y"), (16, "X")) if have_counts else ():
        print(name, "pass: wave-rows", rows)
        for i, label in enumerate(NAMES):
            if label != "-":
                print("   %-34s %12d   per wave-row %.4f" % (label, buf[base + i], buf[base + i] / rows))


if __name__ == "__main__":
    main()
