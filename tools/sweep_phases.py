#!/usr/bin/env python3
"""Where a wave's time goes in the sweep passes, from a diagnostic build:
   make -C voxelized_geometry_tools_amd/csrc product OBJDIR=phases OUT=../libvgt_hip_phases.so HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -DVGT_SWEEP_PHASES"
   VGT_HIP_LIB=.../libvgt_hip_phases.so python tools/sweep_phases.py [size] [dist]
Ticks of s_memtime, summed over the waves (lane 0 adds once per item), as shares of the items' time; the marks cost the X pass
6 % and disturb the records Y pass far more (its figures are not to be used)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from voxelized_geometry_tools_amd import capi

NAMES = ["sweep 1: waiting for the band's rows (loads drained at the top of a band)", "sweep 1: ring checks (spills, refills)",
         "sweep 1: the rows", "sweep 2: refill steps", "sweep 2: the rows (refill steps included)", "items, whole",
         "the first band's loads", "the coarse hull's first sweep (X pass, kCoarse)"]


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
    for _ in range(2):
        ctx.sdf_dev(occ.data_ptr(), shape, 0.01, sdf.data_ptr(), ws.data_ptr(), nbytes, mm.data_ptr(), True, False)
    lib.vgt_hip_debug_sweep_stats(buf, 1)
    ctx.timing_start(1)
    ctx.sdf_dev(occ.data_ptr(), shape, 0.01, sdf.data_ptr(), ws.data_ptr(), nbytes, mm.data_ptr(), True, False)
    torch.cuda.synchronize()
    print("kernel ms (pass 1, Y, X) of this build:", ctx.timing_stop())
    lib.vgt_hip_debug_sweep_stats(buf, 1)
    for base, name in ((0, "Y"), (16, "X")):
        total = buf[base + 5]
        print(name, "pass: items' time summed over the waves: %d ticks" % total)
        for i, label in enumerate(NAMES):
            if label != "-" and total:
                print("   %-76s %14d  %5.1f %%" % (label, buf[base + i], 100.0 * buf[base + i] / total))


if __name__ == "__main__":
    main()
