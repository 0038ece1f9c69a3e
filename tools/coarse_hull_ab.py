#!/usr/bin/env python3
"""Coarse hull in front of the X sweep (csrc/edt_sweep_kernels.hip, kCoarse) on and off, interleaved on one box: per-kernel ms of
device-resident extractions, testing library.  Usage: python tools/coarse_hull_ab.py [nx,ny,nz ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from voxelized_geometry_tools_amd import capi


def main():
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(1024, 1024, 1024), (512, 512, 512), (256, 256, 256)]
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0, testing=True)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for shape in shapes:
        for dist in ("spheres", "salt"):
            occ = bench.device_occupancy(torch, shape, dist, 42, dev)
            sdf = torch.empty(shape, dtype=torch.float32, device=dev)
            nbytes = capi.sdf_workspace_bytes(shape)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            rows = {True: [], False: []}
            ref = None
            for rep in range(3):
                for on in (True, False):
                    ctx.set_sweep_coarse_hull(on)
                    for _ in range(2):
                        ctx.sdf_dev(occ.data_ptr(), shape, 0.01, sdf.data_ptr(), ws.data_ptr(), nbytes, None)
                    torch.cuda.synchronize()
                    ctx.timing_start(8)
                    for _ in range(8):
                        ctx.sdf_dev(occ.data_ptr(), shape, 0.01, sdf.data_ptr(), ws.data_ptr(), nbytes, None)
                    torch.cuda.synchronize()
                    rows[on].append(ctx.timing_stop().mean(axis=0))
                    if ref is None:
                        ref = sdf.clone()
                    else:
                        assert os.environ.get("VGT_AB_NO_CHECK") or torch.equal(ref.view(torch.int32), sdf.view(torch.int32))
            ctx.set_sweep_coarse_hull(False)
            vox = float(np.prod(shape))
            for on in (True, False):
                k = np.mean(rows[on], axis=0)
                print("%-16s %-8s coarse hull %-3s  p1 %.4f  Y %.4f  X %.4f  sum %.4f ms  whole-SDF frac %.3f" % (
                    "x".join(map(str, shape)), dist, "on" if on else "off", k[0], k[1], k[2], k.sum(),
                    24 * vox / (k.sum() * 1e-3) / 8e12))
            del occ, sdf, ws, ref


if __name__ == "__main__":
    main()
