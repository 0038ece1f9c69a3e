#!/usr/bin/env python3
"""Small maps (the reference's own sizes, example/tutorial.cpp: 40^3): ms per blocking call through the host-pointer entry
point and per device-resident extraction.  Usage: python tools/bench_small_maps.py [edge ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from voxelized_geometry_tools_amd import capi, synthetic


def main():
    out = {}
    with capi.Context(0) as ctx:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        for edge in ([int(a) for a in sys.argv[1:]] or (16, 40, 64)):
            shape = (edge,) * 3
            occ = np.ascontiguousarray(synthetic.occupancy_spheres(shape, seed=42))
            out_host = np.empty_like(occ)
            for _ in range(5):
                ctx.sdf_from_occupancy(occ, 0.01, out=out_host)
            t0 = time.perf_counter()
            for _ in range(200):
                ctx.sdf_from_occupancy(occ, 0.01, out=out_host)
            host_ms = (time.perf_counter() - t0) / 200 * 1e3
            occ_dev = torch.from_numpy(occ).cuda()
            sdf = torch.empty(shape, dtype=torch.float32, device="cuda:0")
            nbytes = capi.sdf_workspace_bytes(shape)
            ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
            mm = torch.empty(2, dtype=torch.float32, device="cuda:0")
            run = lambda: ctx.sdf_dev(occ_dev.data_ptr(), shape, 0.01, sdf.data_ptr(), ws.data_ptr(), nbytes, mm.data_ptr(), True, False)
            for _ in range(10):
                run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                run()
            torch.cuda.synchronize()
            stream_ms = (time.perf_counter() - t0) / 200 * 1e3
            same = np.array_equal(sdf.cpu().numpy().view(np.uint32), out_host.view(np.uint32))
            out["%d^3" % edge] = {"host_entry_point_ms_per_blocking_call": round(host_ms, 4),
                                  "device_resident_ms_in_a_stream": round(stream_ms, 4), "host_equals_device_resident": bool(same)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
