#!/usr/bin/env python3
"""Times MakeAllObjectSDFs through the host-pointer batch entry point (vgt_hip_cells_object_sdfs) against one call per
object, and the box's page-locked D2H rate next to it.  Usage: python tools/bench_object_sdfs.py [edge] [objects]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from voxelized_geometry_tools_amd import capi


def main():
    edge = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    objects = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    shape = (edge, edge, edge)
    rng = np.random.default_rng(42)
    rec = np.zeros(shape, dtype=capi.TAGGED_OBJECT_CELL)
    for oid in range(1, objects + 1):
        lo = [int(rng.integers(0, s - 24)) for s in shape]
        ext = [int(rng.integers(4, 24)) for _ in shape]
        box = tuple(slice(a, a + e) for a, e in zip(lo, ext))
        rec["occupancy"][box] = 1.0
        rec["object_id"][box] = oid
    with capi.Context(0) as ctx:
        cells = ctx.cells(rec, shape)
        ids = cells.object_ids()

        def timed(fn, repeat=5):
            fn()
            times = []
            for _ in range(repeat):
                t0 = time.perf_counter()
                result = fn()
                times.append((time.perf_counter() - t0) * 1e3)
            return min(times), sum(times) / len(times), result

        batched_ms, batched_mean_ms, batched = timed(lambda: cells.separate_object_sdfs(0.01, ids), repeat=8)
        looped_ms, _, looped = timed(lambda: cells.separate_object_sdfs_one_by_one(0.01, ids), repeat=2)
        same = all(np.array_equal(batched[int(i)][0].view(np.uint32), looped[int(i)][0].view(np.uint32)) for i in ids)
        cells.close()
    # the link: a page-locked buffer of the same size, device to host
    nbytes = len(ids) * int(np.prod(shape)) * 4
    dev = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
    pinned = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    pinned.copy_(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        pinned.copy_(dev, non_blocking=True)
    torch.cuda.synchronize()
    d2h_ms = (time.perf_counter() - t0) / 5 * 1e3
    print(json.dumps({"shape": shape, "objects": len(ids), "batched_ms": round(batched_ms, 3), "batched_mean_ms": round(batched_mean_ms, 3),
                      "one_call_per_object_ms": round(looped_ms, 3), "bit_equal": bool(same),
                      "bytes": nbytes, "pinned_d2h_ms": round(d2h_ms, 3),
                      "pinned_d2h_GBps": round(nbytes / d2h_ms / 1e6, 1)}))


if __name__ == "__main__":
    main()
