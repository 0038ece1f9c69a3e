#!/usr/bin/env python3
"""Secondary benchmark: BASELINE config 3 -- 1M-point synthetic cloud -> 256^3 occupancy through
the HIP DDA raycast voxelizer + filter.  Prints one JSON line.

    python bench_raycast.py [--points 1000000] [--grid 256] [--steps 10] [--warmup 3]

Cloud A: sensor at the grid centre; cloud B: same points, sensor outside the grid (slab entry).
Points are resident in HBM when the timed region starts (vgt_hip_raycast_points_f32_dev).
Parity: every tracking count equals the CPU oracle's (float32 restatement) bit for bit.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def atomic_roofline(cloud_key, raycast_ms, visits, distinct_cells, points):
    """The raycaster against its bounds.  SURVEY 8d prices the path at 12 B per point + 8 B per visit against HBM; what the
    kernel actually sends to memory is one atomic per distinct cell a workgroup's table holds at a flush, so the second
    yardstick is the L2's atomic rate: atomic requests per launch (committed rocprofv3 --pmc TCC_ATOMIC pass,
    profiles/raycast_atomic_current.json, reported only for the sources it was collected on and while the call timed here
    is within 10 % of the duration it was collected at) over the rates of the committed micro-benchmark
    (tools/microbench/scattered_atomics.hip): uniformly random cells, and one 256-byte line per wave instruction.  The
    table is flushed in address order, so the kernel is not held to the random rate; since round 5 it is the walk's
    instruction issue that bounds it (profiles/r5/experiments.md, "Raycaster").  Counters cannot be read from inside
    this process."""
    algorithmic = 12.0 * points + 8.0 * visits
    out = {"visits": int(visits), "distinct_cells_touched": int(distinct_cells),
           "hbm": {"algorithmic_bytes": int(algorithmic), "achieved_GBps": round(algorithmic / (raycast_ms * 1e-3) / 1e9, 1),
                   "peak_GBps": 8000.0, "frac": round(algorithmic / (raycast_ms * 1e-3) / 8e12, 4),
                   "note": "SURVEY 8d: 12 B/point + 8 B/visit; the time is the whole call (direction sort + walk)"},
           "note": "floor of the atomics = one per distinct (cell, counter) touched; the workgroups' LDS tables merge visits of "
                   "the SAME workgroup between two flushes only"}
    path = os.path.join(ROOT, "profiles", "raycast_atomic_current.json")
    try:
        with open(path) as fh:
            doc = json.load(fh)
        from voxelized_geometry_tools_amd import synthetic
        if doc.get("sources_sha256") != synthetic.kernel_sources_sha256("voxelizer"):
            out["counters_source"] = ("profiles/raycast_atomic_current.json was collected on another voxelizer_kernels.hip "
                                      "(commit %s): collect it again" % doc.get("commit", "?"))
            return out
        entry = doc["clouds"][cloud_key]
        rate = float(doc["scattered_atomic_rate_G_per_s"])
        recorded_ms = float(entry["raycast_kernel_ms"])
        out["scattered_atomic_rate_G_per_s"] = rate
        out["one_line_per_wave_atomic_rate_G_per_s"] = (doc.get("microbench") or {}).get("one_line_G_atomics_per_s")
        out["rate_source"] = doc.get("rate_source")
        if recorded_ms > 0 and abs(raycast_ms - recorded_ms) <= 0.10 * recorded_ms:
            atomics = float(entry["l2_atomics_per_launch"])
            out.update({"l2_atomics_per_launch": int(atomics), "atomics_per_visit": round(atomics / max(visits, 1), 4),
                        "atomics_per_distinct_cell": round(atomics / max(distinct_cells, 1), 3),
                        "achieved_G_atomics_per_s": round(atomics / (raycast_ms * 1e-3) / 1e9, 2),
                        "frac_of_scattered_atomic_rate": round(atomics / (raycast_ms * 1e-3) / 1e9 / rate, 3),
                        "counters_source": "profiles/raycast_atomic_current.json (commit %s, call %.3f ms)" % (
                            doc.get("commit", "?"), recorded_ms)})
        else:
            out["counters_source"] = "profiles/raycast_atomic_current.json is for a %.3f ms call (commit %s): not this build" % (
                recorded_ms, doc.get("commit", "?"))
    except (OSError, KeyError, ValueError, TypeError):
        out["counters_source"] = None
    return out


def measure(points=1_000_000, grid=256, steps=10, warmup=3, check=True, threads_per_block=-1, ctx=None):
    """-> {"A_inside": {...}, "B_outside": {...}}: per cloud the raycast time (points resident in HBM), visits, rates and,
    with `check`, whether every tracking count equals the CPU oracle's."""
    import torch
    from voxelized_geometry_tools_amd import capi, synthetic

    n = grid
    counts = (n, n, n)
    vs = np.float32(5.12 / n)
    ivs = np.float32(1.0) / vs
    sizes = [np.float32(c) * vs for c in counts]
    pts = synthetic.raycast_cloud(points, seed=42)
    env = np.zeros(counts, dtype=np.float32)
    env[:, :, 0] = 1.0
    dev = torch.device("cuda", torch.cuda.current_device())
    pts_dev = torch.from_numpy(pts).to(dev)
    own_ctx = ctx is None
    if own_ctx:
        ctx = capi.Context(0, threads_per_block)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    out = {}
    for name, sensor in (("A_inside", (2.56, 2.56, 2.56)), ("B_outside", (-1.0, 2.56, 2.56))):
        xf = synthetic.translation_xform(*sensor).astype(np.float32)
        grids = ctx.tracking_grids(n ** 3, 1)

        def step():
            grids.raycast_f32_dev(0, pts_dev.data_ptr(), points, 3.0, xf, vs, ivs, sizes, counts)

        for _ in range(warmup):
            step()
        grids.clear()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        step()
        ev1.record()
        torch.cuda.synchronize()
        single_ms = ev0.elapsed_time(ev1)
        got = grids.retrieve(0, counts)
        visits = int(got.sum())
        distinct = int(np.count_nonzero(got))
        grids.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        fg = ctx.filter_grid(env)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fg.filter(grids, 1.0, 1, 1)
        ctx.synchronize()
        filter_ms = (time.perf_counter() - t0) * 1e3
        entry = {"raycast_ms": round(ms, 4), "first_call_ms": round(single_ms, 4),
                 "Mpoints_per_s": round(points / ms / 1e3, 1),
                 "Mvisits_per_s": round(visits / ms / 1e3, 1), "visits": visits,
                 "filter_ms": round(filter_ms, 4),
                 # algorithmic bytes: 12 B/point + 8 B/visit (4-byte atomic RMW), SURVEY.md 8d
                 "achieved_GBps": round((12.0 * points + 8.0 * visits) / (ms * 1e-3) / 1e9, 1)}
        if points == 1_000_000 and grid == 256:
            entry["atomic_roofline"] = atomic_roofline(name, ms, visits, distinct, points)
        if check:
            from oracle import oracle as O
            t0 = time.perf_counter()
            want = O.raycast_f32(pts, 3.0, xf, vs, ivs, sizes, counts)
            cpu_s = time.perf_counter() - t0
            entry["counts_bit_exact"] = bool(np.array_equal(got, want))
            entry["cpu_oracle_Mpoints_per_s"] = round(points / cpu_s / 1e6, 2)
            entry["cpu_threads"] = O.max_threads()
        out[name] = entry
        grids.close()
    if own_ctx:
        ctx.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--threads-per-block", type=int, default=-1, help="HIP_THREADS_PER_BLOCK of the helper (-1 = default)")
    args = ap.parse_args()
    out = measure(args.points, args.grid, args.steps, args.warmup, not args.no_check, args.threads_per_block)
    print(json.dumps({"metric": "Mpoints/s, 1M-point cloud -> %d^3 occupancy, HIP DDA raycast" % args.grid,
                      "config": {"workload": "BASELINE config 3", "points": args.points, "grid": args.grid},
                      "results": out}))


if __name__ == "__main__":
    main()
