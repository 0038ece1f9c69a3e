#!/usr/bin/env python3
"""Headline benchmark: Mvoxels/s of a float SDF extraction (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 1024] [--dist spheres]

One "step" = one full ExtractSignedDistanceField<float> pass (pass 1 to class records, Y pass, X pass +
finalize + min/max) over a synthetic occupancy grid that is already resident in HBM; the
SDF is left in HBM.  Workloads (--workload): c4 = BASELINE config 4 (1024^3, distribution D1
"spheres", seed 42, resolution 0.01), the headline and the default at N = 1; c5 = BASELINE
config 5 (2048 x 2048 x 1024), the default at N > 1, where the grid is partitioned into Z slabs,
one process per GPU, with one RCCL exchange of 4-byte per-line boundary summaries
(voxelized_geometry_tools_amd/multi_gpu.py).  `--gpus 1 --workload c5` runs config 5 on one GPU
(the reference point of the scaling series: an N > 1 line names it in `same_workload_one_gpu`).
Rank 0 prints ONE JSON line.

Launching: `python bench.py --gpus N` starts the N ranks itself (a `torch.distributed.run`
child, started before this process touches the GPU) unless it already runs under a launcher
(WORLD_SIZE set), in which case WORLD_SIZE must equal N -- a mismatch is an error, never a
silent 1-GPU run.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ALG_BYTES_PER_VOXEL_PASS = 8.0  # 3 passes x (4 B read + 4 B write) = 24 B/voxel
KERNEL_NAMES = ["PassZClassRecords", "PassY", "PassXFinalize"]  # (variant 1: pass 1 is the int16 Z scan)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="auto", choices=["auto", "c4", "c5"],
                    help="c4 = 1024^3 (BASELINE config 4), c5 = 2048x2048x1024 (config 5); auto: c4 at N = 1, c5 at N > 1")
    ap.add_argument("--size", type=int, default=0, help="cube edge instead of the workload's shape (experiments)")
    ap.add_argument("--dist", default="spheres", choices=["spheres", "salt", "unknown_mix", "empty", "single"])
    ap.add_argument("--salt-p", type=float, default=0.01, help="fill probability of --dist salt")
    ap.add_argument("--variant", type=int, default=0, help="EDT variant (0 default; 1 = the testing library's cross-check pipeline: int16 Z scan + pruned search)")
    ap.add_argument("--force-slab", action="store_true",
                    help="run the Z-slab (multi-GPU) code path even with one rank: NCCL init, summary all-gather, "
                         "fix-up kernel, extrema all-reduce (smoke test of the N > 1 path on a single GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-pointer (PCIe-inclusive) measurement")
    ap.add_argument("--no-raycast", action="store_true", help="skip the raycast voxelizer section (BASELINE config 3)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary SDF workloads (salt, unknown mix, 512^3, 64^3, batches)")
    ap.add_argument("--cpu-seconds", type=float, default=30.0)
    return ap.parse_args()


def device_occupancy(torch, shape, dist, seed, device, z_offset=0, full_shape=None, salt_p=0.01):
    """Synthetic occupancy built directly in HBM; identical to synthetic.make_occupancy."""
    from voxelized_geometry_tools_amd import synthetic
    full_shape = full_shape or shape
    nx, ny, nz = shape
    occ = torch.zeros(shape, dtype=torch.float32, device=device)
    if dist in ("spheres", "unknown_mix"):
        centres, r2 = synthetic.sphere_list(full_shape, seed)
        for (cx, cy, cz), rr in zip(centres, r2):
            cz = int(cz) - z_offset
            r = int(np.ceil(np.sqrt(rr)))
            x0, x1 = max(int(cx) - r, 0), min(int(cx) + r + 1, nx)
            y0, y1 = max(int(cy) - r, 0), min(int(cy) + r + 1, ny)
            z0, z1 = max(cz - r, 0), min(cz + r + 1, nz)
            if x0 >= x1 or y0 >= y1 or z0 >= z1:
                continue
            dx = (torch.arange(x0, x1, device=device, dtype=torch.int64) - int(cx)) ** 2
            dy = (torch.arange(y0, y1, device=device, dtype=torch.int64) - int(cy)) ** 2
            dz = (torch.arange(z0, z1, device=device, dtype=torch.int64) - cz) ** 2
            d2 = dx[:, None, None] + dy[None, :, None] + dz[None, None, :]
            occ[x0:x1, y0:y1, z0:z1][d2.to(torch.float64) <= float(rr)] = 1.0
        if dist == "unknown_mix":
            g = torch.Generator(device=device)
            g.manual_seed(seed + 1)
            occ[torch.rand(shape, device=device, generator=g) < 0.01] = 0.5
    elif dist == "salt":
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        occ = (torch.rand(shape, device=device, generator=g) < salt_p).to(torch.float32)
    elif dist == "single":
        if z_offset == 0:
            occ[0, 0, 0] = 1.0
    return occ


def profiled_traffic(kernel, default_workload, timed_kernel_ms):
    """HBM bytes per launch of `kernel` from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE, tools/collect_profiles.sh): counters cannot be read from inside this process.  The file
    names the commit, a hash of the EDT sources, the kernel and the kernel's duration it was collected with; the value is
    only reported when this tree's EDT sources hash to the same value AND the kernel timed in THIS run is within 10 % of
    that duration (else null)."""
    if not default_workload:
        return None, None
    path = os.path.join(ROOT, "profiles", "pmc_hbm_traffic_current.json")
    try:
        with open(path) as fh:
            doc = json.load(fh)
        from voxelized_geometry_tools_amd import synthetic
        if doc.get("sources_sha256") != synthetic.kernel_sources_sha256("edt"):
            return None, ("profiles/pmc_hbm_traffic_current.json was collected on other EDT sources (commit %s): "
                          "collect it again (tools/collect_profiles.sh)" % doc.get("commit", "?"))
        entry = doc["kernels"][kernel]
        recorded_ms = float(entry["kernel_ns"]) * 1e-6
        # (the hash says it is this build; the duration only guards against a box at another clock / memory level, and the
        # profiler's own run is a few per cent slower than an unprofiled one)
        if recorded_ms <= 0 or abs(timed_kernel_ms - recorded_ms) > 0.10 * recorded_ms:
            return None, "profiles/pmc_hbm_traffic_current.json is for a %.3f ms kernel (commit %s): not this build" % (
                recorded_ms, doc.get("commit", "?"))
        return round(entry["hbm_bytes"] / 1e9, 3), (
            "profiles/pmc_hbm_traffic_current.json (GB per launch; commit %s, %s, %.3f ms)" % (
                doc.get("commit", "?"), entry.get("kernel_name", kernel)[:60], recorded_ms))
    except (OSError, KeyError, ValueError, TypeError):
        return None, None


def one_gpu_reference(workload_key):
    """ms per step of the same workload on ONE GPU, from the committed bench line of that run (the scaling series'
    reference point: the N = 1 default of this script is the c4 headline, not c5)."""
    path = os.path.join(ROOT, "profiles", "bench_%s_one_gpu_current.json" % workload_key)
    try:
        with open(path) as fh:
            doc = json.load(fh)
        return {"ms_per_step": doc["ms_per_step"], "source": "profiles/bench_%s_one_gpu_current.json" % workload_key,
                "commit": doc.get("commit")}
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(budget_s):
    """Times the CPU oracle (a port of the reference's OpenMP CPU path, NOT the reference
    binary -- unbuildable here) on a bounded sample of the same distribution."""
    import tempfile
    from oracle import oracle as O
    from voxelized_geometry_tools_amd import synthetic
    lib = None
    try:  # rebuild with -march=native for this host, like the reference's CMake does
        out = os.path.join(tempfile.mkdtemp(prefix="vgt_oracle_"), "libvgt_oracle_native.so")
        lib = O.load(O.build(march="native", out=out))
    except Exception:
        lib = O.load()
    cores = int(lib.vgt_oracle_max_threads())

    def mem_ok(edge):  # the oracle holds two double fields + float in/out: ~28 B/voxel with slack
        try:
            with open("/proc/meminfo") as fh:
                avail_kb = next(int(l.split()[1]) for l in fh if l.startswith("MemAvailable"))
        except (OSError, StopIteration):
            return edge <= 512
        return 32.0 * edge ** 3 < 0.6 * avail_kb * 1024.0

    # grow the sample towards the headline size (1024^3) while the extrapolated time fits the budget
    edge, rate, dt = 128, None, 0.0
    while True:
        occ = synthetic.occupancy_spheres((edge,) * 3, 42)
        t0 = time.perf_counter()
        O.sdf_from_occupancy(occ, 0.01, True, False, 0, lib=lib)
        dt = time.perf_counter() - t0
        rate = occ.size / dt / 1e6
        del occ
        if edge >= 1024:
            break
        nxt = min(edge * 2, 1024)
        # EDT cost is ~linear in voxels; stop once the next size would blow the budget or the host RAM
        if dt * (nxt / edge) ** 3 > budget_s or not mem_ok(nxt):
            break
        edge = nxt
    return {"value": round(rate, 3), "unit": "Mvoxels/s", "cores": cores, "kind": "port",
            "sample": "%d^3 D1 spheres seed 42, res 0.01, %.2f s on %d OpenMP threads "
                      "(oracle/vgt_oracle.c, -O3 -march=native)%s" % (
                          edge, dt, cores, "" if edge == 1024 else
                          "; 1024^3 did not fit the %.0f s budget / host RAM on this box" % budget_s)}


def end_to_end(ctx, torch, occ_dev, shape, res):
    """SURVEY 8d "second number": the same extraction through the host-pointer entry point of the C ABI
    (vgt_hip_sdf_from_occupancy_f32: H2D + three passes + D2H), once from ordinary pageable host memory (what a
    caller of the reference holds; the library page-locks it for the call) and once from pinned host memory,
    next to the PCIe rates of plain pinned copies measured here."""
    nbytes = int(np.prod(shape)) * 4
    occ_pinned = torch.empty(shape, dtype=torch.float32, pin_memory=True)
    out_pinned = torch.empty(shape, dtype=torch.float32, pin_memory=True)
    occ_pinned.copy_(occ_dev)
    torch.cuda.synchronize()
    scratch = torch.empty(shape, dtype=torch.float32, device=occ_dev.device)

    def timed(fn, repeat=2):
        best = None
        for _ in range(repeat):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best

    h2d = timed(lambda: scratch.copy_(occ_pinned, non_blocking=True))
    d2h = timed(lambda: out_pinned.copy_(scratch, non_blocking=True))
    del scratch
    occ_np = occ_pinned.numpy()
    out_np = out_pinned.numpy()
    # cold: what a first-time caller pays -- fresh pageable arrays (the output's pages never touched), a context
    # without cached device buffers: page-locking both arrays, hipMalloc of in / out / workspace, then the pipeline
    ctx.trim()
    occ_cold = np.array(occ_np, copy=True)
    out_cold = np.empty(shape, dtype=np.float32)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.sdf_from_occupancy(occ_cold, res, out=out_cold)
    cold = time.perf_counter() - t0
    del occ_cold, out_cold
    # page-locking alone: hipHostRegister + hipHostUnregister of one fresh pageable array of the grid's size
    register_ms = None
    try:
        probe = np.empty(shape, dtype=np.float32)
        probe[...] = 0.0
        rt = torch.cuda.cudart()
        t0 = time.perf_counter()
        rc = rt.cudaHostRegister(probe.ctypes.data, probe.nbytes, 0)
        t1 = time.perf_counter()
        if int(rc) == 0:
            rt.cudaHostUnregister(probe.ctypes.data)
            register_ms = round((t1 - t0) * 1e3, 2)
        del probe
    except Exception:
        register_ms = None
    pinned = timed(lambda: ctx.sdf_from_occupancy(occ_np, res, out=out_np))
    occ_pageable = np.array(occ_np, copy=True)
    out_pageable = np.zeros(shape, dtype=np.float32)         # zeros: pages touched before the timed call
    ctx.sdf_from_occupancy(occ_pageable, res, out=out_pageable)
    pageable = timed(lambda: ctx.sdf_from_occupancy(occ_pageable, res, out=out_pageable))
    same = bool(np.array_equal(out_pageable.view(np.uint32), out_np.view(np.uint32)))
    ctx.trim()
    bound = h2d + d2h
    return {"end_to_end_ms": round(pageable * 1e3, 2), "end_to_end_pinned_ms": round(pinned * 1e3, 2),
            "end_to_end_cold_ms": round(cold * 1e3, 2), "register_ms": register_ms,
            "pcie_lower_bound_ms": round(bound * 1e3, 2),
            "pcie_GBps": {"h2d": round(nbytes / h2d / 1e9, 1), "d2h": round(nbytes / d2h / 1e9, 1),
                          "end_to_end": round(2 * nbytes / pageable / 1e9, 1),
                          "end_to_end_pinned": round(2 * nbytes / pinned / 1e9, 1)},
            "vs_pcie_lower_bound": round(pinned / bound, 3), "host_results_identical": same,
            "note": "host-pointer entry point: H2D, pass 1 + Y pass per X chunk as it arrives, X pass per Y range, D2H "
                    "per range (three streams).  end_to_end_ms: warm call from pageable host memory (the library "
                    "page-locks the two arrays per call; after one earlier call on the same arrays the pages are "
                    "resident and the device buffers cached); end_to_end_pinned_ms: from pinned memory; "
                    "end_to_end_cold_ms: FIRST call on fresh pageable arrays with no cached device buffers (page "
                    "faults of the untouched output, page-locking, hipMalloc); register_ms: hipHostRegister of one "
                    "fresh array of the grid's size alone"}


def raycast_section():
    """BASELINE config 3 in the driver-run line: the raycast kernel on clouds A and B (bench_raycast.measure: points
    resident in HBM, counts checked bit for bit against the CPU oracle) and the voxelizer end to end through the C++
    host layer (HipPointCloudVoxelizer::VoxelizePointClouds: upload of points and static grid, raycast, filter,
    download) for 1, 2 and 8 concurrent clouds, measured by tests/cpp/bench_voxelize (a child process)."""
    import subprocess
    import bench_raycast
    out = {"workload": "BASELINE config 3: 1M-point cloud -> 256^3 grid, max_range 3.0, every 100th point NaN",
           "bytes_convention": "12 B/point + 8 B/visit (SURVEY 8d)"}
    out.update(bench_raycast.measure(steps=5, warmup=2, check=True))
    binary = os.path.join(ROOT, "tests", "cpp", "bench_voxelize")
    if os.path.exists(binary):
        try:
            run = subprocess.run([binary], capture_output=True, text=True, timeout=120)
            out["voxelize_end_to_end"] = json.loads(run.stdout.strip().splitlines()[-1]) if run.returncode == 0 else {
                "error": (run.stdout + run.stderr)[-400:]}
        except Exception as exc:
            out["voxelize_end_to_end"] = {"error": repr(exc)}
    else:
        out["voxelize_end_to_end"] = {"error": "tests/cpp/bench_voxelize not built (make -C tests/cpp bench_voxelize)"}
    return out


def cpp_host_layer():
    """The SDF entry points of the C++ host layer as a caller of the reference's interface sees them -- a map in host
    memory in, a NEW SignedDistanceField out per call (tests/cpp/bench_sdf_host, a child process): the 1024^3 map, a
    40^3 map, 64 maps of 64^3 as one batch, MakeAllObjectSDFs, single tagged / component maps, the two-slab entry point."""
    import subprocess
    binary = os.path.join(ROOT, "tests", "cpp", "bench_sdf_host")
    if not os.path.exists(binary):
        return {"error": "tests/cpp/bench_sdf_host not built (make -C tests/cpp bench_sdf_host)"}
    try:
        run = subprocess.run([binary, "1024", "3", "slabs"], capture_output=True, text=True, timeout=300)
        if run.returncode != 0:
            return {"error": (run.stdout + run.stderr)[-400:]}
        out = json.loads(run.stdout.strip().splitlines()[-1])
        out["note"] = ("ms per blocking call, result grids freshly allocated by every call (best / mean over the calls; "
                       "first = the process's first call with context set-up and first page-locking of the input)")
        return out
    except Exception as exc:
        return {"error": repr(exc)}


def multi_host_path(torch, shape, res):
    """vgt_hipx_sdf_multi (one process, host arrays in and out, one Z slab per listed device) with ALL EIGHT slabs on
    this one GPU: the data movement of the 8-GPU call -- rows of nz / 8 floats at a pitch of nz (512 B at nz = 1024) in
    both directions -- against the PCIe time of plain pinned copies of the same bytes."""
    from voxelized_geometry_tools_amd import capi, synthetic
    occ = synthetic.occupancy_spheres(shape, 42)
    out = np.zeros(shape, dtype=np.float32)
    devices = [torch.cuda.current_device()] * 8
    t0 = time.perf_counter()
    capi.sdf_multi(devices, occ, res, out=out)
    first = time.perf_counter() - t0
    first_phases = capi.sdf_multi_last_timing()
    best, phases = None, None
    for _ in range(2):
        t0 = time.perf_counter()
        capi.sdf_multi(devices, occ, res, out=out)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, phases = dt, capi.sdf_multi_last_timing()
    capi.sdf_multi_release()
    nbytes = occ.nbytes
    return {"shape": list(shape), "slabs": 8, "row_bytes": int(shape[2] // 8 * 4), "call_ms": round(best * 1e3, 2),
            "first_call_ms": round(first * 1e3, 2), "first_call_setup_ms": round(first_phases["setup_ms"], 2),
            "phases_ms": {k: round(v, 2) for k, v in phases.items()},
            "GBps": {"upload": round(nbytes / (phases["upload_ms"] * 1e-3) / 1e9, 1) if phases["upload_ms"] > 0 else None,
                     "download": round(nbytes / (phases["download_ms"] * 1e-3) / 1e9, 1) if phases["download_ms"] > 0 else None},
            "note": "eight slabs on ONE device (devices = [d] * 8): the slabs share the device's copy engines and CUs, so "
                    "upload_ms / download_ms are the slowest slab's strided copies while the others run; call_ms is the "
                    "whole call from pageable host arrays (page-locked per call), device state kept from the first call"}


def secondary_workloads(ctx, torch, capi, device, res, steps=3, warmup=1):
    """SURVEY 8d's secondary numbers in the driver-run line: the other distributions at the headline size and D1 at the
    smaller sizes, device-resident like the headline, `steps` timed steps each OUTSIDE the headline's timed region.
    ms = wall clock per step around the timed steps (fence on both sides); kernel_ms / whole_sdf_frac from the library's
    kernel events (24 B/voxel credited, as for the headline)."""
    out = {}
    cases = [("1024^3 D2 salt p=0.01", (1024,) * 3, "salt"), ("1024^3 D3 unknown mix", (1024,) * 3, "unknown_mix"),
             ("512^3 D1 spheres", (512,) * 3, "spheres"), ("64^3 D1 spheres", (64,) * 3, "spheres")]
    for name, shape, dist_name in cases:
        try:
            occ = device_occupancy(torch, shape, dist_name, 42, device)
            sdf = torch.empty(shape, dtype=torch.float32, device=device)
            ws_bytes = capi.sdf_workspace_bytes(shape, 0)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
            minmax = torch.zeros(2, dtype=torch.float32, device=device)

            def step():
                ctx.sdf_dev(occ.data_ptr(), shape, res, sdf.data_ptr(), ws.data_ptr(), ws_bytes, minmax.data_ptr())

            for _ in range(warmup):
                step()
            torch.cuda.synchronize()
            ctx.timing_start(steps)
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            wall_ms = (time.perf_counter() - t0) / steps * 1e3
            per_step = ctx.timing_stop().astype(np.float64)
            avg = per_step.mean(axis=0) if len(per_step) else np.zeros(3)
            vox = float(np.prod(shape))
            dom = int(np.argmax(avg))
            out[name] = {
                "ms": round(wall_ms, 4), "Mvoxels_per_s": round(vox / (wall_ms * 1e-3) / 1e6, 1),
                "kernel_ms": {k: round(float(v), 4) for k, v in zip(KERNEL_NAMES, avg)},
                "whole_sdf_frac": round(3 * ALG_BYTES_PER_VOXEL_PASS * vox / (avg.sum() * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
                if avg.sum() > 0 else None,
                "whole_sdf_frac_wall_clock": round(3 * ALG_BYTES_PER_VOXEL_PASS * vox / (wall_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "dominant_kernel": KERNEL_NAMES[dom],
                "dominant_frac": round(ALG_BYTES_PER_VOXEL_PASS * vox / (avg[dom] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
                if avg[dom] > 0 else None}
            del occ, sdf, ws, minmax
        except Exception as exc:
            out[name] = {"error": repr(exc)}
    out.update(batched_workloads(ctx, torch, capi, device, res))
    return out


def synthetic_occupancy(shape):
    from voxelized_geometry_tools_amd import synthetic
    return synthetic.make_occupancy(shape, "spheres", seed=42)


def batched_workloads(ctx, torch, capi, device, res, steps=5, warmup=2):
    """Many small grids in one call (vgt_hip_sdf_batch_dev) against the same grids one call each, device-resident, and
    the per-object fields of a tagged map (vgt_hip_cells_object_sdfs, host buffers in and out) against one
    vgt_hip_cells_sdf call per object: the reference loops one ExtractSignedDistanceField per map / per object id
    (I/tagged_object_occupancy_map.hpp:249-290)."""
    out = {}
    try:
        shape, batch = (64, 64, 64), 64
        grids = torch.stack([device_occupancy(torch, shape, "spheres", 42 + b, device) for b in range(batch)])
        fields = torch.empty_like(grids)
        ws_bytes = capi.sdf_batch_workspace_bytes(batch, shape)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
        mm = torch.zeros((batch, 2), dtype=torch.float32, device=device)
        one_bytes = capi.sdf_workspace_bytes(shape, 0)

        def batched():
            ctx.sdf_batch_dev(grids.data_ptr(), batch, shape, res, fields.data_ptr(), ws.data_ptr(), ws_bytes, mm.data_ptr())

        def looped():
            for b in range(batch):
                ctx.sdf_dev(grids[b].data_ptr(), shape, res, fields[b].data_ptr(), ws.data_ptr(), one_bytes,
                            mm[b].data_ptr())

        def timed(fn):
            for _ in range(warmup):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / steps * 1e3

        looped_ms = timed(looped)
        reference = fields.clone()
        fields.zero_()
        ctx.timing_start(steps + warmup)
        batched_ms = timed(batched)
        per_step = ctx.timing_stop().astype(np.float64)
        avg = per_step[warmup:].mean(axis=0) if len(per_step) > warmup else np.zeros(3)
        vox = float(np.prod(shape)) * batch
        out["batch 64 x 64^3 D1 spheres"] = {
            "ms": round(batched_ms, 4), "Mvoxels_per_s": round(vox / (batched_ms * 1e-3) / 1e6, 1),
            "kernel_ms": {k: round(float(v), 4) for k, v in zip(KERNEL_NAMES, avg)},
            "whole_sdf_frac": round(3 * ALG_BYTES_PER_VOXEL_PASS * vox / (avg.sum() * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
            if avg.sum() > 0 else None,
            "same_grids_one_call_each_ms": round(looped_ms, 4), "speedup_over_single_calls": round(looped_ms / batched_ms, 1),
            "bit_equal_to_single_calls": bool(torch.equal(fields.view(torch.int32), reference.view(torch.int32)))}
        del grids, fields, ws, mm, reference
    except Exception as exc:
        out["batch 64 x 64^3 D1 spheres"] = {"error": repr(exc)}
    try:
        # the reference's own use: one small map through the host-pointer entry point (example/tutorial.cpp extracts
        # the field of a 40^3 map; its tests use 4 x 8 x 12): upload, three passes, download, blocking
        rows = {}
        for edge in (16, 40, 64):
            occ_host = np.ascontiguousarray(synthetic_occupancy((edge,) * 3))
            out_host = np.empty_like(occ_host)
            for _ in range(3):
                ctx.sdf_from_occupancy(occ_host, res, out=out_host)
            t0 = time.perf_counter()
            for _ in range(20):
                ctx.sdf_from_occupancy(occ_host, res, out=out_host)
            rows["%d^3" % edge] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
        out["one small map through the host entry point (ms per blocking call)"] = rows
    except Exception as exc:
        out["one small map through the host entry point (ms per blocking call)"] = {"error": repr(exc)}
    try:
        shape, objects = (128, 128, 128), 32
        rng = np.random.default_rng(42)
        rec = np.zeros(shape, dtype=capi.TAGGED_OBJECT_CELL)
        for oid in range(1, objects + 1):
            lo = [int(rng.integers(0, s - 24)) for s in shape]
            ext = [int(rng.integers(4, 24)) for _ in shape]
            box = tuple(slice(a, a + e) for a, e in zip(lo, ext))
            rec["occupancy"][box] = 1.0
            rec["object_id"][box] = oid
        cells = ctx.cells(rec, shape)
        ids = cells.object_ids()

        def timed_host(fn, repeat=3):
            fn()
            t0 = time.perf_counter()
            for _ in range(repeat):
                result = fn()
            return (time.perf_counter() - t0) / repeat * 1e3, result

        batched_ms, batched = timed_host(lambda: cells.separate_object_sdfs(res, ids))
        looped_ms, looped = timed_host(lambda: cells.separate_object_sdfs_one_by_one(res, ids))
        # the link's own rate on this box: the same bytes from the device into one page-locked buffer
        nbytes = len(ids) * int(np.prod(shape)) * 4
        dev_buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        pinned = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        pinned.copy_(dev_buf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            pinned.copy_(dev_buf, non_blocking=True)
        torch.cuda.synchronize()
        link_ms = (time.perf_counter() - t0) / 3 * 1e3
        del dev_buf, pinned
        same = all(np.array_equal(batched[int(i)][0].view(np.uint32), looped[int(i)][0].view(np.uint32)) and
                   batched[int(i)][1:] == looped[int(i)][1:] for i in ids)
        out["tagged map 128^3, %d object SDFs (host buffers)" % len(ids)] = {
            "ms": round(batched_ms, 3), "one_call_per_object_ms": round(looped_ms, 3),
            "speedup_over_single_calls": round(looped_ms / batched_ms, 2), "bit_equal_to_single_calls": bool(same),
            "download_bound_ms": round(link_ms, 3), "link_GBps": round(nbytes / link_ms / 1e6, 1),
            "note": "download_bound = the objects' fields (268 MB) from the device into ONE page-locked buffer on this box, "
                    "measured in this run; both paths return freshly allocated host arrays"}
        cells.close()
    except Exception as exc:
        out["tagged map 128^3 object SDFs"] = {"error": repr(exc)}
    return out


def slab_run_check(torch, dist, capi, ctx, runner, sdf, minmax, full_shape, local_shape, z_offset, rank, world, device,
                   dist_name, salt_p, res):
    """N > 1 (and --force-slab): what the line's numbers were measured ON is checked in the run itself, after the timed
    region.  (1) Every rank's summaries arrived intact: rank r's own checksum of what it sent, all-gathered, against the
    checksum of block r of the gathered buffer.  (2) Rank 0 extracts the WHOLE grid alone with the plain one-GPU
    pipeline (when its free memory allows: config 5 needs ~57 GB) and compares its slab of the distributed run -- whose
    values depend on every other rank's summaries -- and the reduced extrema with it, bit for bit."""
    out = {"backend": dist.get_backend(), "world_size_seen_by_the_collectives": dist.get_world_size()}
    own = runner.summary.view(torch.int32).to(torch.int64).sum().reshape(1)
    sums = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(sums, own)
    got = runner.gathered.view(torch.int32).to(torch.int64).reshape(world, -1).sum(dim=1)
    out["summaries_of_all_ranks_intact"] = bool(torch.equal(sums, got))
    if rank == 0:
        need = float(np.prod(full_shape)) * 8 + capi.sdf_workspace_bytes(full_shape, 0) + 4 * 2 ** 30
        free = torch.cuda.mem_get_info()[0]
        if free < need:
            out["rank0_slab_vs_single_gpu"] = "skipped: %.0f GiB free, %.0f needed" % (free / 2 ** 30, need / 2 ** 30)
        else:
            occ_full = device_occupancy(torch, full_shape, dist_name, 42, device, 0, full_shape, salt_p)
            sdf_full = torch.empty(full_shape, dtype=torch.float32, device=device)
            nbytes = capi.sdf_workspace_bytes(full_shape, 0)
            ws_full = torch.empty(nbytes, dtype=torch.uint8, device=device)
            mm_full = torch.zeros(2, dtype=torch.float32, device=device)
            ctx.sdf_dev(occ_full.data_ptr(), full_shape, res, sdf_full.data_ptr(), ws_full.data_ptr(), nbytes, mm_full.data_ptr())
            torch.cuda.synchronize()
            same = True
            for x0 in range(0, full_shape[0], 64):  # (in pieces: the comparison's temporaries stay small)
                a = sdf[x0:x0 + 64].view(torch.int32)
                b = sdf_full[x0:x0 + 64, :, z_offset:z_offset + local_shape[2]].view(torch.int32)
                same = same and bool(torch.equal(a, b))
            out["rank0_slab_vs_single_gpu"] = {"bit_equal": same, "voxels_compared": int(np.prod(local_shape)),
                                               "extrema_equal": bool(torch.equal(minmax, mm_full))}
            del occ_full, sdf_full, ws_full
            torch.cuda.empty_cache()
    dist.barrier()
    return out


def launch_ranks(args):
    """`--gpus N` without a launcher: run this script under torch.distributed.run with N ranks.
    Nothing in this (parent) process has initialised the GPU; the child is a subprocess, not an exec."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import torch
        have = torch.cuda.device_count()  # does not initialise the GPU on this image
        if have < args.gpus:
            raise SystemExit("bench.py --gpus %d: only %d HIP device(s) visible" % (args.gpus, have))
        raise SystemExit(launch_ranks(args))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE=%s: refusing to report a run of a different size"
                         % (args.gpus, os.environ.get("WORLD_SIZE")))
    # stdout carries exactly ONE line, the JSON: everything else that writes to file descriptor 1 (librccl prints a
    # version banner through C stdio, flushed at exit) is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    from voxelized_geometry_tools_amd import capi

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist_on = world > 1 or args.force_slab
    if dist_on:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)

    res = 0.01
    workload_key = args.workload if args.workload != "auto" else ("c5" if dist_on else "c4")
    base_shape = (1024, 1024, 1024) if workload_key == "c4" else (2048, 2048, 1024)
    full_shape = (args.size,) * 3 if args.size else base_shape
    shape_text = "x".join(str(v) for v in full_shape) if len(set(full_shape)) > 1 else "%d^3" % full_shape[0]
    if not dist_on:
        local_shape = full_shape
        z_offset = 0
        workload = "%s float SDF, D1 %s seed 42, res 0.01, device-resident" % (shape_text, args.dist)
        parallelism = "single GPU"
    else:
        from voxelized_geometry_tools_amd import multi_gpu
        local_shape, z_offset = multi_gpu.slab_of(full_shape, rank, world)
        workload = "%s float SDF, D1 %s seed 42, res 0.01, Z-slab x%d, device-resident" % (shape_text, args.dist, world)
        parallelism = "zslab%d" % world

    occ = device_occupancy(torch, local_shape, args.dist, 42, device, z_offset, full_shape, args.salt_p)
    sdf = torch.empty(local_shape, dtype=torch.float32, device=device)
    ws_bytes = capi.sdf_workspace_bytes(local_shape, args.variant)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
    minmax = torch.zeros(2, dtype=torch.float32, device=device)
    # (the cross-check variants exist in the testing library only; the headline runs the product library)
    ctx = capi.Context(local_rank, testing=args.variant != 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_edt_variant(args.variant)
    if dist_on:
        runner = multi_gpu.SlabSdf(ctx, torch, dist, full_shape, rank, world, device)

    phase_events = []  # per timed step: 4 events (start, scan done, exchange done, end)

    def step(record=False):
        if not dist_on:
            ctx.sdf_dev(occ.data_ptr(), local_shape, res, sdf.data_ptr(), ws.data_ptr(), ws_bytes, minmax.data_ptr())
        else:
            events = None
            if record:
                events = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                phase_events.append(events)
            runner.run(occ, sdf, ws, minmax, res, events=events)

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # per-kernel durations: HIP events recorded by the library around each kernel of the timed steps, on
    # the stream the kernels run on, read back once after the timed region (no per-step synchronisation)
    ctx.timing_start(args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(record=True)
    fence()
    elapsed = time.perf_counter() - t0
    per_step_ms = ctx.timing_stop()
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_sum = per_step_ms.astype(np.float64).sum(axis=0) if len(per_step_ms) else np.zeros(3)
    if dist_on:  # the slowest rank's kernels bound the step: report max over ranks
        kt = torch.tensor(kernel_sum, dtype=torch.float64, device=device)
        dist.all_reduce(kt, op=dist.ReduceOp.MAX)
        kernel_sum = kt.cpu().numpy()

    phase_ms = None
    if dist_on and phase_events:
        # whole-step phases from the stream's events; the last interval is split with the library's kernel events
        # (fix-up / Y / X), the remainder being the carry kernel, the extrema all-reduce and launch gaps
        steps = float(len(phase_events))
        scan = sum(e[0].elapsed_time(e[1]) for e in phase_events) / steps
        exchange = sum(e[1].elapsed_time(e[2]) for e in phase_events) / steps
        rest = sum(e[2].elapsed_time(e[3]) for e in phase_events) / steps
        ky, kx = kernel_sum[1] / steps, kernel_sum[2] / steps
        pt = torch.tensor([scan, exchange, max(rest - ky - kx, 0.0), ky, kx], dtype=torch.float64, device=device)
        dist.all_reduce(pt, op=dist.ReduceOp.MAX)
        phase_ms = {k: round(float(v), 4) for k, v in zip(multi_gpu.SlabSdf.PHASES, pt.cpu().numpy())}

    run_check = None
    if dist_on:
        run_check = slab_run_check(torch, dist, capi, ctx, runner, sdf, minmax, full_shape, local_shape, z_offset, rank,
                                   world, device, args.dist, args.salt_p, res)

    total_vox = float(np.prod(full_shape))
    ms_per_step = elapsed / args.steps * 1e3
    value = total_vox / (elapsed / args.steps) / 1e6
    avg_ms = kernel_sum / max(len(per_step_ms), 1)
    dom = int(np.argmax(avg_ms))
    local_vox = float(np.prod(local_shape))
    achieved = ALG_BYTES_PER_VOXEL_PASS * local_vox / (avg_ms[dom] * 1e-3) / 1e9 if avg_ms[dom] > 0 else 0.0
    whole = 3 * ALG_BYTES_PER_VOXEL_PASS * local_vox / (avg_ms.sum() * 1e-3) / 1e9 if avg_ms.sum() > 0 else 0.0
    mm = minmax.cpu().numpy()

    if rank == 0:
        headline = not dist_on and not args.size and workload_key == "c4" and args.dist == "spheres" and args.variant == 0
        traffic, traffic_src = profiled_traffic(KERNEL_NAMES[dom], headline, float(avg_ms[dom]))
        line = {
            "metric": "Mvoxels/s for %s float SDF extract @%d GPU%s; %% HBM roofline%s" % (
                shape_text, world, "" if world == 1 else "s", " (per device, slowest rank)" if dist_on else ""),
            "value": round(value, 1), "unit": "Mvoxels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            # N > 1: the SAME grid (config 5) cut into more slabs = strong scaling, against the one-GPU run of that
            # grid named in same_workload_one_gpu -- never against the N = 1 default of this script (config 4)
            "scaling": "strong", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": workload, "workload_key": workload_key if not args.size else "custom",
                       "shape": list(full_shape), "parallelism": parallelism,
                       "edt_variant": args.variant, "sdf_min_max": [float(mm[0]), float(mm[1])]},
            "roofline": {"bound": "hbm", "kernel": KERNEL_NAMES[dom],
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "traffic_source": traffic_src,
                         "kernel_ms": {k: round(float(v), 4) for k, v in zip(KERNEL_NAMES, avg_ms)},
                         "whole_sdf_achieved": round(whole, 1),
                         "whole_sdf_frac": round(whole / HBM_PEAK_GBPS, 4),
                         "algorithmic_bytes_per_voxel": {"per_pass": 8, "whole_sdf": 24}},
        }
        if dist_on:
            line["phase_ms"] = phase_ms
            line["exchange"] = {"collective": "all_gather_into_tensor", "record_bytes_per_line": 4,
                                "bytes_received_per_rank": runner.exchange_bytes_received,
                                "ranks": world, "in_run_check": run_check}
            if not args.size:
                line["same_workload_one_gpu"] = one_gpu_reference(workload_key)
        if headline and not args.no_secondary:
            del occ, sdf, ws  # (8 + 7.4 GB: the secondary workloads allocate their own)
            occ = sdf = ws = None
            torch.cuda.empty_cache()
            line["secondary"] = secondary_workloads(ctx, torch, capi, device, res)
            occ = device_occupancy(torch, local_shape, args.dist, 42, device, z_offset, full_shape, args.salt_p)
        if headline and not args.no_raycast:
            try:
                line["raycast"] = raycast_section()
            except Exception as exc:
                line["raycast"] = {"error": repr(exc)}
        if not dist_on and not args.no_end_to_end and workload_key == "c4":
            try:
                line["host_path"] = end_to_end(ctx, torch, occ, local_shape, res)
            except Exception as exc:  # the headline number must not depend on host RAM for pinned buffers
                line["host_path"] = {"error": repr(exc)}
            if headline:
                try:
                    line["multi_host_path"] = multi_host_path(torch, local_shape, res)
                    bound = line["host_path"].get("pcie_lower_bound_ms")
                    if bound:
                        line["multi_host_path"]["vs_pcie_lower_bound"] = round(line["multi_host_path"]["call_ms"] / bound, 3)
                except Exception as exc:
                    line["multi_host_path"] = {"error": repr(exc)}
            if headline:
                line["cpp_host_layer"] = cpp_host_layer()
        if not args.no_cpu_baseline and not dist_on:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        else:
            line["cpu_baseline"] = None
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
