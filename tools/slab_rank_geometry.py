#!/usr/bin/env python3
"""Every rank's REAL share of the 8 / 4 / 2-GPU run of BASELINE config 5 (2048 x 2048 x 1024), timed on ONE GPU.

For world = 8, 4, 2 the slab pipeline of a middle rank (rank = world // 2: neighbours on both sides) runs device-resident
exactly as multi_gpu.SlabSdf runs it -- pass 1 with per-line summaries, the carry kernel over the gathered summaries of
ALL ranks (every rank's slab is generated and summarised here, one after the other, so the carries are the real ones),
the record fix-up, the Y and the X pass -- with the library's per-kernel events.  The collective itself cannot be
measured on one GPU: `predicted_step_ms` adds a stated estimate of it and is labelled as a prediction.

    python tools/slab_rank_geometry.py [--out profiles/r4/slab_rank_geometry.json] [--steps 5]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FULL = (2048, 2048, 1024)
ALL_GATHER_GBPS = 300.0  # assumed effective all-gather receive rate per GPU over xGMI (7 links x ~153 GB/s peak)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    import torch
    import bench
    from voxelized_geometry_tools_amd import capi, multi_gpu

    device = torch.device("cuda", 0)
    ctx = capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    res = 0.01
    result = {"grid": list(FULL), "distribution": "D1 spheres seed 42", "assumed_all_gather_GBps": ALL_GATHER_GBPS,
              "worlds": {}}
    lines = FULL[0] * FULL[1]
    for world in (8, 4, 2):
        gathered = torch.empty((world, lines, 2), dtype=torch.int16, device=device)
        rank = world // 2
        keep = None
        for r in range(world):
            local_shape, z0 = multi_gpu.slab_of(FULL, r, world)
            occ = bench.device_occupancy(torch, local_shape, "spheres", 42, device, z0, FULL)
            nbytes = capi.sdf_workspace_bytes(local_shape)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            ctx.sdf_slab_begin(occ.data_ptr(), local_shape, z0, ws.data_ptr(), nbytes, gathered[r].data_ptr(), True)
            ctx.synchronize()
            if r == rank:
                keep = (local_shape, z0, occ)
            del ws
        local_shape, z0, occ = keep
        nbytes = capi.sdf_workspace_bytes(local_shape)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        sdf = torch.empty(local_shape, dtype=torch.float32, device=device)
        summary = torch.empty((lines, 2), dtype=torch.int16, device=device)
        carries = torch.empty((lines, 4), dtype=torch.int16, device=device)
        mm = torch.zeros(2, dtype=torch.float32, device=device)
        ms_begin = np.zeros(1, dtype=np.float32)
        ms_finish = np.zeros(3, dtype=np.float32)
        acc = np.zeros(5)
        for i in range(args.steps + 1):
            ctx.sdf_slab_begin(occ.data_ptr(), local_shape, z0, ws.data_ptr(), nbytes, summary.data_ptr(), True, ms_begin)
            start = torch.cuda.Event(enable_timing=True)
            stop = torch.cuda.Event(enable_timing=True)
            start.record()
            ctx.sdf_slab_carries(gathered.data_ptr(), world, rank, FULL[0], FULL[1], FULL[2], carries.data_ptr())
            stop.record()
            ctx.sdf_slab_finish(local_shape, z0, FULL[2], res, carries.data_ptr(), sdf.data_ptr(), ws.data_ptr(), nbytes,
                                mm.data_ptr(), False, ms_finish)
            torch.cuda.synchronize()
            if i:
                acc += [ms_begin[0], start.elapsed_time(stop), ms_finish[0], ms_finish[1], ms_finish[2]]
        acc /= args.steps
        vox = float(np.prod(local_shape))
        compute = float(acc.sum())
        gather_bytes = (world - 1) * lines * 4
        gather_ms = gather_bytes / (ALL_GATHER_GBPS * 1e9) * 1e3
        # the same slab through the plain single-device pipeline (no summaries, no carries): what the slab path adds
        plain = np.zeros(3)
        ws_plain = torch.empty(capi.sdf_workspace_bytes(local_shape), dtype=torch.uint8, device=device)
        ms3 = np.zeros(3, dtype=np.float32)
        for i in range(args.steps + 1):
            ctx.sdf_dev(occ.data_ptr(), local_shape, res, sdf.data_ptr(), ws_plain.data_ptr(), ws_plain.numel(), mm.data_ptr(),
                        kernel_ms=ms3)
            if i:
                plain += ms3
        plain /= args.steps
        result["worlds"][str(world)] = {
            "rank": rank, "local_shape": list(local_shape), "z_offset": int(z0),
            "phase_ms": {k: round(float(v), 4) for k, v in zip(
                ("pass1_records+summaries", "carries", "record_fixup", "Y", "X"), acc)},
            "compute_ms": round(compute, 4),
            "whole_sdf_frac_of_8TBps": round(24.0 * vox / (compute * 1e-3) / 8e12, 4),
            "same_slab_plain_pipeline_ms": {"kernels": [round(float(v), 4) for v in plain], "sum": round(float(plain.sum()), 4)},
            "predicted_all_gather_ms": round(gather_ms, 4),
            "predicted_step_ms": round(compute + gather_ms, 4),
            "predicted_note": "PREDICTION, not a measurement: this rank's measured kernels + %d MiB received by the "
                              "all-gather at an ASSUMED %.0f GB/s; no collective has run" % (gather_bytes >> 20, ALL_GATHER_GBPS)}
        del occ, ws, sdf, ws_plain
        torch.cuda.empty_cache()
    # the whole grid on one GPU, plain pipeline: the series' reference point
    occ = bench.device_occupancy(torch, FULL, "spheres", 42, device)
    sdf = torch.empty(FULL, dtype=torch.float32, device=device)
    ws = torch.empty(capi.sdf_workspace_bytes(FULL), dtype=torch.uint8, device=device)
    mm = torch.zeros(2, dtype=torch.float32, device=device)
    ms3 = np.zeros(3, dtype=np.float32)
    tot = np.zeros(3)
    for i in range(4):
        ctx.sdf_dev(occ.data_ptr(), FULL, res, sdf.data_ptr(), ws.data_ptr(), ws.numel(), mm.data_ptr(), kernel_ms=ms3)
        if i:
            tot += ms3
    tot /= 3
    result["one_gpu_whole_grid"] = {"kernel_ms": [round(float(v), 4) for v in tot], "sum_ms": round(float(tot.sum()), 4),
                                    "whole_sdf_frac_of_8TBps": round(24.0 * float(np.prod(FULL)) / (tot.sum() * 1e-3) / 8e12, 4)}
    for world, entry in result["worlds"].items():
        entry["predicted_speedup_vs_one_gpu"] = round(float(tot.sum()) / entry["predicted_step_ms"], 3)
    text = json.dumps(result, indent=1)
    print(text)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as fh:
            fh.write(text + "\n")


if __name__ == "__main__":
    main()
