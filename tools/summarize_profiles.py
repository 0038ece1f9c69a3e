#!/usr/bin/env python3
"""Condenses a tools/collect_profiles.sh output directory into the files kept under profiles/."""
import collections
import csv
import glob
import json
import os
import sys

EXPECTED = {"PassZClassRecords": "4 GiB read (float occupancy) + 0.25 GiB write (class records)",
            "PassY": "0.25 GiB read (class records) + 4 GiB write (int32)",
            "PassXFinalize": "4 GiB read (int32) + 4 GiB write (float)"}


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxelized_geometry_tools_amd import synthetic  # noqa: E402


def classify(name):
    if "vgt::" not in name:
        return None
    if "ClassRecord" in name and "SweepPassKernel" not in name:
        return "PassZClassRecords"
    if "ScanZ" in name:
        return "PassZClassRecords"  # (cross-check builds: the int16 scan is their pass 1)
    if "SweepPassKernel" in name:
        # SweepPassKernel<InT, OutT, kFinal, ...>: the X pass writes floats
        args = name.split("SweepPassKernel", 1)[1]
        return "PassXFinalize" if ("float" in args[:40] or "IifL" in args[:8]) else "PassY"
    if "PassKernel" in name or "Brute" in name:
        # the bool template argument kFinal tells the X pass from the Y pass
        return "PassXFinalize" if ("true" in name or "b1" in name or "Finalize" in name) else "PassY"
    return None


def main():
    out = sys.argv[1]
    commit = "unknown"
    try:
        commit = open(os.path.join(out, "commit.txt")).read().strip()
    except OSError:
        pass
    summary = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -- python3 bench.py --steps 2 "
                          "--warmup 1 --no-cpu-baseline (1024^3 D1 spheres), separate passes",
               "commit": commit,
               "sources_sha256": synthetic.kernel_sources_sha256("edt"),
               "note": "MI355X_MICROARCH.md: counters are in KB; on gfx950 FETCH_SIZE reports half the bytes of a "
                       "coalesced streaming read (128-B requests tallied at 64 B), WRITE_SIZE is exact.  "
                       "'fetch_bytes_corrected' doubles the raw value: pass 1 and the X pass read one 256-B row segment "
                       "per wave instruction, the Y pass its records 1 KiB per wave instruction, the spill refills 32 B per "
                       "lane -- all whole 128-B requests at the L2 (check: pass 1's corrected fetch is the 4 GiB of the "
                       "occupancy grid).  kernel_ns / kernel_name: the kernel these bytes belong to (rocprofv3 "
                       "--kernel-trace --stats of the same build, AverageNs); bench.py reports the traffic only when "
                       "the kernel it times agrees within 10 % (the profiler's own run is a few per cent slower than an unprofiled one).",
               "kernels": {}}
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(os.path.join(out, "pmc_" + counter, "*", "*counter_collection.csv"))
        for f in files:
            for row in csv.DictReader(open(f)):
                k = classify(row["Kernel_Name"])
                if k and row["Counter_Name"] == counter:
                    per[k][counter].append(float(row["Counter_Value"]))
    for k, d in per.items():
        fetch = sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1) * 1024.0
        write = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1) * 1024.0
        wide = True
        summary["kernels"][k] = {"launches": len(d["FETCH_SIZE"]), "fetch_bytes_raw": fetch,
                                 "fetch_bytes_corrected": fetch * (2.0 if wide else 1.0), "write_bytes": write,
                                 "hbm_bytes": fetch * (2.0 if wide else 1.0) + write,
                                 "algorithmic": EXPECTED[k]}
    stats = glob.glob(os.path.join(out, "stats", "*", "*kernel_stats.csv"))
    if stats:
        rows = list(csv.reader(open(stats[0])))
        with open(os.path.join(out, "rocprof_kernel_stats.csv"), "w") as fh:
            csv.writer(fh).writerows(rows[:12])
        for row in csv.DictReader(open(stats[0])):
            k = classify(row["Name"])
            if k in summary["kernels"] and "kernel_ns" not in summary["kernels"][k]:
                summary["kernels"][k]["kernel_ns"] = float(row["AverageNs"])
                summary["kernels"][k]["kernel_name"] = row["Name"][:160]
    json.dump(summary, open(os.path.join(out, "pmc_hbm_traffic.json"), "w"), indent=1)
    # SQ counters of the three SDF kernels (one pass, 8 SQ slots)
    sq = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(out, "pmc_sq", "*", "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            k = classify(row["Kernel_Name"])
            if k:
                sq[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    if sq:
        doc = {"command": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU "
                          "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -- python3 bench.py --steps 1 "
                          "--warmup 1 (1024^3 D1)",
               "units": "averages per launch, in 1e9; SQ_INSTS_* = wave-instructions, SQ_WAVE_CYCLES / SQ_WAIT_* / "
                        "SQ_ACTIVE_INST_* = quad-cycles (MI355X_MICROARCH.md)",
               "kernels": {k: {c: round(sum(v) / len(v) / 1e9, 4) for c, v in d.items()} for k, d in sq.items()}}
        for k, d in doc["kernels"].items():
            if "SQ_INSTS_VALU" in d:
                d["valu_lane_instructions_per_voxel"] = round(d["SQ_INSTS_VALU"] * 1e9 * 64 / 2 ** 30, 1)
            if "SQ_INSTS_SALU" in d:
                d["salu_wave_instructions_per_64_voxels"] = round(d["SQ_INSTS_SALU"] * 1e9 * 64 / 2 ** 30, 1)
        json.dump(doc, open(os.path.join(out, "sq_counters.json"), "w"), indent=1)
    # raycaster
    rstats = glob.glob(os.path.join(out, "raycast_stats", "*", "*kernel_stats.csv"))
    if rstats:
        rows = list(csv.reader(open(rstats[0])))
        with open(os.path.join(out, "rocprof_kernel_stats_raycast.csv"), "w") as fh:
            csv.writer(fh).writerows(rows[:12])
    atomic = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(out, "raycast_pmc_atomic", "*", "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            if "vgt::" in name:
                short = next((k for k in ("RaycastKernel", "DirectionBinKernel", "BinOffsetsKernel", "ScatterOrderKernel",
                                          "FilterKernel") if k in name), "other")
                atomic[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    if atomic and "RaycastKernel" in atomic:
        # the bench line's atomic_roofline reads this (bench_raycast.atomic_roofline): clouds alternate A, B per launch
        counts = next(iter(atomic["RaycastKernel"].values()))
        half = len(counts) // 2
        durations = {}
        for f in rstats:
            for row in csv.DictReader(open(f)):
                if "RaycastKernel" in row["Name"]:
                    durations = {"avg_ms": float(row["AverageNs"]) * 1e-6}
        micro = {}
        try:
            micro = json.load(open(os.path.join(out, "microbench_scattered_atomics.json")))
        except (OSError, ValueError):
            pass
        bench = {}
        try:
            bench = json.load(open(os.path.join(out, "bench_raycast_config3.json")))["results"]
        except (OSError, ValueError, KeyError):
            pass
        current = {"commit": commit,
                   "sources_sha256": synthetic.kernel_sources_sha256("voxelizer"),
                   "command": "rocprofv3 --pmc TCC_ATOMIC(_sum) --kernel-trace -- python3 bench_raycast.py --no-check; "
                              "tools/microbench/scattered_atomics",
                   "scattered_atomic_rate_G_per_s": micro.get("scattered_G_atomics_per_s"),
                   "rate_source": "tools/microbench/scattered_atomics.hip on the same box: uniformly random cells of an "
                                  "int32[2 x 256^3] grid, one atomic per lane",
                   "microbench": micro,
                   "clouds": {}}
        for key, values in (("A_inside", counts[:half]), ("B_outside", counts[half:])):
            if values and key in bench:
                current["clouds"][key] = {"l2_atomics_per_launch": sum(values) / len(values),
                                          "raycast_kernel_ms": bench[key]["raycast_ms"]}
        json.dump(current, open(os.path.join(out, "raycast_atomic_current.json"), "w"), indent=1)
    if atomic:
        json.dump({"command": "rocprofv3 --pmc <TCC atomic request counters> --kernel-trace -- python3 bench_raycast.py "
                              "--no-check (config 3, clouds A then B)",
                   "note": "TCC_ATOMIC = atomic requests that reach the L2 (summed over its channels), per launch; the "
                           "launches alternate cloud A (sensor inside the grid, 155.2 M seen-free visits + 1 M end voxels) "
                           "and cloud B (sensor outside, 14.6 M visits).  One request per visit would be 156 M / 15.6 M.",
                   "kernels": {k: {c: {"per_launch": [round(x) for x in v]} for c, v in d.items()}
                               for k, d in atomic.items()}},
                  open(os.path.join(out, "raycast_atomic_counters.json"), "w"), indent=1)
    print(json.dumps(summary["kernels"], indent=1))


if __name__ == "__main__":
    main()
