"""Per-dispatch durations of the raycast path's kernels from a rocprofv3 --kernel-trace CSV (bench_raycast.py runs cloud
A's calls first, then cloud B's): prints the median per kernel for each half of the RaycastKernel dispatches.
  rocprofv3 --kernel-trace -d gpurun_out/rk -o rk --output-format csv -- python3 bench_raycast.py --no-check
  python tools/raycast_kernel_times.py gpurun_out/rk/rk_kernel_trace.csv"""
import csv
import statistics
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = ("RaycastKernel", "DirectionBinKernel", "ScatterOrderKernel", "fillBuffer")
    per = {n: [] for n in names}
    for r in rows:
        for n in names:
            if n in r["Kernel_Name"]:
                per[n].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    ray = per["RaycastKernel"]
    half = len(ray) // 2
    cut = ray[half][0]
    for label, keep in (("A", lambda s: s < cut), ("B", lambda s: s >= cut)):
        out = []
        for n in names:
            d = [e - s for s, e in per[n] if keep(s)]
            out.append("%s %.1f us (n=%d)" % (n, statistics.median(d) / 1e3, len(d)))
        # span of one call: DirectionBin start -> Raycast end
        starts = [s for s, _ in per["DirectionBinKernel"] if keep(s)]
        ends = [e for s, e in ray if keep(s)]
        spans = [e - s for s, e in zip(starts, ends)]
        out.append("call span %.1f us" % (statistics.median(spans) / 1e3))
        print(label + ": " + "; ".join(out))


if __name__ == "__main__":
    main(sys.argv[1])
