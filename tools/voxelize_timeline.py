"""Timeline of one VoxelizePointClouds call with 8 clouds from a rocprofv3 --kernel-trace [--memory-copy-trace] CSV pair of
tests/cpp/bench_voxelize: prints, for the LAST 8-cloud call, every kernel / copy with start and end relative to the call's
first event, so that overlaps between the clouds' streams can be read off.
  rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/vt -o vt --output-format csv -- tests/cpp/bench_voxelize
  python tools/voxelize_timeline.py gpurun_out/vt"""
import csv
import glob
import sys


def main(d):
    events = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            short = next((k for k in ("RaycastKernel", "DirectionBinKernel", "ScatterOrderKernel", "FilterKernel", "fillBuffer",
                                      "AccumulateCounts") if k in name), name[:30])
            events.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Stream_Id", r.get("Queue_Id", "?"))))
    for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            events.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", ""), "-"))
    events.sort()
    # the last FilterKernel ends the last call; walk back to the 8th RaycastKernel before it
    last_filter = max(i for i, e in enumerate(events) if e[2] == "FilterKernel")
    count, i = 0, last_filter
    while i > 0 and count < 8:
        i -= 1
        if events[i][2] == "RaycastKernel":
            count += 1
    # include the sort / memset / copies just before the first raycast
    first = i
    while first > 0 and events[first - 1][0] > events[i][0] - 2_000_000 and events[first - 1][2] != "FilterKernel":
        first -= 1
    t0 = events[first][0]
    for s, e, name, q in events[first:last_filter + 1]:
        print("%9.1f %9.1f  %-22s %8.1f us  q=%s" % ((s - t0) / 1e3, (e - t0) / 1e3, name, (e - s) / 1e3, q))


if __name__ == "__main__":
    main(sys.argv[1])
