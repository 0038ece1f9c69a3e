#!/bin/bash
# VALU / SALU instruction counts of the SDF kernels for several builds: tools/valu_by_lib.sh "<lib> ..." [bench args]
libs=$1; shift
export TMPDIR=/tmp
for lib in $libs; do
  export VGT_HIP_LIB=$PWD/voxelized_geometry_tools_amd/$lib
  rm -rf /tmp/pmc_v
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/pmc_v -- python3 bench.py --no-cpu-baseline --no-end-to-end --steps 1 --warmup 1 "$@" > /dev/null 2>&1
  python3 - $lib <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_v/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "PassKernel" not in n: continue
        k = "X" if ("float" in n.split("PassKernel")[1][:20]) else "Y"
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    v = {c: sum(x) / len(x) for c, x in acc[k].items()}
    print(sys.argv[1], k, "VALU/voxel %.1f  SALU/voxel %.1f  LDS/voxel %.2f  wait %.2f" % (v["SQ_INSTS_VALU"] * 64 / 2**30, v["SQ_INSTS_SALU"] * 64 / 2**30, v["SQ_INSTS_LDS"] * 64 / 2**30, v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]))
PY
done
