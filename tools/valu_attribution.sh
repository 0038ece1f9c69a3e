export TMPDIR=/tmp
for s in 0 2 1 3 11 32; do
  VGT_HIP_LIB=voxelized_geometry_tools_amd/libvgt_hip_dbg.so VGT_HULL_SKIP=$s rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/attr_$s -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
python3 - <<PY
import csv,glob,collections
for s in (0,2,1,3,11,32):
    fs=glob.glob("gpurun_out/attr_%d/*/*counter_collection.csv"%s)
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        n=r["Kernel_Name"]
        k="X" if ("HullPass" in n and "true" in n) else ("Y" if "HullPass" in n else None)
        if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(s, {k:{c:round(sum(x)/len(x)/1e9,3) for c,x in v.items()} for k,v in agg.items()})
PY
