export TMPDIR=/tmp
OUT=gpurun_out/occ; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_LEVEL_WAVES SQ_CYCLES --kernel-trace --output-format csv -d $OUT/p1 -- python3 bench.py --no-cpu-baseline --no-end-to-end --no-raycast --steps 1 --warmup 1 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/occ/p1/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "vgt::" not in n: continue
        k = "ScanZ" if "ScanZ" in n else ("X" if "<int, float" in n else ("Y" if "Sweep" in n else n[:30]))
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in acc:
    print(k, {c: sum(v)/len(v) for c, v in acc[k].items()})
PY
