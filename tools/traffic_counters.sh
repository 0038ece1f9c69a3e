export TMPDIR=/tmp
OUT=gpurun_out/traf; rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  d=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$d -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-raycast > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/traf/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "vgt::" not in n: continue
        k = "ScanZ" if "ScanZ" in n else ("X" if "<int, float" in n else ("Y" if "Sweep" in n else None))
        if k: acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k, {c: round(sum(v)/len(v)/1e6, 2) for c, v in acc[k].items()}, "(1e6; SIZE counters in KB -> GB)")
PY
