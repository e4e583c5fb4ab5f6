#!/bin/bash
# kernel-trace stats of the fp32 reverse sampler (20 steps, sample_num 100)
export TMPDIR=/tmp; R=$PWD; OUT=$R/gpurun_out/prof_samp; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --sampler-steps 50 > $OUT/bench.log 2>&1
cd $R
cp $(ls -t $OUT/run/*/*kernel_stats.csv | head -1) $OUT/stats.csv
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/prof_samp/stats.csv")))
f32 = [r for r in rows if any(k in r["Name"] for k in ("f32", "<float>", "float*", "float const")) and "adamw" not in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in f32)
print("fp32-looking kernels total %.1f ms" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f"{r['Name'][:110]:110s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:8.1f} us")
PY
