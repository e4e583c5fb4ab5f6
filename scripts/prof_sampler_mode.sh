#!/bin/bash
# kernel-trace stats of the reverse sampler in one mode (f32 | f32_split | bf16): scripts/prof_sampler_mode.sh f32_split [steps]
MODE=${1:-f32_split}; STEPS=${2:-50}
export TMPDIR=/tmp; R=$PWD; OUT=$R/gpurun_out/prof_samp_$MODE; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/run -- python3 $R/scripts/sampler_time.py $MODE $STEPS > $OUT/run.log 2>&1
cd $R
cp $(ls -t $OUT/run/*/*kernel_stats.csv | head -1) $OUT/stats.csv
tail -2 $OUT/run.log
python3 - "$OUT/stats.csv" "$STEPS" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); steps = int(sys.argv[2]) + 3
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("all kernels: %.3f ms per reverse step" % (tot / 1e6 / steps))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:30]:
    print(f"{r['Name'][:120]:120s} {int(r['Calls'])/steps:7.1f}/step {float(r['TotalDurationNs'])/1e6/steps:8.3f} ms/step {float(r['AverageNs'])/1e3:8.1f} us")
PY
