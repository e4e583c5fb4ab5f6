#!/bin/bash
# kernel-trace stats of one of the other BASELINE configurations: prof_config.sh cfg3|cfg4 -> gpurun_out/prof_<cfg>/stats.csv
# (the headline cfg2 step runs too -- 4 + 2 steps -- the extra configuration runs --steps/--warmup as well; the stats mix both:
#  cfg-specific kernels are told apart by their template arguments / launch counts)
CFG=${1:-cfg4}
export TMPDIR=/tmp; R=$PWD; OUT=$R/gpurun_out/prof_$CFG; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/run -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sampler --no-roofline --windows 0 --config $CFG > $OUT/bench.json 2> $OUT/bench.log
cd $R
cp $(ls -t $OUT/run/*/*kernel_stats.csv | head -1) $OUT/stats.csv
python3 - "$OUT/stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:32]:
    print(f"{r['Name'][:120]:120s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:8.1f} us")
PY
