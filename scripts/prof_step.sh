#!/bin/bash
# kernel-trace stats of the train step -> gpurun_out/prof_step/stats.csv (+ per-category summary)
export TMPDIR=/tmp; R=$PWD; OUT=$R/gpurun_out/prof_step; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/run -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-sampler --no-roofline > $OUT/bench.log 2>&1
cd $R
cp $(ls -t $OUT/run/*/*kernel_stats.csv | head -1) $OUT/stats.csv
python3 scripts/prof_summary.py $OUT/stats.csv
