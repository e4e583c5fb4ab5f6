#!/bin/bash
# Round-4 profile of the train step at the revision named by $GIT_HEAD (the caller passes `git rev-parse --short HEAD`: .git does not travel
# to the GPU box): rocprofv3 kernel-trace stats + PMC passes (each in its own run, kernel-trace only, as MI355X_MICROARCH.md "rocprofv3 PMC
# slots" / "HBM" prescribe) + a CALIBRATION pass of SQ_VALU_MFMA_BUSY_CYCLES on scripts/micro/mfma_rate.hip, a loop known to issue
# v_mfma_f32_16x16x32_bf16 at the full rate -> gpurun_out/pmc4/{kernel_stats.csv, pmc.json}
export TMPDIR=/tmp; R=$PWD; OUT=$R/gpurun_out/pmc4; rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-sampler --windows 0"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_rate.hip -o $OUT/mfma_rate || exit 1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-sampler --windows 0 > $OUT/stats.json 2> $OUT/stats.err || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py $ARGS > $OUT/fetch.json 2> $OUT/fetch.err || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py $ARGS > $OUT/write.json 2> $OUT/write.err || exit 1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 $R/bench.py $ARGS > $OUT/sq.json 2> $OUT/sq.err || exit 1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -- python3 $R/bench.py $ARGS > $OUT/tcc.json 2> $OUT/tcc.err || exit 1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/cal -- $OUT/mfma_rate > $OUT/cal.txt 2> $OUT/cal.err || exit 1
cd $R
cp $(ls -t $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
python3 scripts/pmc_r02.py $OUT
