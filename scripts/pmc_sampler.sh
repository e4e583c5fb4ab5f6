#!/bin/bash
# PMC passes over the reverse sampler of record (fp32 storage, split products; 20 reverse steps x 100 images), each counter set in its
# own --kernel-trace --pmc run as MI355X_MICROARCH.md prescribes -> gpurun_out/pmc_sampler/pmc.json (per kernel: HBM-side bytes per
# launch = FETCH_SIZE x2 on gfx950 + WRITE_SIZE, MFMA busy, LDS bank-conflict share, wait shares)
MODE=${1:-f32_split}
export TMPDIR=/tmp; R=$PWD; OUT=$R/gpurun_out/pmc_sampler; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/scripts/sampler_time.py $MODE 20 > $OUT/fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/scripts/sampler_time.py $MODE 20 > $OUT/write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 $R/scripts/sampler_time.py $MODE 20 > $OUT/sq.log 2>&1 || exit 1
cd $R
python3 scripts/pmc_sampler.py $OUT $MODE
