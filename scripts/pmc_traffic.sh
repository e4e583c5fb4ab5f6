#!/bin/bash
# HBM traffic of the contraction-kernel family per launch, from PMC counters (separate passes, as
# MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
# half of a wide coalesced read stream, so it is doubled).  Also a kernel-trace --stats summary.
export TMPDIR=/tmp; R=$PWD; OUT=$R/gpurun_out/traffic; rm -rf $OUT; mkdir -p $OUT
cd /tmp
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-sampler"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py $ARGS > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py $ARGS > $OUT/write.json 2> $OUT/write.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-sampler > $OUT/stats.json 2> $OUT/stats.err
cp $(ls -t $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
cd $R
python3 - <<'PY'
import csv, glob, json
def per_launch(kind):
    import os
    f = max(glob.glob(f"gpurun_out/traffic/{kind}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    # every kernel an mdm_gemm call launches in the bf16 step (main contraction, tap-split epilogue, batched split-K
    # sum), divided by the number of mdm_gemm calls = optimizer steps executed (adamw launches) x calls per step
    # (two weight gradients may share one kernel launch, so kernel launches are not counted)
    calls_per_step = json.load(open(f"gpurun_out/traffic/{kind}.json"))["roofline"]["launches_per_step"]
    tot, steps = 0.0, 0
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        main = any(x in k for x in ("gemm_ring_kernel", "gemm_bf16_kernel", "conv_lin_kernel", "conv_lin2_kernel", "conv_halo_kernel", "wgrad_lin"))
        if main or "splitk_" in k:
            tot += float(r["Counter_Value"])
        if "adamw_kernel" in k:
            steps += 1
    return tot, steps * calls_per_step
f, nf = per_launch("fetch"); w, nw = per_launch("write")
out = {"kernel_family": "conv_halo / conv_lin2 / wgrad_lin / gemm_ring / gemm_bf16 kernels + the splitk_* kernels that finish them (all bf16 mdm_gemm calls)",
       "fetch_KiB_raw_per_launch": f / nf, "write_KiB_per_launch": w / nw, "launches_counted": [nf, nw],
       "hbm_bytes_per_launch": (2.0 * f / nf + w / nw) * 1024.0,
       "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM); WRITE_SIZE as reported; KiB -> bytes x1024",
       "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-sampler"}
json.dump(out, open("gpurun_out/traffic/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
