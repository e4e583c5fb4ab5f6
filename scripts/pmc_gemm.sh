#!/bin/bash
# PMC counters of the contraction kernel on selected shapes (separate passes, no tracing flags besides kernel-trace)
export TMPDIR=/tmp; R=$PWD; OUT=$R/gpurun_out/pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
         "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  SHAPES=${SHAPES:-0,10} PASSES=${PASSES:-fwd} rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -- python3 $R/scripts/bench_gemm.py > $OUT/p$i.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/pmc/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60] + " grid=" + r.get("Grid_Size", "?")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] in ("SQ_WAVES", "FETCH_SIZE", "TCC_HIT_sum", "SQ_ACTIVE_INST_VALU", "TCP_TCC_READ_REQ_sum"): cnt[(k, r["Counter_Name"])] += 1
for k, v in agg.items():
    if "gemm" not in k: continue
    print(k)
    for c, x in sorted(v.items()):
        n = max(1, max(cnt[(k, cc)] for cc in ("SQ_WAVES", "FETCH_SIZE", "TCC_HIT_sum", "SQ_ACTIVE_INST_VALU", "TCP_TCC_READ_REQ_sum")))
        print(f"   {c:34s} {x / n:16.1f} per launch")
PY
