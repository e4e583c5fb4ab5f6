#!/bin/bash
# A/B two builds of libmdm_hip.so on ONE box, interleaved: ab_libs.sh libA.so libB.so [rounds]
A=$1; B=$2; R=${3:-3}; mkdir -p gpurun_out; : > gpurun_out/ab.txt
for i in $(seq 1 $R); do
  for L in $A $B; do
    MDM_LIB_PATH=$PWD/$L timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-sampler > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$L failed"; tail -3 gpurun_out/ab.err; exit 1; }
    python - "$L" <<'PY' | tee -a gpurun_out/ab.txt
import json, sys
j = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:40s} {j['ms_per_step']:.4f} ms/step  family {j['roofline']['kernel_ms_per_step']} ms")
PY
  done
done
