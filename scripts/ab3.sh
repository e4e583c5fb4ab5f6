#!/bin/bash
# A/B/C of several builds on ONE box, interleaved: ab3.sh rounds lib1 lib2 ...
R=$1; shift; mkdir -p gpurun_out; : > gpurun_out/ab.txt
for i in $(seq 1 $R); do
  for L in "$@"; do
    MDM_LIB_PATH=$PWD/$L timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-sampler > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$L failed"; tail -3 gpurun_out/ab.err; exit 1; }
    python - "$L" <<'PY' | tee -a gpurun_out/ab.txt
import json, sys
j = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:40s} {j['ms_per_step']:.4f} ms/step  family {j['roofline']['kernel_ms_per_step']} ms  loss {j['config']['final_loss']}")
PY
  done
done
