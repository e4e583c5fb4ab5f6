#!/bin/bash
# same-box A/B of two bench.py flag sets, interleaved: ab_flags.sh "<flags A>" "<flags B>" [rounds]
A="$1"; B="$2"; R=${3:-3}; mkdir -p gpurun_out; : > gpurun_out/ab.txt
for i in $(seq 1 $R); do
  for F in "$A" "$B"; do
    timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-sampler --no-roofline $F > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "[$F] failed"; tail -3 gpurun_out/ab.err; exit 1; }
    python - "[$F]" <<'PY' | tee -a gpurun_out/ab.txt
import json, sys
j = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:30s} {j['ms_per_step']:.4f} ms/step  loss {j['config']['final_loss']}")
PY
  done
done
