#!/bin/bash
# usage: bench_sweep.sh "<bench args 1>" "<bench args 2>" ...  -> one short line per run in gpurun_out/sweep.txt
mkdir -p gpurun_out; : > gpurun_out/sweep.txt
for a in "$@"; do
  timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-sampler $a > gpurun_out/sw.json 2> gpurun_out/sw.err
  rc=$?
  python - "$a" $rc <<'PY' | tee -a gpurun_out/sweep.txt
import json, sys
try:
    j = json.loads(open("gpurun_out/sw.json").read().strip().splitlines()[-1])
    print(f"{sys.argv[1]!r:40s} rc={sys.argv[2]} ms/step {j['ms_per_step']:.4f} img/s {j['value']:.0f} fam {j['roofline']['kernel_ms_per_step']} ms {j['roofline']['launches_per_step']} launches")
except Exception as e:
    print(f"{sys.argv[1]!r:40s} rc={sys.argv[2]} FAILED {e}")
PY
  if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then exit $rc; fi
done
