#!/bin/bash
# same-box A/B of one environment switch: ab_env.sh VAR=VALUE [pairs]   (A = default, B = with the variable set), interleaved runs of the headline step
V=$1; N=${2:-3}
for i in $(seq 1 $N); do
  python bench.py --no-cpu-baseline --no-sampler --no-roofline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("A default      ", d["ms_per_step"], d["value"])'
  env $V python bench.py --no-cpu-baseline --no-sampler --no-roofline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("B '$V'", d["ms_per_step"], d["value"])'
done
