#!/bin/bash
# does a second stream run next to the first one under any queue setting?  (bench.py --overlap: weight-gradient groups on a side stream)
mkdir -p gpurun_out; : > gpurun_out/env_probe2.txt
run() {
  local extra="$1"; shift
  env "$@" timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-sampler --no-roofline $extra > gpurun_out/ep.json 2> gpurun_out/ep.err || { echo "$* $extra failed"; tail -3 gpurun_out/ep.err; return; }
  python - "$* $extra" <<'PY' | tee -a gpurun_out/env_probe2.txt
import json, sys
j = json.loads(open("gpurun_out/ep.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:70s} {j['ms_per_step']:.4f} ms/step  loss {j['config']['final_loss']}")
PY
}
env | grep -i "GPU_MAX_HW\|HIP_\|HSA_\|ROC" | head -20
run "" X=0
run "--overlap" X=0
run "--overlap" GPU_MAX_HW_QUEUES=8
run "--overlap" DEBUG_HIP_FORCE_GRAPH_QUEUES=4 GPU_MAX_HW_QUEUES=8
run "--overlap" HIP_LAUNCH_BLOCKING=0 GPU_MAX_HW_QUEUES=2
python scripts/queue_probe.py 2>&1 | tail -5
GPU_MAX_HW_QUEUES=8 python scripts/queue_probe.py 2>&1 | tail -5
