#!/bin/bash
# step time under HIP runtime environment knobs (same box, interleaved): env_probe.sh
mkdir -p gpurun_out; : > gpurun_out/env_probe.txt
run() {
  env "$@" timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-sampler --no-roofline > gpurun_out/ep.json 2> gpurun_out/ep.err || { echo "$* failed"; tail -3 gpurun_out/ep.err; return; }
  python - "$*" <<'PY' | tee -a gpurun_out/env_probe.txt
import json, sys
j = json.loads(open("gpurun_out/ep.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:60s} {j['ms_per_step']:.4f} ms/step  loss {j['config']['final_loss']}")
PY
}
for i in 1 2; do
run X=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run DEBUG_HIP_FORCE_GRAPH_QUEUES=4
done
