#!/bin/bash
# do the grouped weight-gradient kernels (side stream) run next to the chain kernels?  kernel-trace timestamps say.
export TMPDIR=/tmp; R=$PWD; OUT=$R/gpurun_out/ovl; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/run -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-sampler $@ > $OUT/bench.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, os
f = max(glob.glob("gpurun_out/ovl/run/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id"), r.get("Stream_Id")) for r in rows]
ks.sort()
grp = [k for k in ks if "wgrad_group" in k[2]]
print("group launches", len(grp), "queues", set(k[3] for k in grp), "streams", set(k[4] for k in grp))
print("chain queues", set(k[3] for k in ks if "conv_halo" in k[2]), set(k[4] for k in ks if "conv_halo" in k[2]))
for g in grp[-5:]:
    inside = [k for k in ks if k is not g and k[0] < g[1] and k[1] > g[0]]
    busy = sum(min(k[1], g[1]) - max(k[0], g[0]) for k in inside)
    print(f"group {(g[1]-g[0])/1e3:.1f} us: {len(inside)} other kernels overlap it, {busy/1e3:.1f} us of their time")
PY
