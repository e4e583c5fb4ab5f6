#!/usr/bin/env python3
"""Scan what hipcc made of every kernel for code that should not be there (DESIGN finding 47): flat_* accesses (a pointer whose address
space got lost, e.g. `cond ? *p : zero` turned into a pointer select), scratch_* traffic (spills, private arrays) and narrow global
loads where vectors were written.  Compiles each csrc/*.hip with --save-temps into a scratch directory.
    python scripts/isa_scan.py [/tmp/isa_scan]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/isa_scan"
os.makedirs(out, exist_ok=True)
src = os.path.join(ROOT, "masked-diffusion-model_amd", "csrc")
NARROW = ("global_load_dword", "global_load_ushort", "global_load_short_d16", "global_load_short_d16_hi", "global_load_ubyte")
for f in sorted(x for x in os.listdir(src) if x.endswith(".hip")):
    base = f[:-4]
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-unused-variable",
                    "--save-temps", "-c", os.path.join(src, f), "-o", base + ".o"], cwd=out, check=True, stderr=subprocess.DEVNULL)
    cur, stats = None, {}
    for line in open(os.path.join(out, base + "-hip-amdgcn-amd-amdhsa-gfx950.s")):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            cur = m.group(1); stats[cur] = [0, 0, 0, 0]; continue
        if cur is None or not line.strip():
            continue
        op = line.split()[0]
        if op.startswith("flat_"): stats[cur][0] += 1
        if op.startswith("scratch_"): stats[cur][1] += 1
        if op in NARROW: stats[cur][2] += 1
        if op.startswith("global_load_dwordx"): stats[cur][3] += 1
    for k, (fl, sc, na, wi) in stats.items():
        if fl or sc or na >= 12:
            name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:120]
            print(f"{base:6s} flat {fl:4d}  scratch {sc:4d}  narrow loads {na:4d} (wide {wi:4d})  {name}")
