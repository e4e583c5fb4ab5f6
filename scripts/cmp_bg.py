#!/usr/bin/env python3
"""Side-by-side of bench_gemm.py logs: python scripts/cmp_bg.py gpurun_out/bg_1.log gpurun_out/bg_3.log ..."""
import sys
rows, tots = {}, []
for k, f in enumerate(sys.argv[1:]):
    for l in open(f):
        if "|" in l and "@" in l:
            name = l[:38].strip(); parts = [p.split() for p in l.split("|")[1:]]
            rows.setdefault(name, {})[k] = [float(p[0]) for p in parts]
        if l.startswith("weighted"): tots.append(l.strip()[:110])
n = len(sys.argv) - 1
print(f"{'shape':36s} | " + " | ".join(f"{c:^{7*n}s}" for c in ("fwd us", "dgrad us", "wgrad us")))
for name, r in rows.items():
    print(f"{name:36s} | " + " | ".join(" ".join(f"{(r[k][c] if r[k][c] < 1e8 else 0):6.1f}" for k in range(n)) for c in range(3)))
for t in tots: print(t)
