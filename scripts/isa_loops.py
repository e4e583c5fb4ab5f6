#!/usr/bin/env python3
"""What hipcc made of the DMA loops (DESIGN finding 32).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 --save-temps=obj -c masked-diffusion-model_amd/csrc/gemm.hip -o /tmp/isa/gemm.o
    python scripts/isa_loops.py /tmp/isa/gemm-hip-amdgcn-amd-amdhsa-gfx950.s                 # scan: every loop with an LDS-DMA ring
    python scripts/isa_loops.py /tmp/isa/gemm-hip-amdgcn-amd-amdhsa-gfx950.s <mangled-name-part>   # condensed listing of one kernel

scan: for every backward branch whose body holds `global_load_lds` AND MFMAs, the number of `s_waitcnt vmcnt(0)` and of ordinary
global / scalar loads inside it -- a `vmcnt(0)` in such a loop waits for every DMA in flight (the prefetch ring collapses), and an
ordinary vector load whose result the loop consumes brings one with it.
listing: waits, barriers and branches verbatim, everything else as run lengths (mfma / ds_read / ds_write / dma / gload / other)."""
import re
import sys


def functions(s):
    for n in re.findall(r'^(_ZN3mdm\S*):', s, re.M):
        i = s.index('\n' + n + ':')
        j = s.find('.Lfunc_end', i)
        if j > 0:
            yield n, s[i:j].split('\n')


def loops(lines):
    labels = {}
    for k, l in enumerate(lines):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = k
    for k, l in enumerate(lines):
        m = re.search(r's_cbranch\S*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in labels and labels[tgt] < k:
                yield lines[labels[tgt]:k]


def scan(s):
    for n, lines in functions(s):
        rep = []
        for body in loops(lines):
            ndma = sum('global_load_lds' in b for b in body)
            nmf = sum('v_mfma' in b for b in body)
            nv0 = sum(bool(re.search(r's_waitcnt.*vmcnt\(0\)', b)) for b in body)
            gl = [b.strip() for b in body if re.match(r'\s*(global_load_(dword|ubyte|ushort|sbyte)|buffer_load)', b) and 'lds' not in b]
            if ndma and nmf >= 8 and (nv0 or gl):
                rep.append(f"    loop of {len(body)} lines: {nmf} mfma, {ndma} dma, {nv0} x vmcnt(0), {len(gl)} ordinary loads {gl[:2]}")
        if rep:
            print(n[:140])
            print('\n'.join(rep))


def listing(s, pat):
    n, lines = next((n, l) for n, l in functions(s) if pat in n)
    out, run, cnt = [], None, 0

    def flush():
        nonlocal run, cnt
        if run:
            out.append(f"   {run} x{cnt}")
        run, cnt = None, 0
    for ln in lines:
        t = ln.strip()
        if not t or t.startswith(';'):
            continue
        if t.startswith('v_mfma'):
            key = 'mfma'
        elif t.startswith('ds_read') or t.startswith('ds_load'):
            key = 'ds_read ' + t.split()[0]
        elif t.startswith('ds_write') or t.startswith('ds_store'):
            key = 'ds_write ' + t.split()[0]
        elif t.startswith('global_load_lds'):
            key = 'dma'
        elif t.startswith('global_load') or t.startswith('buffer_load'):
            key = 'gload'
        elif t.startswith('global_store'):
            key = 'gstore'
        elif t.startswith(('s_waitcnt', 's_barrier', 's_cbranch', 's_branch', '.LBB')):
            flush()
            out.append(t.split(';')[0].strip())
            continue
        else:
            key = 'other'
        if key == run:
            cnt += 1
        else:
            flush()
            run, cnt = key, 1
    flush()
    print(n)
    print('\n'.join(out))


if __name__ == "__main__":
    text = open(sys.argv[1]).read()
    if len(sys.argv) > 2:
        listing(text, sys.argv[2])
    else:
        scan(text)
