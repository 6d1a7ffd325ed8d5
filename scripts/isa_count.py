#!/usr/bin/env python3
"""Static instruction mix of the sweep kernels (gfx950 ISA): per kernel the VALU/SALU/VMEM/... counts of the whole
body and of its largest loop.  Used to steer VALU-bound tuning without a GPU:  python scripts/isa_count.py [extra flags]"""
import re, subprocess, sys, collections
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
src = REPO / "adaptive_sph_amd/csrc/sph_sweeps.hip"
out = "/tmp/sph_sweeps.s"
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-I", str(REPO / "include"),
       "-S", "--cuda-device-only", "-o", out, str(src)] + sys.argv[1:]
subprocess.run(cmd, check=True)
txt = open(out).read()
funcs = re.split(r"\n(_Z\w+):\s*;? *@?\w*\n", txt)
def cls(op):
    if op.startswith("v_"): return "valu"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("global_load") or op.startswith("buffer_load"): return "vmem_rd"
    if op.startswith("global_store") or op.startswith("buffer_store"): return "vmem_wr"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_atomic"): return "atomic"
    return "other"
for k in range(1, len(funcs), 2):
    name, body = funcs[k], funcs[k + 1]
    if "k_sweep" not in name: continue
    body = body.split(".Lfunc_end")[0]
    m = re.search(r"(Op\w+?)(Lb|E)", name)
    short = re.sub(r"^_Z\d+k_sweepI\d+", "", name)[:60]
    lines = [l.strip() for l in body.split("\n")]
    # basic blocks by label
    blocks, cur = collections.OrderedDict(), "entry"
    blocks[cur] = collections.Counter()
    for l in lines:
        if re.match(r"^\.LBB\d+_\d+:", l):
            cur = l.split(":")[0]; blocks[cur] = collections.Counter(); continue
        if not l or l.startswith(";") or l.startswith("."): continue
        op = l.split()[0]
        blocks[cur][cls(op)] += 1
    tot = collections.Counter()
    for c in blocks.values(): tot.update(c)
    vg = re.search(r"; NumVgprs: (\d+)", funcs[k + 1]); sc = re.search(r"; ScratchSize: (\d+)", funcs[k + 1]); oc = re.search(r"; Occupancy: (\d+)", funcs[k + 1])
    big = sorted(blocks.items(), key=lambda kv: -kv[1]["valu"])[:3]
    print(f"{short:60s} total valu={tot['valu']:4d} salu={tot['salu']:4d} vmem_rd={tot['vmem_rd']:3d} nop={tot['nop']:3d} vgpr={vg.group(1) if vg else '?'} scratch={sc.group(1) if sc else '?'} occ={oc.group(1) if oc else '?'}")
    for lab, c in big:
        print(f"      {lab:10s} valu={c['valu']:4d} salu={c['salu']:3d} vmem_rd={c['vmem_rd']:3d} nop={c['nop']:3d} wait={c['wait']:2d}")
