#!/usr/bin/env python3
"""Instruction mix of a kernel's loops, read off build/sparrow.s (hipcc -S --cuda-device-only of the library's translation unit).
    python scripts/r05/isa_loop_mix.py 'k_mlp_rows<8, 8, 2, 8, true, 1>' [--dump file] [--s build/other.s]
Prints, per basic-block range between a `Loop Header` label and its last back edge, the count of every mnemonic class (VALU by
mnemonic, MFMA, LDS, VMEM, SALU, waits) -- the static twin of the PMC SQ_INSTS_* counters, available without a GPU."""
import re, subprocess, sys, os, collections
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
args = [a for a in sys.argv[1:]]
spath = os.path.join(root, "build", "sparrow.s")
dump = None
if "--s" in args: i = args.index("--s"); spath = args[i + 1]; del args[i:i + 2]
if "--dump" in args: i = args.index("--dump"); dump = args[i + 1]; del args[i:i + 2]
pat = args[0]
txt = open(spath).read()
syms = re.findall(r"^(_Z\w+):\s", txt, re.M)
dem = subprocess.run(["c++filt"] + syms, capture_output=True, text=True).stdout.strip().split("\n")
def cls(m):
    if m.startswith("v_mfma"): return "MFMA"
    if m.startswith("ds_"): return "LDS:" + m
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")): return "VMEM:" + m
    if m.startswith("s_waitcnt"): return "wait"
    if m.startswith("s_"): return "SALU"
    if m.startswith("v_"): return "VALU:" + m
    return "other:" + m
for sym, name in zip(syms, dem):
    name = name.replace("void (anonymous namespace)::", "").replace("void sprk_dev::", "").split("((anonymous")[0].split("(sprk_dev::")[0]
    if pat not in name: continue
    m = re.search(r"^%s:\s.*?\n(.*?)\.amdhsa_kernel" % re.escape(sym), txt, re.S | re.M)
    if not m: continue
    body = m.group(1).split("\n")
    if dump: open(dump, "w").write("\n".join(body))
    # label -> index; loops = (header idx, last branch to that header idx)
    lab = {}
    for i, l in enumerate(body):
        mm = re.match(r"^(\.LBB\d+_\d+):.*Loop Header", l)
        if mm: lab[mm.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        mm = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in lab and lab[mm.group(1)] < i: loops.append((lab[mm.group(1)], i))
    best = {}
    for h, e in loops: best[h] = max(best.get(h, 0), e)
    print("== %s: %d lines, %d loops" % (name, len(body), len(best)))
    for h, e in sorted(best.items(), key=lambda t: t[0] - t[1])[:3]:
        c = collections.Counter()
        for l in body[h:e + 1]:
            t = l.strip()
            if not t or t.startswith((";", ".")) or t.endswith(":"): continue
            c[cls(t.split()[0])] += 1
        tot = collections.Counter()
        for k, v in c.items(): tot[k.split(":")[0]] += v
        print("  loop lines %d..%d: %s" % (h, e, dict(tot)))
        for k, v in c.most_common(40):
            if ":" in k: print("     %-40s %d" % (k, v))
