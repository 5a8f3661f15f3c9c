#!/usr/bin/env python3
"""Prints, for every kernel of build/sparrow.s whose name matches the argument, the sequence of VMEM loads, vmcnt waits, barriers and
MFMA groups in LAYOUT order with the loop structure hipcc annotates -- the view in which round 4 found (a) `dst[c] = src[c]` compiled to
a wait per 16 bytes, (b) a vmcnt(0) in k_din_fused's slot loop in front of an instruction that merely encoded a pending load's
register, (c) two vmcnt(0) behind k_mlp_rows' row gathers.  Cold blocks (slow paths) appear inline: read with the source next to it.
    python scripts/r04/isa_wait_sequence.py k_deepfm_pairs [--compile]"""
import re, subprocess, sys, os
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
if "--compile" in sys.argv:
    os.makedirs(os.path.join(root, "build"), exist_ok=True)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", "include", "-I", "sparrowrecsys_amd/csrc", "-DSPRK_SINGLE_TU", "--cuda-device-only", "-S",
                    "sparrowrecsys_amd/csrc/sparrow_hip.hip", "-o", "build/sparrow.s"], cwd=root, check=True, stderr=subprocess.DEVNULL)
pat = [a for a in sys.argv[1:] if not a.startswith("--")][0]
txt = open(os.path.join(root, "build", "sparrow.s")).read()
for sym in re.findall(r"^(_ZN(?:12_GLOBAL__N_1|8sprk_dev)\w+):\s", txt, re.M):
    name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip().replace("void (anonymous namespace)::", "").replace("void sprk_dev::", "").split("((anonymous")[0].split("(sprk_dev::")[0]
    if pat not in name: continue
    body = re.search(r"^%s:\s.*?\n(.*?)\.amdhsa_kernel" % re.escape(sym), txt, re.S | re.M).group(1).split("\n")
    out, prev, cnt = [], None, 0
    def flush():
        global prev, cnt
        if prev: out.append(prev if cnt == 1 else "%s x%d" % (prev, cnt))
        prev, cnt = None, 0
    for l in body:
        t = l.strip()
        k = None
        if re.match(r"^\.LBB\d+_\d+:.*Loop Header", t): k = "\n  LOOP{"
        elif t.startswith("global_load_lds"): k = "dma"
        elif t.startswith("global_load") or t.startswith("buffer_load"): k = "load"
        elif t.startswith("scratch_"): k = "SCRATCH"
        elif t.startswith("global_store"): k = "store"
        elif "vmcnt" in t: k = "wait(%s)" % re.search(r"vmcnt\((\d+)\)", t).group(1)
        elif t.startswith("s_barrier"): k = "BARRIER"
        elif t.startswith("v_mfma"): k = "mfma"
        if k is None: continue
        if k == prev: cnt += 1
        else:
            flush(); prev, cnt = k, 1
    flush()
    print("== %s\n   %s\n" % (name, " ".join(out)))
