#!/bin/bash
# k_din_attn_cols hides its row loads from hipcc's waitcnt pass (k_din_cols.h).  This script compiles the device code to ISA and
# checks, for every instantiation, that (1) the slot loop contains no s_waitcnt vmcnt other than the three manual vmcnt(2 sets)
# (a compiler-placed one would drain the prefetch) and (2) no instruction touches a register set between the asm statement that
# loads it and the asm statement that waits for it.  Run in the build container: bash scripts/r03/check_din_cols_isa.sh
cd "$(dirname "$0")/../.."
mkdir -p build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I sparrowrecsys_amd/csrc --cuda-device-only -S sparrowrecsys_amd/csrc/sparrow_hip.hip -o build/sparrow.s 2>/dev/null
python3 - <<'PY'
import re, sys
txt = open('build/sparrow.s').read()
bad = 0
for name in ("ILi2ELb0", "ILi2ELb1", "ILi1ELb0", "ILi1ELb1"):
    m = re.search(r"^_ZN12_GLOBAL__N_1\d+k_din_attn_cols%s\w*:\s.*?\n(.*?)\.amdhsa_kernel" % name, txt, re.S | re.M)
    body = m.group(1).split("\n")
    asm_lines = set()
    inside = False
    for i, l in enumerate(body):
        if "#ASMSTART" in l: inside = True
        elif "#ASMEND" in l: inside = False
        elif inside: asm_lines.add(i)
    loads = [i for i in asm_lines if "global_load_dwordx4" in body[i]]
    waits = [i for i in asm_lines if re.search(r"s_waitcnt vmcnt\((4|2)\)", body[i])]
    drain = [i for i in asm_lines if "s_waitcnt vmcnt(0)" in body[i]]
    lo, hi = min(waits), max(drain)
    stray = [i for i in range(lo, hi) if "vmcnt" in body[i] and i not in asm_lines]
    def regs(s):
        out = set()
        for a, b in re.findall(r"v\[(\d+):(\d+)\]", s): out |= set(range(int(a), int(b) + 1))
        out |= set(int(x) for x in re.findall(r"(?<![\w\[:])v(\d+)\b", s))
        return out
    touched = 0
    for i in loads:
        dm = re.search(r"global_load_dwordx4 v\[(\d+):(\d+)\]", body[i])
        dst = set(range(int(dm.group(1)), int(dm.group(2)) + 1))
        j = i + 1
        while j < len(body) and j not in waits and j not in drain:
            t = body[j].strip()
            if t and not t.startswith((";", ".")) and j not in asm_lines and (regs(t) & dst):
                touched += 1
                print("   %s: line %d touches a set before its wait: %s" % (name, j, t[:90]))
                break
            j += 1
    print("k_din_attn_cols<%s>: %d hidden loads, %d manual waits, compiler vmcnt waits inside the slot loop: %d, early touches: %d"
          % (name, len(loads), len(waits), len(stray), touched))
    bad += len(stray) + touched
sys.exit(1 if bad else 0)
PY
