#!/usr/bin/env python3
"""k_din_fused hides its row loads from hipcc's waitcnt pass (k_din_fused.h: the loads and the `s_waitcnt vmcnt(N)` that own
their registers are asm statements).  Two properties of the GENERATED code keep that correct and fast, so they are checked on the
ISA of every instantiation (build/sparrow.s; run scripts/kernel_resources.sh or pass --compile first):
  (1) no instruction touches a register with a hidden load still in flight (a copy of such a register -- hipcc did emit them
      when the slot loop had two arms -- reads stale rows);
  (2) no compiler-placed vmcnt wait sits inside the slot loop (it would be counted without the hidden loads: never too weak,
      but it drains the prefetch every round).
The walk follows the layout order and takes every backward branch once, so each loop body is seen twice with the state carried
over the back edge.  Exit status 1 on any finding."""
import re, subprocess, sys, os
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
if "--compile" in sys.argv:
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", "include", "-I", "sparrowrecsys_amd/csrc", "--cuda-device-only", "-S",
                    "sparrowrecsys_amd/csrc/sparrow_hip.hip", "-o", "build/sparrow.s"], cwd=root, check=True, stderr=subprocess.DEVNULL)
txt = open(os.path.join(root, "build", "sparrow.s")).read()

def regs(s):
    out = set()
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", s): out |= set(range(int(a), int(b) + 1))
    out |= set(int(x) for x in re.findall(r"(?<![\w\[:])v(\d+)\b", s))
    return out

bad = 0
names = re.findall(r"^(_ZN12_GLOBAL__N_111k_din_fused\w+):\s", txt, re.M)
for sym in names:
    m = re.search(r"^%s:\s.*?\n(.*?)\.amdhsa_kernel" % re.escape(sym), txt, re.S | re.M)
    body = m.group(1).split("\n")
    inst = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
    inst = inst.replace("void (anonymous namespace)::", "").split("((anonymous")[0].split("(")[0]
    in_asm, asm_lines = False, set()
    for i, l in enumerate(body):
        if "#ASMSTART" in l: in_asm = True
        elif "#ASMEND" in l: in_asm = False
        elif in_asm: asm_lines.add(i)
    loads = [i for i in sorted(asm_lines) if "global_load_dwordx4" in body[i]]
    drain = [i for i in asm_lines if "s_waitcnt vmcnt(0)" in body[i]]
    if not loads or not drain:
        print("%s: no hidden loads found" % inst); bad += 1; continue
    lo, hi = min(loads), max(drain)
    labels = {re.match(r"^(\.LBB\d+_\d+):", l).group(1): i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    # blocks laid out inside the region but reached only from BEFORE it (hipcc moves cold prologue arms behind the loop) are not on
    # any path with a load in flight: skip them
    targets = {}
    for i, l in enumerate(body):
        bm = re.match(r"\s*s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if bm: targets.setdefault(bm.group(1), []).append(i)
    order = sorted(labels.items(), key=lambda kv: kv[1])
    skip = set()
    for k, (lab, pos) in enumerate(order):
        if pos <= lo or pos > hi: continue
        prev = pos - 1
        while prev > 0 and (not body[prev].strip() or body[prev].strip().startswith((";", "."))): prev -= 1
        falls_in = not re.match(r"\s*(s_branch|s_endpgm|s_setpc)", body[prev]) and prev not in skip
        if not falls_in and all(t < lo or t in skip for t in targets.get(lab, [])):
            end = order[k + 1][1] if k + 1 < len(order) else len(body)
            skip |= set(range(pos, end))
    pending = []            # FIFO of (set of registers) per outstanding hidden load
    taken, i, steps, touches, seen_waits = set(), lo, 0, [], 0
    loop_lines = set()
    while i <= hi and steps < 200000:
        steps += 1
        t = body[i].strip()
        if i in skip:
            i += 1
            continue
        if i in asm_lines:
            dm = re.search(r"global_load_dwordx4 v\[(\d+):(\d+)\]", t)
            wm = re.search(r"s_waitcnt vmcnt\((\d+)\)", t)
            if dm: pending.append(set(range(int(dm.group(1)), int(dm.group(2)) + 1)))
            elif wm:
                seen_waits += 1
                n = int(wm.group(1))
                if len(pending) > n: pending = pending[len(pending) - n:]
        elif t and not t.startswith((";", ".")):
            bm = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", t)
            if bm and labels.get(bm.group(1), 1 << 30) <= i and lo <= labels[bm.group(1)] and (i, bm.group(1)) not in taken:
                taken.add((i, bm.group(1)))
                loop_lines |= set(range(labels[bm.group(1)], i + 1))
                i = labels[bm.group(1)]
                continue
            inflight = set().union(*pending) if pending else set()
            hit = regs(t) & inflight
            if hit and not t.startswith("s_"):
                touches.append((i, t[:100]))
        i += 1
    stray = [i for i in sorted(loop_lines) if "vmcnt" in body[i] and i not in asm_lines]
    uniq = sorted(set(touches))
    for ln, t in uniq[:6]: print("   %s: line %d touches a register with a load in flight: %s" % (inst, ln, t))
    for ln in stray[:6]: print("   %s: compiler-placed wait inside the slot loop, line %d: %s" % (inst, ln, body[ln].strip()))
    print("%s: %d hidden loads, %d manual waits walked, early touches %d, compiler vmcnt waits inside the loop %d" % (inst, len(loads), seen_waits, len(uniq), len(stray)))
    bad += len(uniq) + len(stray)
sys.exit(1 if bad else 0)
