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
    os.makedirs(os.path.join(root, "build"), exist_ok=True)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", "include", "-I", "sparrowrecsys_amd/csrc", "-DSPRK_SINGLE_TU", "--cuda-device-only", "-S",
                    "sparrowrecsys_amd/csrc/sparrow_hip.hip", "-o", "build/sparrow.s"], cwd=root, check=True, stderr=subprocess.DEVNULL)
txt = open(os.path.join(root, "build", "sparrow.s")).read()

def regs(s):
    out = set()
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", s): out |= set(range(int(a), int(b) + 1))
    out |= set(int(x) for x in re.findall(r"(?<![\w\[:])v(\d+)\b", s))
    return out

bad = 0
names = re.findall(r"^(_ZN(?:12_GLOBAL__N_1|8sprk_dev)11k_din_fused\w+):\s", txt, re.M)
for sym in names:
    m = re.search(r"^%s:\s.*?\n(.*?)\.amdhsa_kernel" % re.escape(sym), txt, re.S | re.M)
    body = m.group(1).split("\n")
    inst = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
    inst = inst.replace("void (anonymous namespace)::", "").replace("void sprk_dev::", "").split("((anonymous")[0].split("(")[0]
    in_asm, asm_lines = False, set()
    for i, l in enumerate(body):
        if "#ASMSTART" in l: in_asm = True
        elif "#ASMEND" in l: in_asm = False
        elif in_asm: asm_lines.add(i)
    loads = [i for i in sorted(asm_lines) if "global_load_dwordx4" in body[i]]     # (the walk starts at the first 16-byte hidden load: the numerics' two dword loads in front of the ids are the oldest entries of every queue and are not tracked)
    drain = [i for i in asm_lines if "s_waitcnt vmcnt(0)" in body[i]]
    if not loads or not drain:
        print("%s: no hidden loads found" % inst); bad += 1; continue
    lo, hi = min(loads), max(drain)
    labels = {re.match(r"^(\.LBB\d+_\d+):", l).group(1): i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    # A worklist over the control-flow graph from the first hidden load: a state is the FIFO of outstanding operations (the registers
    # each one owns); a conditional branch forks, an edge is re-walked only with a state not seen on it before, a walk ends at the
    # final drain (vmcnt(0) with nothing left), at s_endpgm, or where it leaves [first hidden load's block .. last drain] backwards
    # (the next task of a persistent launch starts drained).
    def key(p): return tuple(tuple(sorted(x)) for x in p)
    dma_lines, wait_lines, touches, loop_lines, skip = set(), set(), [], set(), set()
    work, seen = [(lo, [])], set()
    steps = 0
    while work and steps < 2000000:
        i, pending = work.pop()
        pending = list(pending)
        while i < len(body) and steps < 2000000:
            steps += 1
            t = body[i].strip()
            if i in asm_lines:
                dm = re.search(r"global_load_dword(?:x\d)? v\[(\d+):(\d+)\]", t) or re.search(r"global_load_dword v(\d+)()", t)
                wm = re.search(r"s_waitcnt vmcnt\((\d+)\)", t)
                if dm: pending.append(set(range(int(dm.group(1)), int(dm.group(2) or dm.group(1)) + 1)))
                elif wm:
                    wait_lines.add(i)
                    n = int(wm.group(1))
                    if len(pending) > n: pending = pending[len(pending) - n:]
                    if i == hi: break                          # the final drain of the task
            elif "global_load_lds_dwordx4" in t:
                pending.append(set())                         # an LDS-DMA piece: occupies a vmcnt slot, owns no register
                dma_lines.add(i)
            elif t and not t.startswith((";", ".")):
                if t.startswith("s_endpgm"): break
                bm = re.match(r"(s_c?branch\w*)\s+(\.LBB\d+_\d+)", t)
                if bm:
                    tgt = labels[bm.group(2)]
                    k = (i, tgt, key(pending))
                    if tgt >= lo - 400 and k not in seen:    # (a branch far back: the task loop's back edge, taken drained)
                        seen.add(k)
                        if bm.group(1) == "s_branch": i = tgt; continue
                        work.append((tgt, list(pending)))
                    elif bm.group(1) == "s_branch": break
                else:
                    inflight = set().union(*pending) if pending else set()
                    if regs(t) & inflight and not t.startswith("s_"): touches.append((i, t[:100]))
            i += 1
    dma, seen_waits = len(dma_lines), len(wait_lines)
    # the slot loop(s): loops (hipcc marks their header labels) behind the first hidden load that hold a manual wait with a count
    for lab, pos in labels.items():
        if pos < lo or "Loop Header" not in body[pos]: continue
        # the loop's blocks: the header and every block hipcc marks "in Loop: Header=BB<header>" (a rotated loop's body sits in FRONT of
        # its header in the layout), each from its label to the next label
        hid = lab[2:]                                        # ".LBB229_49" -> "BB229_49"
        order = sorted(labels.values())
        blocks = [p2 for l2, p2 in labels.items() if l2 == lab or ("Header=%s " % hid) in body[p2] + " "]
        lines = set()
        for p2 in blocks:
            nxt = [q for q in order if q > p2]
            lines |= set(range(p2, nxt[0] if nxt else len(body)))
        if any(w in lines and "vmcnt(0)" not in body[w] for w in wait_lines): loop_lines |= lines
    stray = [i for i in sorted(loop_lines) if "vmcnt" in body[i] and i not in asm_lines]
    # compiler-placed waits between the first hidden load and the loop: each is as strict as the operations the COMPILER knows to be
    # younger (the DMA pieces), so it may only show up with a count >= the pieces still wanted in flight
    early = [(i, body[i].strip()) for i in range(lo, hi) if "vmcnt" in body[i] and i not in asm_lines and i not in loop_lines and i not in skip]
    uniq = sorted(set(touches))
    for ln, t in uniq[:6]: print("   %s: line %d touches a register with a load in flight: %s" % (inst, ln, t))
    for ln in stray[:6]: print("   %s: compiler-placed wait inside the slot loop, line %d: %s" % (inst, ln, body[ln].strip()))
    print("%s: %d hidden loads, %d DMA pieces behind them, %d manual waits walked, early touches %d, compiler vmcnt waits inside the loop %d, outside it %s" %
          (inst, len(loads), dma, seen_waits, len(uniq), len(stray), [re.search(r"vmcnt\((\d+)\)", t).group(1) for _, t in early][:8]))
    bad += len(uniq) + len(stray)
sys.exit(1 if bad else 0)
