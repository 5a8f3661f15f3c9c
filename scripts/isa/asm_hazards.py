#!/usr/bin/env python3
"""Hazards hipcc's recognizer cannot see: instructions INSIDE asm statements (v_fma_mix_f32 of dyn_split.h / k_din_attn.h, the permlane swaps
of rows4_sum / rows4_max) that read a VGPR a transcendental (1 wait state on gfx940+) or an MFMA (its passes) wrote just before.
usage: asm_hazards.py file.s [kernel-regex]   -- prints every suspicious adjacency; exit 1 if any."""
import re, sys
MFMA_WAIT = 19       # the longest XDL write -> VALU read distance of the 16x16 shapes used here (16 passes + 3)
TRANS = ("v_rcp", "v_exp", "v_log", "v_rsq", "v_sqrt", "v_sin", "v_cos")
def regs(tok):
    tok = (tok.strip().split() or [""])[0].rstrip(",")
    m = re.match(r"^-?\|?v(\d+)\|?$", tok)
    if m: return {int(m.group(1))}
    m = re.match(r"^v\[(\d+):(\d+)\]$", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()
def parse(line):
    t = line.split(";")[0].strip()
    if not t or t.startswith(".") or t.endswith(":"): return None
    parts = t.split(None, 1)
    op = parts[0]
    ops = [o for o in re.split(r",\s*", parts[1])] if len(parts) > 1 else []
    return op, ops
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
cur, body, found = None, [], 0
def check(name, body):
    global found
    in_asm = False
    for i, (raw, ins) in enumerate(body):
        if "#ASMSTART" in raw: in_asm = True
        if "#ASMEND" in raw: in_asm = False
        if ins is None or not in_asm: continue
        op, ops = ins
        if not (op.startswith("v_fma_mix") or op.startswith("v_permlane")): continue
        src = set()
        for o in (ops if op.startswith("v_permlane") else ops[1:]): src |= regs(o)
        # nearest earlier writer of every source register (wait states between: one per instruction, s_nop n = n + 1)
        for reg in sorted(src):
            dist, j = 0, i - 1
            while j >= 0 and dist < 40:
                bins = body[j][1]
                j -= 1
                if bins is None: continue
                bop, bops = bins
                if bop == "s_nop":
                    dist += int(bops[0]) + 1
                    continue
                writes = bops and not bop.startswith(("global_store", "scratch_store", "ds_write", "buffer_store", "s_", "global_atomic"))
                if writes and reg in regs(bops[0]):
                    need = 1 if bop.startswith(TRANS) else (MFMA_WAIT if bop.startswith("v_mfma") else (2 if op.startswith("v_permlane") and bop.startswith("v_") else 0))
                    if dist < need:
                        print("%s: %s reads v%d written by %s %d wait states earlier (needs %d)" % (name[:60], op, reg, bop, dist, need)); found += 1
                    break
                dist += 1
for line in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur, body = m.group(1), []
        continue
    if cur:
        if ".end_amdhsa_kernel" in line or ".Lfunc_end" in line:
            if pat.search(cur): check(cur, body)
            cur = None
            continue
        body.append((line, parse(line) if "#ASM" not in line else None))
print("%d suspicious adjacencies" % found)
sys.exit(1 if found else 0)
