#!/usr/bin/env python3
"""LDS / VMEM loads whose destination registers are still an operand (A, B, C) of an MFMA issued a few instructions earlier: the load's data
comes back asynchronously; the MFMA reads its operands when the matrix pipe STARTS it, which under contention (four waves per SIMD, chains of
dependent MFMAs) can be later than the compiler's fixed wait-state count assumes.  usage: isa_mfma_war.py file.s kernel-regex [window]"""
import re, sys, collections
def regs(tok):
    tok = (tok.strip().split() or [""])[0].rstrip(",")
    m = re.match(r"^-?\|?v(\d+)\|?$", tok)
    if m: return {int(m.group(1))}
    m = re.match(r"^-?\|?v\[(\d+):(\d+)\]\|?$", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()
def kernels(path):
    cur, body = None, []
    for l in open(path):
        m = re.match(r"^(_Z\w+):", l)
        if m: cur, body = m.group(1), []; continue
        if cur is not None:
            if ".end_amdhsa_kernel" in l: yield cur, body; cur = None; continue
            body.append(l)
pat = re.compile(sys.argv[2]); WIN = int(sys.argv[3]) if len(sys.argv) > 3 else 12
LOADS = ("ds_read", "ds_bpermute", "ds_swizzle", "global_load", "scratch_load", "buffer_load", "flat_load")
tot = 0
for name, body in kernels(sys.argv[1]):
    if not pat.search(name): continue
    ins = []
    for n, l in enumerate(body, 1):
        t = l.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"): continue
        p = t.split(None, 1); ops = re.split(r",\s*", p[1]) if len(p) > 1 else []
        ins.append((p[0], ops, t, n))
    hist = collections.Counter(); ex = []
    for i, (op, ops, t, n) in enumerate(ins):
        if not op.startswith("v_mfma"): continue
        srcs = {"A": regs(ops[1]), "B": regs(ops[2]), "C": regs(ops[3]) if len(ops) > 3 else set()}
        d = 0
        for j in range(i + 1, min(i + 1 + 3 * WIN, len(ins))):
            bop, bops, bt, bn = ins[j]
            if bop == "s_nop": d += int(bops[0]) + 1; continue
            if bop.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_barrier")) or d >= WIN: break
            if bop.startswith(LOADS) and not bop.startswith("global_load_lds") and bops:
                w = regs(bops[0])
                for k, s in srcs.items():
                    if w & s:
                        hist[(k, bop.split("_b")[0][:12], d)] += 1
                        ex.append("    line %d %s  <-  +%d ws  line %d %s" % (n, t[:62], d, bn, bt[:44]))
            d += 1
    print(name[:80], " total", sum(hist.values()))
    for k, v in sorted(hist.items()): print("   ", k, v)
    for e in ex[:int(sys.argv[4]) if len(sys.argv) > 4 else 0]: print(e)
    tot += sum(hist.values())
