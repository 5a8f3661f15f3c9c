#!/usr/bin/env python3
"""VGPR reads that some path reaches WITHOUT an earlier write (forward must-defined dataflow over a kernel's ISA): such a read returns
whatever the SIMD's previous wave left in the register -- results that differ from run to run, for whole waves, by small amounts when the
leftover is another wave's copy of the same quantity.  usage: isa_undef_reads.py file.s kernel-regex"""
import re, sys
def regs(tok):
    tok = (tok.strip().split() or [""])[0].rstrip(",")
    m = re.match(r"^-?\|?v(\d+)\|?$", tok)
    if m: return {int(m.group(1))}
    m = re.match(r"^-?\|?v\[(\d+):(\d+)\]\|?$", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()
NO_DEF = ("global_store", "scratch_store", "ds_write", "buffer_store", "global_load_lds", "s_", "ds_bpermute_dummy", "flat_store", "v_cmpx")
def parse_kernel(lines):
    ins, labels = [], {}
    for raw in lines:
        t = raw.split(";")[0].strip()
        if not t or t.startswith("."): 
            m = re.match(r"^(\.L\w+):", t)
            if m: labels[m.group(1)] = len(ins)
            continue
        m = re.match(r"^(\.?\w+):$", t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        p = t.split(None, 1)
        ops = re.split(r",\s*", p[1]) if len(p) > 1 else []
        ins.append((p[0], ops, t))
    return ins, labels
def defs_uses(op, ops):
    allr = [regs(o) for o in ops]
    if op.startswith(NO_DEF) or (op.startswith("global_atomic") and not (ops and ops[0].startswith("v") and len(ops) > 3)):
        return set(), set().union(*allr) if allr else set()
    if op.startswith("v_cmp"):                       # dst is vcc / an SGPR pair
        return set(), set().union(*allr[1:]) if len(allr) > 1 else set()
    if op.startswith(("v_readfirstlane", "v_readlane")):
        return set(), set().union(*allr[1:])
    if op.startswith("v_permlane") and "swap" in op:
        both = allr[0] | allr[1]
        return both, both
    if op.startswith("v_writelane"):
        return allr[0], set()                        # (a partial write: counts as a def, its old lanes are not "read")
    d = allr[0] if allr else set()
    u = set().union(*allr[1:]) if len(allr) > 1 else set()
    if op.startswith(("v_fmac", "v_mac", "v_dot2c", "v_pk_fmac")): u |= d
    if op.startswith("v_cndmask") or op.startswith("v_addc") or op.startswith("v_subb"): pass
    return d, u
def analyse(name, lines):
    ins, labels = parse_kernel(lines)
    n = len(ins)
    succ = [[] for _ in range(n)]
    for i, (op, ops, t) in enumerate(ins):
        if op == "s_endpgm": continue
        if op == "s_branch":
            succ[i].append(labels[ops[0]]); continue
        if op.startswith("s_cbranch"):
            succ[i].append(labels[ops[0]])
        if i + 1 < n: succ[i].append(i + 1)
    pred = [[] for _ in range(n)]
    for i in range(n):
        for j in succ[i]: pred[j].append(i)
    ALL = set(range(512))
    IN = [ALL.copy() for _ in range(n)]
    IN[0] = {0}
    OUT = [None] * n
    du = [defs_uses(op, ops) for op, ops, t in ins]
    work = list(range(n))
    inq = [True] * n
    while work:
        i = work.pop(0); inq[i] = False
        if i != 0:
            s = None
            for p in pred[i]:
                if OUT[p] is None: continue
                s = OUT[p].copy() if s is None else (s & OUT[p])
            if s is None: s = ALL.copy() if pred[i] else set()
            IN[i] = s
        o = IN[i] | du[i][0]
        if OUT[i] != o:
            OUT[i] = o
            for j in succ[i]:
                if not inq[j]: work.append(j); inq[j] = True
    found = 0
    for i, (op, ops, t) in enumerate(ins):
        bad = du[i][1] - IN[i]
        if bad:
            print("%s  instr %d: %s   reads v%s before any write on some path" % (name[:50], i, t[:90], sorted(bad)))
            found += 1
    return found
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
cur, body, total = None, [], 0
for line in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur, body = m.group(1), []
        continue
    if cur is not None:
        if ".end_amdhsa_kernel" in line or ".Lfunc_end" in line:
            if pat.search(cur): total += analyse(cur, body)
            cur = None
            continue
        body.append(line)
print("%d reads of possibly unwritten VGPRs" % total)
sys.exit(1 if total else 0)
