#!/usr/bin/env python3
"""SGPR (and VCC) reads that some path reaches without an earlier write: a wave-uniform garbage operand (whatever the previous wave left in the
scalar register file).  Forward must-defined dataflow like isa_undef_reads.py.  usage: isa_undef_sgpr.py file.s kernel-regex [n_entry_sgprs]"""
import re, sys
VCC = {1000, 1001}
def sregs(tok):
    tok = (tok.strip().split() or [""])[0].rstrip(",")
    tok = tok.lstrip("-|").rstrip("|")
    m = re.match(r"^s(\d+)$", tok)
    if m: return {int(m.group(1))}
    m = re.match(r"^s\[(\d+):(\d+)\]$", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    if tok == "vcc": return set(VCC)
    if tok == "vcc_lo": return {1000}
    if tok == "vcc_hi": return {1001}
    return set()
NODEF_S = ("s_cmp", "s_bitcmp", "s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_sleep", "s_setprio", "s_setreg", "s_sendmsg", "s_icache", "s_dcache", "s_trap", "s_code_end")
CARRY_OUT = ("v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32", "v_addc_co_u32", "v_subb_co_u32", "v_subbrev_co_u32", "v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale")
def du(op, ops):
    a = [sregs(o) for o in ops]
    U = lambda xs: set().union(*xs) if xs else set()
    d, u = set(), set()
    e32 = op.endswith("_e32")
    if op.startswith("s_"):
        if op.startswith(NODEF_S): u = U(a)
        elif op.startswith(("s_load", "s_buffer_load")): d, u = a[0], U(a[1:])
        else: d, u = (a[0] if a else set()), U(a[1:])
        if op.startswith(("s_cbranch_vcc",)): u |= VCC
        if op.startswith(("s_cselect", "s_addc", "s_subb", "s_cbranch_scc")): pass            # (SCC not modelled)
    elif op.startswith(("v_readfirstlane", "v_readlane")): d, u = a[0], U(a[1:])
    elif op.startswith("v_cmpx"): u = U(a)
    elif op.startswith("v_cmp"):
        if e32: d, u = set(VCC), U(a[1:])
        else: d, u = a[0], U(a[1:])
    elif op.startswith(CARRY_OUT) and not e32 and len(a) > 1 and a[1]:
        d, u = a[1], U(a[2:])
    else:
        u = U(a)
        if e32 and op.startswith(("v_add_co", "v_sub_co", "v_subrev_co")): d = set(VCC)
        if e32 and op.startswith(("v_addc_co", "v_subb_co", "v_subbrev_co")): d = set(VCC); u |= VCC
        if e32 and op.startswith("v_cndmask"): u |= VCC
        if op.startswith("v_div_fmas"): u |= VCC
    return d, u
def analyse(name, body, n_entry):
    ins, labels = [], {}
    for l in body:
        t = l.split(";")[0].strip()
        if t.endswith(":"): labels[t[:-1]] = len(ins); continue
        if not t or t.startswith("."): continue
        p = t.split(None, 1); ops = re.split(r",\s*", p[1]) if len(p) > 1 else []
        ins.append((p[0], ops, t))
    N = len(ins)
    succ = [[] for _ in range(N)]
    for i, (op, ops, t) in enumerate(ins):
        if op == "s_endpgm": continue
        if op == "s_branch": succ[i].append(labels[ops[0]]); continue
        if op.startswith("s_cbranch"): succ[i].append(labels[ops[0]])
        if i + 1 < N: succ[i].append(i + 1)
    pred = [[] for _ in range(N)]
    for i in range(N):
        for j in succ[i]: pred[j].append(i)
    ALL = set(range(0, 110)) | VCC
    IN = [ALL.copy() for _ in range(N)]; IN[0] = set(range(n_entry)); OUT = [None] * N
    D = [du(op, ops) for op, ops, t in ins]
    work = list(range(N)); inq = [True] * N
    while work:
        i = work.pop(0); inq[i] = False
        if i:
            s = None
            for p in pred[i]:
                if OUT[p] is None: continue
                s = OUT[p].copy() if s is None else s & OUT[p]
            IN[i] = s if s is not None else (ALL.copy() if pred[i] else set())
        o = IN[i] | D[i][0]
        if OUT[i] != o:
            OUT[i] = o
            for j in succ[i]:
                if not inq[j]: work.append(j); inq[j] = True
    n = 0
    for i, (op, ops, t) in enumerate(ins):
        bad = D[i][1] - IN[i]
        if bad:
            n += 1
            print("%s instr %d: %s   reads %s before any write on some path" % (name[:40], i, t[:90], ["vcc" if b >= 1000 else "s%d" % b for b in sorted(bad)]))
    return n
pat = re.compile(sys.argv[2]); n_entry = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cur, body, tot = None, [], 0
for line in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):", line)
    if m: cur, body = m.group(1), []; continue
    if cur is not None:
        if ".end_amdhsa_kernel" in line:
            if pat.search(cur): tot += analyse(cur, body, n_entry)
            cur = None; continue
        body.append(line)
print("%d possibly-undefined scalar reads" % tot)
