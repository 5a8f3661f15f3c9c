#!/usr/bin/env python3
"""isa_waitcnt_check.py across basic blocks: explores the kernel's control-flow graph (every branch both ways, states memoised per block
entry) replaying lgkmcnt / vmcnt as in-order queues, and reports registers read or overwritten while the load that writes them can still be
outstanding on SOME path -- e.g. a load issued before a loop's back edge and consumed behind it.
usage: isa_waitcnt_paths.py file.s kernel-regex"""
import re, sys
def regs(tok):
    tok = (tok.strip().split() or [""])[0].rstrip(",")
    m = re.match(r"^-?\|?v(\d+)\|?$", tok)
    if m: return frozenset({int(m.group(1))})
    m = re.match(r"^-?\|?v\[(\d+):(\d+)\]\|?$", tok)
    if m: return frozenset(range(int(m.group(1)), int(m.group(2)) + 1))
    return frozenset()
def kernels(path):
    cur, body = None, []
    for l in open(path):
        m = re.match(r"^(_Z\w+):", l)
        if m: cur, body = m.group(1), []; continue
        if cur is not None:
            if ".end_amdhsa_kernel" in l: yield cur, body; cur = None; continue
            body.append(l)
def analyse(name, body):
    ins, labels = [], {}
    for n, l in enumerate(body, 1):
        t = l.split(";")[0].strip()
        if t.endswith(":"):
            labels[t[:-1]] = len(ins); continue
        if not t or t.startswith("."): continue
        p = t.split(None, 1); ops = re.split(r",\s*", p[1]) if len(p) > 1 else []
        ins.append((p[0], ops, t, n))
    N = len(ins)
    leaders = {0} | set(labels.values())
    for i, (op, ops, t, n) in enumerate(ins):
        if op.startswith(("s_branch", "s_cbranch")) and i + 1 < N: leaders.add(i + 1)
    # ONE long path: fall through every forward branch, follow unconditional ones, take every backward branch `reps` times
    # (a loop body is replayed with whatever its previous trip left outstanding); the queues are capped like the hardware counters
    viol, seen = {}, [0]
    taken = {}
    lg, vm, i, steps = [], [], 0, 0
    while i < N and steps < 200000:
        op, ops, t, n = ins[i]
        steps += 1
        nxt = i + 1
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m and int(m.group(1)) < len(lg): lg = lg[len(lg) - int(m.group(1)):] if int(m.group(1)) else []
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m and int(m.group(1)) < len(vm): vm = vm[len(vm) - int(m.group(1)):] if int(m.group(1)) else []
        elif op == "s_endpgm":
            break
        elif op == "s_branch":
            tgt = labels[ops[0]]
            if tgt > i or taken.get(i, 0) < REPS:
                taken[i] = taken.get(i, 0) + 1; nxt = tgt
        elif op.startswith("s_cbranch"):
            tgt = labels[ops[0]]
            if tgt <= i and taken.get(i, 0) < REPS:
                taken[i] = taken.get(i, 0) + 1; nxt = tgt
            elif tgt > i and MODE == "skip" and tgt - i < 40 and not any(o[0].startswith("s_cbranch") and labels[o[1][0]] <= j for j, o in enumerate(ins[i + 1:tgt], i + 1)):
                pass
        elif not op.startswith("s_"):
            allr = frozenset().union(*[regs(o) for o in ops]) if ops else frozenset()
            for q, nm in ((lg, "lgkmcnt"), (vm, "vmcnt")):
                for dst, ln in q:
                    if dst & allr: viol.setdefault((n, ln), (t, nm, sorted(dst & allr)))
            if op.startswith(("ds_read", "ds_bpermute", "ds_swizzle", "ds_permute")): lg.append((regs(ops[0]), n))
            elif op.startswith("ds_"): lg.append((frozenset(), n))
            elif op.startswith("global_load_lds"): vm.append((frozenset(), n))
            elif op.startswith(("global_load", "scratch_load", "buffer_load", "flat_load")): vm.append((regs(ops[0]), n))
            elif op.startswith(("global_store", "scratch_store", "buffer_store", "global_atomic", "flat_store")): vm.append((frozenset(), n))
        elif op.startswith(("s_load", "s_buffer_load")):
            lg.append((frozenset(), n))
        lg, vm = lg[-15:], vm[-63:]
        i = nxt
    seen = [steps]
    print("%s: %d instructions replayed, %d violations" % (name[:70], seen[0], len(viol)))
    for (n, ln), (t, nm, r) in sorted(viol.items())[:30]:
        print("   line %d: %s   touches v%s of the load at line %d (%s): %s" % (n, t[:60], r, ln, nm, body[ln - 1].strip()[:50]))
    return len(viol)
pat = re.compile(sys.argv[2]); tot = 0
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 2
MODE = "fall"
for name, body in kernels(sys.argv[1]):
    if pat.search(name): tot += analyse(name, body)
sys.exit(1 if tot else 0)
