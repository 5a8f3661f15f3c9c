#!/usr/bin/env python3
"""Replays lgkmcnt / vmcnt along every basic block of a kernel's ISA (queues start empty at a block's entry: only intra-block errors are
seen) and reports a register that is read or overwritten while the load that writes it is still outstanding.
usage: isa_waitcnt_check.py file.s kernel-regex"""
import re, sys
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
pat = re.compile(sys.argv[2]); found = 0
for name, body in kernels(sys.argv[1]):
    if not pat.search(name): continue
    lg, vm = [], []                      # outstanding (dst regs, text)
    for n, l in enumerate(body, 1):
        t = l.split(";")[0].strip()
        if t.endswith(":"):
            lg, vm = [], []; continue
        if not t or t.startswith("."): continue
        p = t.split(None, 1); op = p[0]; ops = re.split(r",\s*", p[1]) if len(p) > 1 else []
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m: lg = lg[len(lg) - int(m.group(1)):] if int(m.group(1)) < len(lg) else lg
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m: vm = vm[len(vm) - int(m.group(1)):] if int(m.group(1)) < len(vm) else vm
            if m and int(m.group(1)) == 0: vm = []
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m and int(m.group(1)) == 0: lg = []
            continue
        if op.startswith(("s_branch", "s_cbranch", "s_barrier", "s_endpgm")):
            if op == "s_endpgm": lg, vm = [], []
            continue
        allr = set()
        for o in ops: allr |= regs(o)
        is_vm_load = op.startswith(("global_load", "scratch_load", "buffer_load", "flat_load")) and not op.startswith("global_load_lds")
        is_lg_load = op.startswith(("ds_read", "ds_bpermute", "ds_swizzle", "ds_permute"))
        srcr = set()
        for o in ops[1:]: srcr |= regs(o)
        for q, nm in ((lg, "lgkmcnt"), (vm, "vmcnt")):
            for dst, txt, ln in q:
                # (a later load of the SAME counter may overwrite an earlier one's registers: they return in order; reading them is not fine)
                touched = (dst & srcr) if ((nm == "vmcnt" and is_vm_load) or (nm == "lgkmcnt" and is_lg_load)) else (dst & allr)
                if touched:
                    print("%s line %d: %s   touches v%s of outstanding (%s) line %d %s" % (name[:40], n, t[:70], sorted(touched), nm, ln, txt[:50])); found += 1
        if op.startswith(("ds_read", "ds_bpermute", "ds_swizzle", "ds_permute")): lg.append((regs(ops[0]), t, n))
        elif op.startswith(("ds_write", "ds_add", "ds_")): lg.append((set(), t, n))
        elif op.startswith("s_load") or op.startswith("s_buffer_load"): lg.append((set(), t, n))
        elif op.startswith(("global_load_lds",)): vm.append((set(), t, n))
        elif op.startswith(("global_load", "scratch_load", "buffer_load", "flat_load")): vm.append((regs(ops[0]), t, n))
        elif op.startswith(("global_store", "scratch_store", "buffer_store", "global_atomic")): vm.append((set(), t, n))
print("%d violations" % found)
