#!/usr/bin/env python3
"""Build libsparrow_hip_<name>.so with ONE unit's device assembly patched between hipcc's compile and assemble steps.

Why: round 5's fence around the DIEN flaky tiles (docs/open_issue_dien_tiles.md) changes the ORDER hipcc schedules blocks in, so it cannot say
whether the failing build lacks wait states somewhere or has them in the wrong place.  This script keeps the failing schedule and only inserts
`s_nop`s (or rewrites waits) at one CLASS of places, in the assembly text:

  raw    in front of every non-MFMA instruction that reads or writes a VGPR an MFMA wrote within the last WINDOW instructions  (MFMA -> VALU/LDS RAW, WAW)
  war    in front of every load (ds_read / global_load / buffer_load) whose destination overlaps a SOURCE of an MFMA of the last WINDOW instructions
  pre    in front of every MFMA one of whose sources a non-MFMA instruction wrote within the last 6 instructions                (VALU -> MFMA)
  mid    in front of every MFMA that follows, within WINDOW instructions, an MFMA it does not depend on through SrcC            (independent MFMAs back to back)
  wait0  every s_waitcnt counter -> 0
  at     in front of every line matching --at REGEX (one site)
  probe  the text of --insert FILE in front of the line matching --at REGEX, the kernel's register counts raised to --vgprs / --sgprs
  none   the unpatched assembly through the same pipeline (control)

usage: isa_patch_build.py <name> <unit.hip> <mode[,mode..]> [--nop N] [--only SUBSTR] [-D...]
The other six units come from build/r06/obj (compiled once, product flags).  Output: scripts/r06/libsparrow_hip_<name>.so (git-ignored, travels to the GPU box).
"""
import os, re, shlex, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "sparrowrecsys_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
BASE = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-pthread", "-I", os.path.join(ROOT, "include"), "-I", CSRC]
WINDOW = 14
SITES = []
REPL = []
AT2 = [None, None]                                             # a second probe site: --at2 REGEX --insert2 FILE


def regs(tok):
    """VGPR numbers a token names: v12 or v[12:15] (a[..] never appears in these kernels)."""
    out = set()
    for m in re.finditer(r"(?<![\w.])v(\d+)\b", tok):
        out.add(int(m.group(1)))
    for m in re.finditer(r"(?<![\w.])v\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def split_ops(line):
    body = line.split(";")[0].strip()
    if not body or body.startswith(".") or body.endswith(":"):
        return None, []
    parts = body.split(None, 1)
    op = parts[0]
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    return op, ops


def is_load(op):
    return op.startswith("ds_read") or op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("ds_bpermute") or op.startswith("scratch_load")


def patch(text, modes, nop, only, at=None, insert=None, vgprs=0, sgprs=0):
    lines = text.split("\n")
    out = []
    hist = []                                                     # (is_mfma, dst set, src set) of the last instructions of this function
    active = not only
    counts = {m: 0 for m in modes}
    pad = ["\ts_nop %d" % min(7, nop - 1 - 8 * i) for i in range((nop + 7) // 8)] if nop > 0 else []
    pad = []
    left = nop
    while left > 0:
        k = min(8, left); pad.append("\ts_nop %d" % (k - 1)); left -= k
    for line in lines:
        if re.match(r"^[A-Za-z_.$][\w.$]*:", line):               # a label: function entry or a block; forget history at function entry
            if line.startswith("_Z"):
                hist = []
                active = (not only) or (only in line)
        op, ops = split_ops(line)
        if active and "probe" in modes:                            # the kernel descriptor / metadata of the probed kernel
            if vgprs and re.match(r"\s*\.amdhsa_next_free_vgpr \d+", line): line = "\t\t.amdhsa_next_free_vgpr %d" % vgprs
            if vgprs and re.match(r"\s*\.amdhsa_accum_offset \d+", line): line = "\t\t.amdhsa_accum_offset %d" % vgprs
            if sgprs and re.match(r"\s*\.amdhsa_next_free_sgpr \d+", line): line = "\t\t.amdhsa_next_free_sgpr %d" % sgprs
        if op is None or not active or line.startswith("\t;") or op.startswith(";"):
            out.append(line)
            continue
        if op == "s_waitcnt" and "wait0" in modes:
            new = re.sub(r"(vmcnt|lgkmcnt|expcnt)\(\d+\)", lambda m: m.group(1) + "(0)", line)
            if new != line:
                counts["wait0"] += 1
            out.append(new)
            hist.append((False, set(), set()))
            continue
        mfma = op.startswith("v_mfma")
        stores = op.startswith("ds_write") or op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("scratch_store")
        if op.startswith("s_") or op.startswith("v_nop"):
            dst, src = set(), set().union(*[regs(o) for o in ops]) if ops else set()
        elif stores:
            dst, src = set(), set().union(*[regs(o) for o in ops])
        else:
            dst = regs(ops[0]) if ops else set()
            src = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
        recent = hist[-WINDOW:]
        why = None
        if "at" in modes and at and re.search(at, line):
            why = "at"
        if "probe" in modes and at and re.search(at, line):
            counts["probe"] += 1
            out.extend(insert.rstrip("\n").split("\n"))
            hist = []
        if "probe" in modes and AT2[0] and re.search(AT2[0], line):
            counts["probe"] += 1
            out.extend(AT2[1].rstrip("\n").split("\n"))
            hist = []
        dropped = False
        for rx, txt in REPL:
            if "probe" in modes and re.search(rx, line):
                counts["probe"] += 1
                out.extend(txt.rstrip("\n").split("\n"))
                hist = []
                dropped = True
        if dropped:
            continue
        for rx, txt in SITES:                                     # any number of further sites: --site REGEX FILE
            if "probe" in modes and re.search(rx, line):
                counts["probe"] += 1
                out.extend(txt.rstrip("\n").split("\n"))
                hist = []
        if why is None and not mfma:
            if "raw" in modes and any(h[0] and (h[1] & (src | dst)) for h in recent):
                why = "raw"
            if why is None and "war" in modes and is_load(op) and any(h[0] and (h[2] & dst) for h in recent):
                why = "war"
        elif why is None:
            if "pre" in modes and any((not h[0]) and (h[1] & src) for h in hist[-6:]):
                why = "pre"
            srcc = regs(ops[3]) if len(ops) > 3 else set()
            if why is None and "mid" in modes and any(h[0] and not (h[1] & srcc) for h in recent):
                why = "mid"
        if why:
            counts[why] += 1
            out.extend(pad)
            hist.extend([(False, set(), set())] * len(pad))
        out.append(line)
        hist.append((mfma, dst, src))
    return "\n".join(out), counts


def run(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, **kw)
    if r.returncode != 0:
        sys.exit("FAILED: %s\n%s\n%s" % (" ".join(cmd)[:400], r.stdout[-2000:], r.stderr[-4000:]))
    return r


def main():
    name, unit, modes = sys.argv[1], sys.argv[2], sys.argv[3].split(",")
    rest = sys.argv[4:]
    nop, only, defs, at, insert, vgprs, sgprs = 8, "", [], None, None, 0, 0
    i = 0
    while i < len(rest):
        if rest[i] == "--nop": nop = int(rest[i + 1]); i += 2
        elif rest[i] == "--only": only = rest[i + 1]; i += 2
        elif rest[i] == "--at": at = rest[i + 1]; i += 2
        elif rest[i] == "--insert": insert = open(rest[i + 1]).read(); i += 2
        elif rest[i] == "--at2": AT2[0] = rest[i + 1]; i += 2
        elif rest[i] == "--insert2": AT2[1] = open(rest[i + 1]).read(); i += 2
        elif rest[i] == "--site": SITES.append((rest[i + 1], open(rest[i + 2]).read())); i += 3
        elif rest[i] == "--replace": REPL.append((rest[i + 1], open(rest[i + 2]).read())); i += 3
        elif rest[i] == "--vgprs": vgprs = int(rest[i + 1]); i += 2
        elif rest[i] == "--sgprs": sgprs = int(rest[i + 1]); i += 2
        else: defs.append(rest[i]); i += 1
    objdir = os.path.join(ROOT, "build", "r06", "obj")
    os.makedirs(objdir, exist_ok=True)
    units = ["sparrow_hip.hip"] + ["tu_%d.hip" % k for k in range(1, 7)]
    procs = []
    for u in units:
        o = os.path.join(objdir, u + ".o")
        srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
        if u != unit and (not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s) for s in srcs)):
            procs.append(subprocess.Popen([HIPCC] + BASE + ["-c", os.path.join(CSRC, u), "-o", o]))
    work = os.path.join(ROOT, "build", "r06", "tmp_" + name)
    os.makedirs(work, exist_ok=True)
    cmdline = [HIPCC, "-###"] + BASE + defs + ["-c", os.path.join(CSRC, unit), "-o", os.path.join(work, "unit.o"), "--save-temps"]
    r = subprocess.run(cmdline, capture_output=True, text=True, cwd=work)
    steps = [shlex.split(l.strip()) for l in r.stderr.split("\n") if l.strip().startswith('"')]
    asm_step = next(k for k, s in enumerate(steps) if "-cc1as" in s and "amdgcn-amd-amdhsa" in s)
    for s in steps[:asm_step]:
        run(s, cwd=work)
    sfile = os.path.join(work, steps[asm_step][-1])
    text = open(sfile).read()
    open(sfile + ".orig", "w").write(text)
    new, counts = patch(text, [m for m in modes if m != "none"], nop, only, at, insert, vgprs, sgprs)
    open(sfile, "w").write(new)
    for s in steps[asm_step:]:
        run(s, cwd=work)
    for p in procs:
        if p.wait() != 0:
            sys.exit("a product unit failed to compile")
    objs = [os.path.join(work, "unit.o") if u == unit else os.path.join(objdir, u + ".o") for u in units]
    outp = os.path.join(ROOT, "scripts", "r06", "libsparrow_hip_%s.so" % name)
    run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + objs + ["-o", outp])
    import shutil
    keep = os.path.join(ROOT, "build", "r06", name + ".s")
    shutil.copy(sfile, keep)
    shutil.rmtree(work, ignore_errors=True)
    print("built %s: modes %s nop %d only %r -> patched places %s" % (os.path.relpath(outp, ROOT), modes, nop, only, counts))


if __name__ == "__main__":
    main()
