#!/usr/bin/env python3
"""The gfx950 behaviour behind the DIEN flaky tiles (docs/open_issue_dien_tiles.md, scripts/ubench/pkfma_opsel_mfma.hip): a packed-f32 VALU
instruction -- v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 -- whose LOW result takes the HIGH dword of a VGPR src1 (op_sel's second bit set) reads that
dword as 0 in lanes 48..63 while another wave of the SIMD keeps v_mfma_f32_16x16x32_f16 in flight.  hipcc emits the form by itself when two scalar
fmas share a factor that sits in the high half of a 64-bit load (SLP + op_sel folding).  This scanner lists every such instruction per kernel.

usage: isa_pk_opsel.py file.s [file.s ...]        (device assembly: hipcc -S --cuda-device-only)
       isa_pk_opsel.py --so libsparrow_hip.so      (the built library: its gfx950 code objects are unbundled and disassembled)
exit status 1 if any VGPR-src1 hit exists.  SGPR-src1 hits (a scalar pair's high half: not a VGPR read) are listed, not counted
(scripts/ubench/pkfma_opsel_mfma.hip measures them clean)."""
import os, re, subprocess, sys, tempfile

PK = re.compile(r"^\s*(v_pk_(?:fma|mul|add)_f32)\s+(.*)$")


def src1_hi_for_lo(op, rest):
    """(is_hit, src1 token) for one packed-f32 instruction's operand text."""
    body = rest.split(";")[0].split("//")[0]
    m = re.search(r"op_sel:\[([01](?:,[01])*)\]", body)
    if not m:
        return False, None
    bits = m.group(1).split(",")
    if len(bits) < 2 or bits[1] != "1":
        return False, None
    ops = [o.strip() for o in body.split(" op_sel")[0].split(",")]
    return True, (ops[2] if len(ops) > 2 else "?")


def scan_text(text, label):
    kernel, hits = "?", []
    for line in text.split("\n"):
        m = re.match(r"^([A-Za-z_][\w.$]*):", line)
        if m and not m.group(1).startswith(".L") and not m.group(1).startswith("BB"):
            kernel = m.group(1)
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)              # llvm-objdump function header
        if m:
            kernel = m.group(1)
        body = re.sub(r"^\s*[0-9a-f]+:\s+", "", line) if re.match(r"^\s*[0-9a-f]+:\s", line) else line
        pm = PK.match(body if body.startswith(("\t", " ")) else "\t" + body)
        if not pm:
            continue
        hit, s1 = src1_hi_for_lo(pm.group(1), pm.group(2))
        if hit:
            hits.append((label, kernel, s1.startswith("v"), body.strip()))
    return hits


def disassemble_so(path):
    llvm = "/opt/rocm/lib/llvm/bin"
    out = []
    with tempfile.TemporaryDirectory() as td:
        # the fat binary sits in .hip_fatbin; clang-offload-bundler wants the section's bytes
        fat = os.path.join(td, "fat.bin")
        subprocess.run([os.path.join(llvm, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, path], check=True)
        data = open(fat, "rb").read()
        # several bundles may be concatenated (one per translation unit): split on the magic string
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), data)]
        for k, st in enumerate(starts):
            part = os.path.join(td, "b%d.bin" % k)
            open(part, "wb").write(data[st:(starts[k + 1] if k + 1 < len(starts) else len(data))])
            co = os.path.join(td, "co%d.o" % k)
            r = subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part, "--output=" + co,
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            d = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True)
            out.append(d.stdout)
    return out


def main():
    args = sys.argv[1:]
    hits = []
    if args and args[0] == "--so":
        texts = disassemble_so(args[1])
        if not texts:
            print("no gfx950 code object found in %s" % args[1])
            return 2
        n_insn = 0
        for k, t in enumerate(texts):
            n_insn += len(re.findall(r"v_pk_(?:fma|mul|add)_f32", t))
            hits += scan_text(t, "%s[%d]" % (os.path.basename(args[1]), k))
        print("%d code objects, %d packed-f32 fma / mul / add instructions scanned" % (len(texts), n_insn))
    else:
        for f in args:
            hits += scan_text(open(f).read(), os.path.basename(f))
    bad = [h for h in hits if h[2]]
    for label, kernel, vg, ins in hits:
        print("%s  %s  %s: %s" % ("VGPR src1.hi -> lo" if vg else "sgpr src1.hi -> lo", label, kernel, ins))
    print("%d packed-f32 instructions take the high dword of a VGPR src1 for their low result (%d more from an SGPR pair)" % (len(bad), len(hits) - len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
