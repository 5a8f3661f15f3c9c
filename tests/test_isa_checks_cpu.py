"""k_din_fused hides its row loads from hipcc's waitcnt pass and owns them with hand-counted `s_waitcnt vmcnt(N)` statements
(sparrowrecsys_amd/csrc/k_din_fused.h).  Whether that is CORRECT is a property of the generated ISA, not of the source: hipcc has
twice copied registers with a load still in flight.  scripts/r04/check_din_fused_isa.py compiles the device
code (no GPU needed) and walks the control-flow graph of every instantiation; this test runs it."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_din_fused_hidden_loads_are_never_touched_in_flight():
    env = dict(os.environ, PATH=os.environ.get("PATH", "") + ":/opt/rocm/bin")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "r04", "check_din_fused_isa.py"), "--compile"],
                       capture_output=True, text=True, env=env, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("k_din_fused<")]
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert len(lines) >= 10, r.stdout[-2000:]                    # KC x MB x TAIL + the two attention-weights instantiations
    for l in lines:
        assert "early touches 0" in l and "inside the loop 0" in l, l
    # the one-batch tail forms count the image's DMA pieces behind the hidden loads; the persistent forms stage the image up front
    # (3 coefficient + 11 image pieces + the A fragments' piece)
    assert any("k_din_fused<2, false, true" in l and "15 DMA pieces" in l for l in lines)
    assert any("k_din_fused<2, true, true" in l and " 0 DMA pieces" in l for l in lines)


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_kernel_family_units_instantiate_what_the_single_unit_build_would():
    """The library is seven translation units (csrc/tu_kernels.h): the heavy kernel templates are instantiated in tu_1 .. tu_6.hip and only
    declared `extern template` in sparrow_hip.hip.  The list (csrc/tu_instances.h) is GENERATED from the kernels a single-unit build
    instantiates implicitly; a stale list is still a correct library (the main unit instantiates what is not listed) but this keeps the
    committed file honest."""
    env = dict(os.environ, PATH=os.environ.get("PATH", "") + ":/opt/rocm/bin")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen_tu_instances.py"), "--check"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    units = sorted(f for f in os.listdir(os.path.join(ROOT, "sparrowrecsys_amd", "csrc")) if f.startswith("tu_") and f.endswith(".hip"))
    assert units == ["tu_%d.hip" % i for i in range(1, 7)]
    txt = open(os.path.join(ROOT, "sparrowrecsys_amd", "csrc", "tu_instances.h")).read()
    for i in range(1, 7):
        assert "SPRK_INST_%d __global__" % i in txt, "family %d has no instantiation" % i


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_generated_isa_of_every_unit_passes_the_static_checks(tmp_path):
    """[r5] What the hunt for k_dien_seq_mfma's flaky tiles left behind (k_dien_fused.h, "an open issue, fenced"; scripts/isa/): every unit's device
    ISA through (1) asm_hazards.py -- an instruction INSIDE an asm statement (the hazard recognizer does not look there) reading a register a
    transcendental, an MFMA or -- for the permlane swaps -- any VALU instruction wrote too recently; (2) isa_waitcnt_check.py -- every basic block
    replayed against in-order lgkmcnt / vmcnt queues: a register read or overwritten while the load that writes it is outstanding; and for
    the two DIEN kernels (3) isa_undef_reads.py -- a VGPR read that some path reaches without a write (only the unused halves of operand
    pairs with op_sel may show up) and (4) isa_waitcnt_paths.py -- the whole kernel replayed twice round every loop.  None of them found the
    cause; all of them are cheap and would have found several of the bugs of rounds 1 to 4."""
    env = dict(os.environ, PATH=os.environ.get("PATH", "") + ":/opt/rocm/bin")
    csrc = os.path.join(ROOT, "sparrowrecsys_amd", "csrc")
    units = ["sparrow_hip.hip"] + ["tu_%d.hip" % i for i in range(1, 7)]
    procs = []
    for u in units:
        out = str(tmp_path / (u + ".s"))
        procs.append((u, out, subprocess.Popen(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", csrc,
                                                "--cuda-device-only", "-S", os.path.join(csrc, u), "-o", out], env=env, stdout=subprocess.DEVNULL,
                                               stderr=subprocess.DEVNULL)))
    for u, out, p in procs:
        assert p.wait(timeout=900) == 0, u
    tool = lambda name: os.path.join(ROOT, "scripts", "isa", name)
    for u, out, p in procs:
        r = subprocess.run([sys.executable, tool("asm_hazards.py"), out, "."], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "0 suspicious adjacencies" in r.stdout, u + "\n" + r.stdout[-2000:]
        r = subprocess.run([sys.executable, tool("isa_waitcnt_check.py"), out, "."], capture_output=True, text=True, timeout=600)
        assert r.stdout.strip().endswith("0 violations"), u + "\n" + r.stdout[-2000:]
    tu4 = [out for u, out, p in procs if u == "tu_4.hip"][0]
    r = subprocess.run([sys.executable, tool("isa_undef_reads.py"), tu4, "k_dien_fused|k_dien_seq_mfma"], capture_output=True, text=True, timeout=600)
    flagged = [l for l in r.stdout.splitlines() if "before any write" in l]
    assert all("v_pk_" in l and "op_sel" in l for l in flagged), "\n".join(flagged[:10])
    r = subprocess.run([sys.executable, tool("isa_waitcnt_paths.py"), tu4, "k_dien_fused|k_dien_seq_mfma", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.count(" 0 violations") == 4, r.stdout[-2000:]
    # [r6] the cause of those flaky tiles (k_dien_fused.h; scripts/ubench/pkfma_opsel_mfma.hip): on gfx950 a packed-f32 VALU instruction whose LOW
    # result takes the HIGH dword of a VGPR src1 (`v_pk_fma_f32 ... op_sel:[0,1,0]`) reads that dword as 0 in lanes 48..63 while another wave of the
    # SIMD issues 16x16 MFMAs with 128-bit operands.  hipcc forms it by itself (SLP + operand folding); NO kernel of the library may carry one --
    # any kernel can share a SIMD with a wave of another launch.  (The SGPR-pair form is measured clean and only listed.)
    r = subprocess.run([sys.executable, tool("isa_pk_opsel.py")] + [out for u, out, p in procs], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "\n0 packed-f32 instructions take the high dword of a VGPR src1" in "\n" + r.stdout, r.stdout[-3000:]


def test_pk_opsel_scanner_sees_the_form(tmp_path):
    """The scanner itself: the failing build's instruction is a hit, the op_sel_hi broadcast and the SGPR-pair form are not."""
    f = tmp_path / "k.s"
    f.write_text("k_a:\n\tv_pk_fma_f32 v[4:5], v[36:37], v[0:1], v[48:49] op_sel:[0,1,0]\n\tv_pk_mul_f32 v[4:5], v[36:37], v[0:1] op_sel:[0,1]\n"
                 "k_b:\n\tv_pk_fma_f32 v[4:5], v[36:37], v[0:1], v[48:49] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 v[4:5], v[36:37], s[0:1], v[48:49] op_sel:[0,1,0]\n"
                 "\tv_pk_fma_f32 v[4:5], v[36:37], v[0:1], v[48:49] op_sel:[1,0,0]\n\tv_pk_add_f32 v[4:5], v[36:37], v[0:1] op_sel:[0,1] op_sel_hi:[1,0]\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "isa", "isa_pk_opsel.py"), str(f)], capture_output=True, text=True)
    assert r.returncode == 1, r.stdout
    assert "3 packed-f32 instructions take the high dword of a VGPR src1 for their low result (1 more from an SGPR pair)" in r.stdout, r.stdout
    assert r.stdout.count("k_a:") == 2 and r.stdout.count("VGPR src1.hi -> lo") == 3
