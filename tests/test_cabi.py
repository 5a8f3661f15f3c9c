"""C-ABI checks that need no GPU: the library builds for gfx950, loads, exports every symbol
include/sparrow_hip.h declares, its structs match the ctypes mirror, and plan validation rejects
malformed plans with SPRK_EINVAL (no compute calls)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from sparrowrecsys_amd import _lib as L
from sparrowrecsys_amd import models as M
from tests.conftest import ROOT


def test_header_symbols_are_exported(lib):
    header = open(os.path.join(ROOT, "include", "sparrow_hip.h")).read()
    declared = set(re.findall(r"\b(sprk_[a-z0-9_]+)\s*\(", header))
    assert declared == set(L.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_the_library_exports_exactly_the_header(lib):
    """[r6, VERDICT r05 item 7] `nm -D`: the defined dynamic symbols ARE include/sparrow_hip.h's entry points -- no helper (round 5 leaked
    split_csv_line / parse_number), no C++ template instantiation, no compiler marker."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if l.strip())
    assert exported == sorted(L.EXPORTED_SYMBOLS), sorted(set(exported) ^ set(L.EXPORTED_SYMBOLS))


def test_header_constants_match_ctypes_mirror():
    header = open(os.path.join(ROOT, "include", "sparrow_hip.h")).read()
    consts = dict(re.findall(r"#define\s+(SPRK_[A-Z_]+)\s+\(?(-?\d+)\)?", header))
    assert int(consts["SPRK_ABI_VERSION"]) == L.ABI_VERSION
    assert int(consts["SPRK_TILE_M"]) == L.TILE_M
    assert (int(consts["SPRK_MAX_SEGS"]), int(consts["SPRK_MAX_OPS"]), int(consts["SPRK_MAX_TAPS"]),
            int(consts["SPRK_MAX_PAIRS"]), int(consts["SPRK_MAX_BUFS"])) == (L.MAX_SEGS, L.MAX_OPS, L.MAX_TAPS, L.MAX_PAIRS, L.MAX_BUFS)
    assert (int(consts["SPRK_EINVAL"]), int(consts["SPRK_ERANGE"]), int(consts["SPRK_EKIND"])) == (L.EINVAL, L.ERANGE, L.EKIND)
    # struct sizes: all-int32/float members, so size = 4 * member count
    assert C.sizeof(L.Seg) == 32 and C.sizeof(L.Op) == 56 and C.sizeof(L.Tap) == 24 and C.sizeof(L.Din) == 60
    assert C.sizeof(L.Plan) == 4 * (7 + 3 + 1 + 1 + 1 + 2 * L.MAX_PAIRS + 1 + 1) + 32 * L.MAX_SEGS + 56 * L.MAX_OPS + 24 * L.MAX_TAPS + 60


def test_runtime_info_without_compute(lib):
    info = L.runtime_info()
    assert info["abi_version"] == L.ABI_VERSION
    assert info["device_count"] >= 0


def _create(lib, plan):
    h = C.c_void_p()
    rc = lib.sprk_create(C.byref(plan), C.byref(h))
    if rc == 0:
        lib.sprk_destroy(h)
    return rc


@pytest.mark.parametrize("model", [M.EmbeddingMLP, M.WideNDeep, M.NeuralCF, M.DeepFM, M.DeepFMv2, M.DIN, M.DIEN])
def test_valid_plans_pass_validation(lib, model):
    plan, _ = model(seed=1).build_plan()
    assert _create(lib, plan) == 0, lib.sprk_last_error()


def test_malformed_plans_are_rejected(lib):
    def fresh():
        return M.DeepFM(seed=1).build_plan()[0]
    p = fresh(); p.abi_version = 99
    assert _create(lib, p) == L.EINVAL and b"abi_version" in lib.sprk_last_error()
    p = fresh(); p.segs[0].dst = 10_000
    assert _create(lib, p) == L.EINVAL
    p = fresh(); p.segs[0].slot = p.n_slots
    assert _create(lib, p) == L.EINVAL
    p = fresh(); p.ops[1].K = 6
    assert _create(lib, p) == L.EINVAL
    p = fresh(); p.ops[1].N = 24
    assert _create(lib, p) == L.EINVAL
    p = fresh(); p.ops[1].dst_buf = p.ops[1].src_buf
    assert _create(lib, p) == L.EINVAL
    p = fresh(); p.taps[0].len = 100_000
    assert _create(lib, p) == L.EINVAL
    p = fresh(); p.n_bufs = 7
    assert _create(lib, p) == L.EINVAL
    p = M.DIN(seed=1).build_plan()[0]; p.din.hidden = 20
    assert _create(lib, p) == L.EINVAL
    assert lib.sprk_create(None, None) == L.EINVAL


def test_forward_before_finalize_is_state_error(lib):
    plan, _ = M.NeuralCF(seed=1).build_plan()
    h = C.c_void_p()
    assert lib.sprk_create(C.byref(plan), C.byref(h)) == 0
    try:
        rc = lib.sprk_forward(h, None, None, None, 4, None, 0, None)
        assert rc == L.ESTATE
        assert lib.sprk_forward_din(h, None, None, None, 4, None, 0, None) == L.EKIND
    finally:
        lib.sprk_destroy(h)


def test_product_fails_loudly_without_gpu(lib, samples):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no HIP device"):
        M.NeuralCF(seed=1).predict(samples)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "sparrowrecsys_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "plan_interp" not in src, f


def test_peer_allgather_arguments_are_validated_before_any_device_call():
    """sprk_peer_create's argument checks need no GPU (SPRK_EINVAL); the allocation itself fails loudly without one."""
    import ctypes as C
    from sparrowrecsys_amd import _lib as L
    lib = L.load_library()
    hd = C.create_string_buffer(64)
    h = C.c_void_p()
    for rank, world, slot in [(0, 0, 64), (2, 2, 64), (-1, 2, 64), (0, 17, 64), (0, 1, 0), (0, 1, 6)]:
        assert lib.sprk_peer_create(rank, world, slot, hd, C.byref(h)) == L.EINVAL
        assert not h.value
    assert lib.sprk_peer_allgather_scores(None, None, 0, None, None) == L.EINVAL
    assert lib.sprk_peer_connect(None, None) == L.EINVAL
    import torch
    if not torch.cuda.is_available():
        assert lib.sprk_peer_create(0, 1, 64, hd, C.byref(h)) == L.EHIP
        assert b"receive buffer" in lib.sprk_last_error()
