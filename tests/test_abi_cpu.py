"""The C-ABI shared library loads without a GPU and exports every symbol include/*.h declares;
the Python binding types exactly that set; argument validation returns error codes (no compute
is launched here).  Also: the product package never imports the oracle."""
import ctypes
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(gags_\w+)\s*\(", src))
    return names


@pytest.fixture(scope="module")
def lib():
    from gags_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    from gags_amd import _lib
    decl = _declared()
    assert len(decl) >= 18
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in decl:
        assert hasattr(raw, name), f"{name} declared in include/gags_raster.h but not exported"
    assert set(_lib.SIGNATURES) == decl, "python binding and header disagree"


def test_library_exports_nothing_but_the_declared_c_abi(lib):
    """Built with -fvisibility=hidden: the dynamic symbol table holds the headers' gags_* functions and no internal
    C++ launcher (17 mangled _Z..._launch symbols used to leak)."""
    import subprocess
    from gags_amd import _lib
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = {ln.split()[-1] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] in "TW"}
    assert exported == _declared(), sorted(exported ^ _declared())


def test_abi_version_and_error_strings(lib):
    assert lib.gags_abi_version() == 2
    assert lib.gags_strerror(0) == b"ok"
    for code in (-1, -2, -3, -4):
        assert lib.gags_strerror(code) not in (b"ok", b"unknown error")
    assert lib.gags_strerror(-99) == b"unknown error"


def test_argument_validation_returns_codes_without_launching(lib):
    assert lib.gags_project_fwd(-1, *([None] * 5), 16, 16, 0.3, 0.01, 1e10, 0.0, *([None] * 5), None) == -1
    assert lib.gags_project_fwd(8, *([None] * 5), 16, 16, 0.3, 0.01, 1e10, 0.0, *([None] * 5), None) == -1
    assert lib.gags_raster_fwd(0, 4, 16, 16, *([None] * 7), 0, None, None, None, None, None, 0, None, 0, None) == -1
    assert lib.gags_raster_fwd_scratch_bytes(1000, 64, 48) >= 4 * 1000 * 260
    assert lib.gags_sort_pairs(4, 40, 0, None, None, None, None, None, 0, None) == -1
    assert lib.gags_depth_order(-1, None, None, None, None, None, 0, None) == -1
    assert lib.gags_depth_order_scratch_bytes(1000) > 2 * 1000 * 4
    assert lib.gags_sort_scratch_bytes(1000) > 1000 * 12
    assert lib.gags_scan_scratch_bytes(100000) >= 4
    assert lib.gags_cumsum_i32(-5, None, None, None, None, 0, None) == -1
    assert lib.gags_sh_fwd(4, 16, 7, None, None, None, None, None, None) == -1
    # round 4's entries: the stored-parameter projection, the permuted prefix sum, the row-list compaction, the split tiers
    assert lib.gags_project_fwd_raw(-1, *([None] * 4), 1.0, None, None, 16, 16, 0.3, 0.01, 1e10, 0.0, *([None] * 9), None) == -1
    assert lib.gags_project_fwd_raw(8, *([None] * 4), 1.0, None, None, 16, 16, 0.3, 0.01, 1e10, 0.0, *([None] * 9), None) == -1
    assert lib.gags_project_bwd_raw(8, *([None] * 4), 1.0, None, None, 16, 16, 0.3, *([None] * 9), None) == -1
    assert lib.gags_cumsum_gather_i32(-1, None, None, None, None, None, 0, None) == -1
    assert lib.gags_cumsum_gather_i32(8, None, None, None, None, None, 0, None) == -1
    assert lib.gags_compact_mask(-1, None, 0, None, None, None, 0, None) == -1
    assert lib.gags_compact_mask(8, None, 8, None, None, None, 0, None) == -1
    assert lib.gags_compact_mask_scratch_bytes(100000) >= 4 * 49
    assert lib.gags_decoder_layer_split(8, 4, 4, *([None] * 2), 4, *([None] * 2), 1, *([None] * 4), 4, 5, None) == -1   # terms = 5
    assert lib.gags_decoder_wgrad_split(8, 4, 4, None, 4, None, None, 4, None, None, None, 0, 1, None) == -1            # terms = 1


def test_missing_library_raises_loudly(monkeypatch, tmp_path):
    from gags_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libgags_hip.so"))
    with pytest.raises(_lib.GagsLibraryError):
        _lib.load()


def test_product_never_touches_the_oracle():
    for path in glob.glob(os.path.join(ROOT, "gags_amd", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.endswith((".py", ".hip", ".h", "Makefile")):
            src = open(path).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), path
            assert "libgags_oracle" not in src and "gags_oracle.c" not in src, path


def test_cpu_tensors_are_rejected_not_rerouted():
    import torch
    from gags_amd.rasterization import rasterization
    with pytest.raises(RuntimeError, match="no CPU path"):
        rasterization(torch.zeros(4, 3), torch.zeros(4, 4), torch.zeros(4, 3), torch.zeros(4), torch.zeros(4, 3),
                      torch.eye(4)[None], torch.eye(3)[None], 16, 16)
