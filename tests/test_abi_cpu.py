"""No-GPU checks of the boundary: the C-ABI library loads and exports every symbol include/metavoice_hip.h declares, the
structs have the C layout, argument validation works without touching a device, and there is no CPU fallback."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT, load_pkg

N = load_pkg("_native")


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "metavoice_hip.h")).read()
    declared = set(re.findall(r"\b(mv_[a-z0-9_]+)\s*\(", header))
    bound = {name for name, _, _ in N.SYMBOLS}
    assert declared == bound, f"header vs binding mismatch: {declared ^ bound}"
    lib = N.lib()
    for name in declared:
        assert hasattr(lib, name)
    assert b"gfx950" in lib.mv_version()


def test_struct_layouts_match_the_header():
    assert C.sizeof(N.mv_config) == 40
    assert C.sizeof(N.mv_sampling) == 32 and N.mv_sampling.seed.offset == 24


def test_argument_validation_needs_no_gpu():
    lib = N.lib()
    h = C.c_void_p()
    cfg = N.mv_config(n_layer=2, n_head=2, dim=300, intermediate=768, vocab=2562, block_size=64, spk_dim=256, norm_eps=1e-5,
                      max_utterances=1, weight_format=0)
    assert lib.mv_engine_create(C.byref(cfg), C.byref(h)) == N.MV_ERR_INVALID
    assert b"head_dim" in lib.mv_last_error()
    with pytest.raises(ValueError):
        N.check(N.MV_ERR_INVALID)
    with pytest.raises(RuntimeError):
        N.check(N.MV_ERR_HIP)
    # operational hooks tolerate a null engine instead of dereferencing it
    t = C.c_int32(7)
    assert lib.mv_decode_path(None, C.byref(t)) == -1 and t.value == 7
    assert lib.mv_last_run_ms(None) == 0.0


def test_no_cpu_fallback(monkeypatch):
    monkeypatch.setattr(N, "_lib", None)
    monkeypatch.setattr(N, "LIB_PATH", "/nonexistent/libmetavoice_hip.so")
    with pytest.raises(N.NativeLibraryMissing):
        N.lib()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "metavoice-src_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{f} imports the oracle"
