"""CPU: the C-ABI library builds, loads and exports every symbol include/ckr.h
declares; compute entry points fail loudly without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from checkers_mcts_amd import build, _lib
    build.build()
    return _lib.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "ckr.h")).read()
    declared = sorted(set(re.findall(r"\b(ckr_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), "libckr.so does not export " + name
    from checkers_mcts_amd import _lib
    assert sorted(_lib.EXPORTS) == declared


def test_every_entry_point_cites_a_reference_interface_and_is_in_the_integration_guide():
    """include/ckr.h: every declaration sits under a comment citing the reference lines it replaces (or says there is none);
    INTEGRATION.md's table names every entry point beside the reference interface a maintainer would bind it to."""
    hdr = open(os.path.join(ROOT, "include", "ckr.h")).read()
    guide = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    declared = sorted(set(re.findall(r"\b(ckr_[a-z_0-9]+)\s*\(", hdr)))
    assert [n for n in declared if n not in guide] == []
    assert len(re.findall(r"[A-Za-z_]+\.py:\d+", hdr)) >= 60                 # file:line citations into the reference


def test_version_and_error_string(lib):
    assert lib.ckr_version() == 130
    assert isinstance(lib.ckr_last_error(), bytes)


def test_struct_sizes_match_header():
    from checkers_mcts_amd import _lib
    assert C.sizeof(_lib.Tuple) == 288
    assert C.sizeof(_lib.GameResult) == 32
    assert C.sizeof(_lib.Stats) == 144
    assert C.sizeof(_lib.Config) == 168
    assert C.sizeof(_lib.NodeInfo) == 40


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from checkers_mcts_amd import _lib
    rc = lib.ckr_movegen_batch(None, 4, None, None, None)
    assert rc == -2
    with pytest.raises(_lib.CkrError):
        _lib.check(rc)
    assert b"no CPU fallback" in lib.ckr_last_error()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "checkers-mcts_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "ckr_oracle" not in txt and "libckr_oracle" not in txt, f
