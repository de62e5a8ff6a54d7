"""CPU suite: liblgd_hip.so builds for gfx950, loads, and exports every symbol that include/lgd_hip.h
declares, with the argument count the ctypes binding uses (no compute calls — there is no GPU here)."""
import os
import re

import lgd_amd  # noqa: F401
from lgd_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_decls():
    src = open(os.path.join(ROOT, "include", "lgd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\bint\s+(lgd_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("void", "") else len([a for a in args.split(",") if a.strip()])
        decls[m.group(1)] = n
    return decls


def test_build_and_symbols():
    import __graft_entry__ as ge
    ge.build()
    lib = _lib.load()
    decls = _header_decls()
    assert len(decls) >= 20
    assert set(decls) == set(_lib.SIGNATURES), set(decls) ^ set(_lib.SIGNATURES)
    for name, nargs in decls.items():
        assert hasattr(lib, name), name
        assert len(_lib.SIGNATURES[name]) == nargs, (name, nargs, len(_lib.SIGNATURES[name]))
    assert lib.lgd_abi_version() == _lib.ABI_VERSION


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _lib.load()
    except RuntimeError as e:
        assert "no CPU/PyTorch fallback" in str(e)
    else:
        raise AssertionError("load() must raise when the HIP library is missing")
