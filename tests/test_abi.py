"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/comet_gpu.h declares; without a GPU, context creation fails loudly (no CPU fallback)."""
import ctypes as C
import subprocess

import pytest

from comet_amd import _lib


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _lib.declared_symbols()
    assert len(declared) >= 40
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    # and nothing undeclared leaks out of the library
    out = subprocess.check_output(["nm", "-D", "--defined-only", str(_lib.LIB_PATH)], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert {s for s in exported if s.startswith("comet_")} == set(declared)


def test_version_and_error_strings():
    lib = _lib.load()
    assert b"gfx950" in lib.comet_version()
    assert isinstance(lib.comet_last_error(), bytes)


def test_no_silent_cpu_fallback():
    """On a box without a gfx950 device comet_ctx_create must fail with COMET_ERR_NO_DEVICE."""
    lib = _lib.load()
    n = C.c_int32(-1)
    assert lib.comet_device_count(C.byref(n)) == 0
    if n.value > 0:
        pytest.skip("a GPU is present; the no-device path is exercised on the CPU box")
    h = C.c_void_p()
    rc = lib.comet_ctx_create(0, C.byref(h))
    assert rc == _lib.ERR_NO_DEVICE and not h.value
    assert b"MI355X" in lib.comet_last_error()
    from comet_amd import Context, CometError
    with pytest.raises(CometError):
        Context(0)


def test_product_code_never_touches_the_oracle():
    """The shipped package must not import, link or load anything under oracle/."""
    import pathlib
    root = pathlib.Path(_lib.__file__).resolve().parent
    files = list(root.rglob("*.py")) + list((root / "csrc").glob("*.hip")) + list((root / "csrc").glob("*.hpp")) + list((root / "csrc").glob("Makefile"))
    for f in files:
        text = f.read_text()
        for needle in ("oracle_lib", "libcomet_oracle", "import oracle", "oracle/", "orc_"):
            assert needle not in text, (f, needle)
    needed = subprocess.check_output(["objdump", "-p", str(_lib.LIB_PATH)], text=True)
    assert "oracle" not in needed
