import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


def _have_gpu() -> bool:
    try:
        from comet_amd import _lib
        import ctypes as C
        lib = _lib.load()
        n = C.c_int32()
        lib.comet_device_count(C.byref(n))
        return n.value > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def ctx():
    """A GPU context. GPU tests FAIL (not skip) when the HIP library or the device is missing."""
    from comet_amd import Context
    return Context(int(os.environ.get("COMET_TEST_DEVICE", "0")))


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib
