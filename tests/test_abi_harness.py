"""tests/abi_harness.c drives the C ABI without Python, the way the cgo shim (go/cometgpu) does: caller-allocated buffers,
io callbacks, pthreads. On the GPU box it must pass; without a device it must fail LOUDLY with COMET_ERR_NO_DEVICE (exit 77),
never fall back to anything."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
BIN = ROOT / "tests" / "abi_harness"


def _build():
    if not BIN.exists():
        subprocess.check_call(["gcc", "-O1", "-std=c11", "-ffp-contract=off", "-pthread", "-I", str(ROOT / "include"), str(ROOT / "tests" / "abi_harness.c"), "-o", str(BIN),
                               "-L", str(ROOT / "comet_amd"), "-lcomet_hip", "-Wl,-rpath," + str(ROOT / "comet_amd"), "-lm"])


def test_harness_builds_and_refuses_to_run_without_a_gpu():
    _build()
    r = subprocess.run([str(BIN)], capture_output=True, text=True, timeout=300)
    assert r.returncode in (0, 77), r.stderr
    if r.returncode == 77:
        assert "no gfx950 device" in r.stderr


@pytest.mark.gpu
def test_harness_passes_on_the_gpu():
    _build()
    r = subprocess.run([str(BIN)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "abi_harness OK" in r.stdout
