"""tests/paper_checks.c: known-answer checks on the GPU through the C ABI without Python — the hand-traced HNSW case of tests/golden/paper_kats.json (graph built by
hnsw_insert_kernel, exported, compared edge for edge; six searches), the reference's document-filter test tables and the behaviours of hnsw_index_search_test.go
(tests/golden/reference_kats_r06.json), and the lifecycle behaviours of the *_index_test.go files with the reference's error messages. On the GPU box it must pass; without a device it must fail LOUDLY (exit 77, "no HIP device"), never fall back to anything.
The same cases are asserted on the CPU oracle in test_paper_kats.py / test_reference_tables_cpu.py."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
BIN = ROOT / "tests" / "paper_checks"


def _build():
    src = ROOT / "tests" / "paper_checks.c"
    import shutil
    stale = BIN.exists() and BIN.stat().st_mtime < src.stat().st_mtime and shutil.which("gcc") is not None      # (__graft_entry__.build() makes it; a copy of the tree may reset mtimes)
    if not BIN.exists() or stale:
        subprocess.check_call(["gcc", "-O1", "-std=c11", "-Wall", "-I", str(ROOT / "include"), str(src), "-o", str(BIN),
                               "-L", str(ROOT / "comet_amd"), "-lcomet_hip", "-Wl,-rpath," + str(ROOT / "comet_amd"), "-Wl,-rpath,$ORIGIN/../comet_amd", "-lm"])


def test_paper_checks_build_and_refuse_to_run_without_a_gpu():
    _build()
    r = subprocess.run([str(BIN)], capture_output=True, text=True, timeout=300)
    assert r.returncode in (0, 77), r.stdout + r.stderr
    if r.returncode == 77:
        assert "no HIP device" in r.stderr or "gfx950" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("what,line", [("hnsw", "paper HNSW case OK"), ("filters", "document-filter tables OK"), ("searches", "HNSW search behaviours OK"),
                                       ("lifecycle", "lifecycle behaviours OK")])
def test_paper_checks_pass_on_the_gpu(what, line):
    _build()
    r = subprocess.run([str(BIN), what], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and line in r.stdout, r.stdout + r.stderr
