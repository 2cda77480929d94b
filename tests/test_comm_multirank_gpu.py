"""comm.hip at world size > 1: the ranks of a communicator run as PROCESSES sharing the one GPU of the box, the five RCCL entry points
served by tests/libshm_rccl.so (POSIX shared memory; COMET_RCCL_LIB). Everything else is the product path: comet_comm_create, the slot
ring of comet_index_search_sharded_async / _wait with three batches in flight, the all-gather of the packed per-shard blocks on the
exchange stream, merge_topk_kernel (and its global-memory form beyond 8192 candidates per query), the all-reduce barrier.
Every rank must end up with the results of the UNSHARDED index, bit for bit (reference analogue: storage_merge.go:13-46)."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
SHIM = ROOT / "tests" / "libshm_rccl.so"


def run_world(case, world, tmp_path, port):
    env = dict(os.environ, COMET_RCCL_LIB=str(SHIM), HSA_ENABLE_IPC_MODE_LEGACY="0")
    outs = [tmp_path / f"{case}_{world}_{r}.npz" for r in range(world)]
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "comm_worker.py"), str(r), str(world), str(port), case, str(outs[r])], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=300)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail(f"{case} world {world}: a rank hung\n" + "\n".join(logs))
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    return [np.load(o) for o in outs]


def expected(ctx, case):
    sys.path.insert(0, str(ROOT / "tests"))
    import comm_worker as cw
    kind, n, d, B, K, nb, kw = cw.CASES[case]
    X, Qs = cw.data(case)
    idx = cw.build(ctx, case, X, 0, 1)
    return [idx.search_batch(q, K, **kw) for q in Qs]


@pytest.mark.skipif(not SHIM.exists(), reason="tests/libshm_rccl.so not built (__graft_entry__.build())")
@pytest.mark.parametrize("case,world,port", [("flat", 2, 29811), ("flat", 4, 29812), ("flat", 8, 29813), ("flat_bigk", 4, 29814),
                                             ("ivf", 4, 29815), ("ivfpq", 2, 29816), ("ivfpq", 4, 29817), ("ivfpq_members", 2, 29818)])
def test_sharded_search_equals_unsharded(ctx, tmp_path, case, world, port):
    want = expected(ctx, case)
    got = run_world(case, world, tmp_path, port)
    for r, res in enumerate(got):
        for i, (wi, ws, wc) in enumerate(want):
            gi, gs, gc = res[f"ids{i}"], res[f"sc{i}"], res[f"cn{i}"]
            assert np.array_equal(gc, wc), (case, world, r, i, gc, wc)
            for b in range(len(wc)):
                c = int(wc[b])
                assert np.array_equal(gs[b, :c].view(np.uint32), ws[b, :c].view(np.uint32)), (case, world, r, i, b)
                if case != "ivfpq_members":       # member shards interleave scan positions: ids may differ only inside runs of equal scores
                    assert np.array_equal(gi[b, :c], wi[b, :c]), (case, world, r, i, b)
                else:
                    assert sorted(gi[b, :c].tolist()) == sorted(wi[b, :c].tolist()) or len(set(gs[b, :c].tolist())) < c, (case, world, r, i, b)
