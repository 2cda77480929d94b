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


def run_world(case, world, tmp_path, port, timeout=300, **extra_env):
    env = dict(os.environ, COMET_RCCL_LIB=str(SHIM), HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    outs = [tmp_path / f"{case}_{world}_{r}.npz" for r in range(world)]
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "comm_worker.py"), str(r), str(world), str(port), case, str(outs[r])], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=timeout)[0])
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


@pytest.mark.skipif(not SHIM.exists(), reason="tests/libshm_rccl.so not built (__graft_entry__.build())")
def test_config3_sharded_world8_matches_oracle(ctx, tmp_path):
    """BASELINE configs[3] in its sharded form — IVFPQ d 768, nlist 4096, nprobe 32, M 96, nbits 8, K 10, inverted lists dealt over EIGHT ranks by length,
    stage 1 on the nearest list's owner, the stage-1 bounds all-reduced, per-shard top-K all-gathered and merged — against the CPU ORACLE searching the index
    the (unsharded) GPU built from the same rows (ivfpq_index_search.go:231-341 semantics; fan-out + merge as storage.go:546-626 / storage_merge.go:13-46)."""
    sys.path.insert(0, str(ROOT / "tests"))
    import comm_worker as cw
    import oracle_lib as orc
    case, world = "ivfpq_c3", 8
    kind, n, d, B, K, nb, kw = cw.CASES[case]
    nlist, M, nbits, _nt = cw.SHAPES[case]
    X, Qs = cw.data(case)
    g = cw.build(ctx, case, X, 0, 1)
    blob = g.to_bytes()
    o = orc.IVFPQ(d, "l2_squared", nlist, M, nbits)
    assert o.from_bytes(blob) == len(blob)
    want = [[o.search(q, K, kw["nprobes"]) for q in Q] for Q in Qs]
    got = run_world(case, world, tmp_path, 29821, timeout=900)
    own = got[0]["owners"]
    assert all(np.array_equal(r["owners"], own) for r in got) and len(set(own.tolist())) == world          # one placement, every rank owns lists
    per_rank = np.bincount(own, minlength=world)
    assert per_rank.min() > 0 and not np.array_equal(own, np.arange(nlist) % world)                        # dealt by length (LPT), not l % world
    for r, res in enumerate(got):
        assert str(res["errors"]) == ""
        for i in range(nb):
            gi, gs, gc = res[f"ids{i}"], res[f"sc{i}"], res[f"cn{i}"]
            for b in range(B):
                cnt, oi, os_ = want[i][b]
                assert gc[b] == cnt, (r, i, b, gc[b], cnt)
                assert np.array_equal(gi[b, :cnt], oi), (r, i, b, gi[b, :cnt], oi)
                assert np.array_equal(gs[b, :cnt].view(np.uint32), np.asarray(os_, np.float32).view(np.uint32)), (r, i, b)


@pytest.mark.skipif(not SHIM.exists(), reason="tests/libshm_rccl.so not built (__graft_entry__.build())")
def test_failing_rank_does_not_hang_its_peers(ctx, tmp_path):
    """A rank whose shard search throws after the batch was entered still joins every collective of the batch: its peers finish, every rank's counts for that
    batch are -code, the other batches are the unsharded results, and the failing rank's wait reports the error."""
    case, world = "ivfpq_fail", 4
    want = expected(ctx, case)
    got = run_world(case, world, tmp_path, 29822, COMET_TEST_FAIL_SEARCH="1:2")
    for r, res in enumerate(got):
        assert ("injected failure" in str(res["errors"])) == (r == 1), (r, str(res["errors"]))
        for i, (wi, ws, wc) in enumerate(want):
            gi, gc = res[f"ids{i}"], res[f"cn{i}"]
            if i == 1:
                assert (gc == -10).all(), (r, gc)              # -COMET_ERR_UNSUPPORTED on every rank
                continue
            assert np.array_equal(gc, wc), (r, i)
            for b in range(len(wc)):
                assert np.array_equal(gi[b, :wc[b]], wi[b, :wc[b]]), (r, i, b)


@pytest.mark.skipif(not SHIM.exists(), reason="tests/libshm_rccl.so not built (__graft_entry__.build())")
def test_ranks_that_disagree_on_the_list_placement_fail_together(tmp_path):
    got = run_world("ivfpq_owners", 2, tmp_path, 29823)
    for res in got:
        assert "disagree on which rank owns which inverted list" in str(res["error"]), str(res["error"])


@pytest.mark.skipif(not SHIM.exists(), reason="tests/libshm_rccl.so not built (__graft_entry__.build())")
@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_gpus2_launches_two_ranks_itself(tmp_path, launcher):
    """`python bench.py --gpus 2` with no launcher around it starts two ranks (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), rank 0 prints ONE line with n_gpus = 2:
    `value` from the throughput layout (a replica per rank, own query streams, no exchange), the row-sharded form with the in-library all-gather + merge beside
    it, both bit-identical to each other and rank 0's batch bit-identical to the CPU oracle. Both ranks share this box's one GPU (COMET_BENCH_DEVICE, the
    shared-memory stand-in for RCCL): the numbers mean nothing here, the path is the driver's."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(COMET_RCCL_LIB=str(SHIM), COMET_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    bench = [str(ROOT / "bench.py"), "--gpus", "2", "--legs", "flat,ivfpq", "--rows", "200000", "--steps", "4", "--warmup", "2", "--regions", "2", "--sustain-s", "0.2", "--cpu-seconds", "2"]
    # "torchrun": the driver's own launch line for N > 1 (one rank per process, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher)
    cmd = [sys.executable] + bench if launcher == "self" else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                                               "--master-port", "29871"] + bench
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak" and line["cpu_baseline"] is None
    assert line["config"]["layout"] == {"ranks_per_index_copy": 1, "index_copies": 2, "queries_per_step_all_ranks": 512, "exchange": None}
    assert line["parity"]["parity_mismatches"] == 0 and line["parity"]["parity_checked_queries"] >= 64
    sr = line["sharded_rows"]
    assert sr["scaling"] == "strong" and sr["sharding"] == "rows/2" and sr["qps"] > 0 and sr["identical_to_the_replica_results_for_batch_0"] is True
    assert line["legs"]["ivfpq"]["qps"] > 0 and line["legs"]["ivfpq_sharded"]["qps"] > 0
    assert line["legs"]["ivfpq_sharded"]["identical_to_the_replica_results_for_batch_0"] is True


@pytest.mark.skipif(not SHIM.exists(), reason="tests/libshm_rccl.so not built (__graft_entry__.build())")
def test_bench_gpus8_the_drivers_launch_line_on_one_gpu(tmp_path):
    """The driver's 8-GPU line — `torch.distributed.run --nproc-per-node 8 bench.py --gpus 8` with the legs that shard (flat, ivfpq, ivfpq10m) — as eight ranks on this
    box's one GPU over the shared-memory stand-in for RCCL, at reduced row counts: the first REAL 8-GPU run is the driver's, so everything but the xGMI timing is
    exercised here: the layouts (`value` from eight replicas, row shards / list shards with the in-library all-gather + merge beside it), configs[3]'s shape as list
    shards over all eight ranks, parity of rank 0's batch with the CPU oracle, sharded == replica results, ONE line from rank 0, done well inside ten minutes."""
    import json
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(COMET_RCCL_LIB=str(SHIM), COMET_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    bench = [str(ROOT / "bench.py"), "--gpus", "8", "--legs", "flat,ivfpq,ivfpq10m", "--rows", "200000", "--big-rows", "400000", "--steps", "4", "--warmup", "2", "--regions", "2",
             "--sustain-s", "0.2", "--cpu-seconds", "2"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29881"] + bench
    t0 = time.time()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    wall = time.time() - t0
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, len(lines)                                    # rank 0 alone prints
    line = json.loads(lines[-1])
    assert len(lines[-1]) < 6 * 1024
    assert line["n_gpus"] == 8 and line["value"] > 0 and line["scaling"] == "weak" and line["cpu_baseline"] is None
    assert line["config"]["layout"]["index_copies"] == 8 and line["config"]["layout"]["queries_per_step_all_ranks"] == 8 * 256
    assert line["parity"]["parity_mismatches"] == 0 and line["parity"]["parity_checked_queries"] >= 64
    sr = line["sharded_rows"]
    assert sr["sharding"] == "rows/8" and sr["qps"] > 0 and sr["identical_to_the_replica_results_for_batch_0"] is True
    legs = line["legs"]
    assert legs["ivfpq"]["qps"] > 0 and legs["ivfpq_sharded"]["qps"] > 0 and legs["ivfpq_sharded"]["identical_to_the_replica_results_for_batch_0"] is True
    assert "error" not in legs.get("ivfpq10m", {"error": "missing"}) and legs["ivfpq10m"]["qps"] > 0, legs.get("ivfpq10m")
    assert wall < 600, wall
