"""N>1 path on CPU: two gloo ranks shard the channels, rank 0 designs the coefficient blob and
broadcasts it (the path's only collective), every rank ends up with the same coefficients."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_channels_partitions_exactly():
    from t41_sdr_amd.dist import shard_channels
    for n, w in ((4096, 8), (4096, 3), (7, 4), (1, 1), (5, 8)):
        spans = [shard_channels(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_channels(10, 3, 3)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import t41_sdr_amd as T
    from t41_sdr_amd.dist import broadcast_coeffs, max_over_ranks, shard_channels
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank 0 owns the "filter change"; the others start from different params
        p = T.default_params(mode=0, FLoCut=300, FHiCut=2400) if rank == 0 else T.default_params()
        mine = T.design_coeffs(p)
        got = broadcast_coeffs(mine, src=0)
        lo, hi = shard_channels(4096, rank, world)
        t = max_over_ranks(1.0 + rank)
        q.put((rank, got.tobytes(), lo, hi, t))
    finally:
        dist.destroy_process_group()


def test_two_rank_coefficient_broadcast(built):
    import torch.multiprocessing as mp
    import t41_sdr_amd as T
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = T.design_coeffs(T.default_params(mode=0, FLoCut=300, FHiCut=2400)).tobytes()
    assert res[0][1] == want and res[1][1] == want
    assert (res[0][2], res[0][3]) == (0, 2048) and (res[1][2], res[1][3]) == (2048, 4096)
    assert res[0][4] == 2.0 and res[1][4] == 2.0


def _run_bench(args, env_extra=None, timeout=240):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    return p, lines


def test_bench_launcher_starts_its_own_ranks(built):
    """`python bench.py --gpus 2` with no torch.distributed environment: the parent starts two rank
    processes itself (the driver's command shape), they rendezvous (gloo in --dry-run), shard the
    channels with t41_sdr_amd.dist, broadcast rank 0's coefficient blob, and rank 0 prints ONE JSON
    line with the world size it observed"""
    p, lines = _run_bench(["--gpus", "2", "--dry-run", "--workload", "ssb"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    d = lines[0]
    assert d["dry_run"] is True and d["n_gpus"] == 2 and d["world_size_observed"] == 2
    assert d["channels"] == [0, 4096] and d["wall_max"] == 2.0
    assert sorted(d["local_ranks"]) == [0, 1]  # every rank would select its own device


def test_bench_refuses_a_world_size_mismatch(built):
    """--gpus must equal what the process group reports: one rank launched under a 1-rank environment
    but told --gpus 2 fails instead of printing a line that claims two GPUs"""
    p, lines = _run_bench(["--gpus", "2", "--dry-run"], env_extra=dict(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert p.returncode != 0 and not lines
    assert "process group has 1 ranks" in (p.stderr + p.stdout)


def test_bench_launcher_propagates_a_failing_rank(built):
    p, lines = _run_bench(["--gpus", "2", "--dry-run", "--workload", "ssb"], env_extra=dict(T41RX_BENCH_FAIL_RANK="1"))
    assert p.returncode != 0 and not lines


def test_bench_refuses_two_ranks_on_one_device(built):
    """a launcher that hands both ranks the same device would read as perfect scaling: every rank's device is gathered
    and a duplicate ends the run (the real run gathers the devices' uuids the same way, bench.py main())"""
    p, lines = _run_bench(["--gpus", "2", "--dry-run", "--workload", "ssb"], env_extra=dict(T41RX_BENCH_FORCE_LOCAL_RANK="0"))
    assert p.returncode != 0 and not lines
    assert "same device" in (p.stderr + p.stdout)


def test_bench_launcher_three_ranks(built):
    """an odd world size through the same path (rendezvous, shard, broadcast, max over ranks, one line)"""
    p, lines = _run_bench(["--gpus", "3", "--dry-run", "--workload", "ssb_agc"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 3 and d["world_size_observed"] == 3 and d["wall_max"] == 3.0
    assert d["channels"] == [0, 4096] and sorted(d["local_ranks"]) == [0, 1, 2]
