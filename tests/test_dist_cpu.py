"""N>1 path on CPU: world_size-2 gloo processes shard rays / views, gather tiles, and must
reproduce the unsharded result."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from matchnerf_amd import dist as mdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_render(first, n):
    """deterministic stand-in for a rendered tile: 5 floats per ray as a function of its index"""
    idx = torch.arange(first, first + n, dtype=torch.float32)
    return torch.stack([idx, idx * 0.5, idx * 0.25, idx + 1000, -idx], 1)


def _worker(rank, world, port, height, width, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, dev = mdist.init_from_env(backend="gloo")
    first, n = mdist.shard_rows(height, width, r, w)
    counts = [mdist.shard_rows(height, width, i, w)[1] for i in range(w)]
    full = mdist.gather_tiles(_fake_render(first, n), counts)
    ok = torch.equal(full, _fake_render(0, height * width))
    # without counts the (possibly ragged) row counts are exchanged first
    ok = ok and torch.equal(mdist.gather_tiles(_fake_render(first, n)), full)
    try:
        mdist.gather_tiles(_fake_render(first, n), [n + 1] * w)
        ok = False
    except ValueError:
        pass
    # view sharding (weak scaling): each rank owns whole frames
    v0, vn = mdist.shard_range(5, r, w)
    frames = torch.cat([_fake_render(v * 7, 7) for v in range(v0, v0 + vn)] + [torch.zeros(0, 5)], 0)  # a rank may own no view
    vcounts = [mdist.shard_range(5, i, w)[1] * 7 for i in range(w)]
    allf = mdist.gather_tiles(frames, vcounts)
    ok = ok and torch.equal(allf, torch.cat([_fake_render(v * 7, 7) for v in range(5)], 0))
    t = mdist.max_over_ranks(float(r + 1), dev)
    ok = ok and t == float(w)
    mdist.barrier()
    q.put((r, bool(ok)))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("height,width", [(8, 6), (7, 5)])
def test_two_rank_gloo_shard_and_gather(height, width):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, height, width, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_eight_rank_gloo_shard_and_gather_of_the_dtu_frame():
    """The first 8-GPU run should not also be the first 8-rank run of the host logic: BASELINE config[3]'s row sharding and tile
    gather at world size 8 (512 rows -> 64 per rank; a 509-row frame for the ragged case), over gloo on the CPU."""
    for height, width in ((512, 20), (509, 3)):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 8, port, height, width, q)) for r in range(8)]
        for p in procs:
            p.start()
        try:
            res = sorted(q.get(timeout=120) for _ in procs)
        finally:
            for p in procs:
                p.join(timeout=30)
                if p.is_alive():
                    p.terminate()
        assert res == [(r, True) for r in range(8)]
    spans = [mdist.shard_rows(512, 640, r, 8) for r in range(8)]
    assert [n for _, n in spans] == [64 * 640] * 8 and [f for f, _ in spans] == [64 * 640 * r for r in range(8)]


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 64, 327680):
        for world in (1, 2, 3, 8):
            spans = [mdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (b0, c0), (b1, _) in zip(spans, spans[1:]):
                assert b0 + c0 == b1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_bench_respawns_itself_under_the_launcher(monkeypatch):
    """`python bench.py --gpus N` without RANK/WORLD_SIZE (how the driver starts it) must start N ranks itself."""
    import argparse
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    import bench
    seen = {}
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    rc = bench.respawn_under_launcher(argparse.Namespace(gpus=4, steps=7, warmup=2, no_cpu_baseline=False))
    cmd = seen["cmd"]
    assert rc == 0 and cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" or "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ
    bench.respawn_under_launcher(argparse.Namespace(gpus=2, steps=1, warmup=0, no_cpu_baseline=True, shard="rows"))
    assert seen["cmd"][-3:] == ["--no-cpu-baseline", "--shard", "rows"]      # the strong-scaling mode reaches the ranks
