"""BASELINE config[3] on hardware: the REAL model rendered by 2 / 3 ranks (row tiles, one all_gather of [rays,5]
tiles) must reproduce the unsharded frame bit for bit.  The GPU box has one MI355X, so both ranks run on device 0
(MNERF_FORCE_DEVICE=0) and the collective goes through gloo — the sharding, the per-rank ray ranges, the gather
and its ragged-count handling are exactly what the nccl (RCCL) path runs; only the transport differs."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, height, width, q):
    try:
        os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port), MNERF_FORCE_DEVICE="0", MNERF_DIST_BACKEND="gloo")
        from matchnerf_amd import dist as mdist, options, synthetic as syn
        from matchnerf_amd.edict import EasyDict
        from matchnerf_amd.models import models_dict
        r, w, dev = mdist.init_from_env()
        opt = options.load_options("configs/test.yaml", verbose=False)
        opt.device = str(dev)
        opt.nerf.sample_intvs = 32
        model = models_dict[opt.model](opt).to(dev).eval()
        model.load_state_dict(syn.to_torch(syn.seeded_state_dict(syn.state_dict_spec(), 1), dev))
        scene = syn.make_scene(height, width, 3, seed=13)
        batch = EasyDict({k: torch.from_numpy(v).to(dev) for k, v in scene.items()})
        with torch.no_grad():
            # ONE set of feature maps for every rank and both renders: library convolutions / GEMMs are not bitwise
            # reproducible across processes, and the claim under test is about the sharded RENDER + gather
            feats = model.get_img_feat(batch.images[:, :3], cur_n_src_views=3)
            for f in feats:
                host = f.cpu()
                torch.distributed.broadcast(host, src=0)
                f.copy_(host.to(dev))
            model.get_img_feat = lambda *a, **k: feats
            sharded = mdist.render_frame_sharded(model, batch)
            whole = model(batch, mode="test")
        ok = all(torch.equal(sharded[k], whole[k]) for k in ("rgb", "depth", "opacity"))
        ok = ok and sharded.rgb.shape == (1, height * width, 3) and bool(torch.isfinite(sharded.rgb).all())
        mdist.barrier()
        q.put((r, bool(ok), float(whole.rgb.mean())))
        torch.distributed.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        q.put((rank, False, repr(e)))


def _run_ranks(height, width, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, height, width, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    return res


@pytest.mark.parametrize("height,width,world", [(32, 48, 2), (32, 48, 3)])  # even split; ragged one (11 + 11 + 10 rows)
def test_sharded_frame_is_bit_identical(height, width, world):
    res = _run_ranks(height, width, world)
    if any(isinstance(r[2], str) for r in res):
        # a worker died with an EXCEPTION (rendezvous port taken between _free_port() and init_process_group, ...):
        # transport trouble, not a result - one more attempt on a fresh port.  A frame mismatch is never retried.
        print("retrying after worker exception:", res)
        res = _run_ranks(height, width, world)
    assert [r[:2] for r in res] == [(r, True) for r in range(world)], res
    assert len({r[2] for r in res}) == 1


def test_bench_self_spawns_two_ranks():
    """`python bench.py --gpus 2` with no launcher (how the driver starts it) re-executes itself under
    torch.distributed.run; here both ranks share the one GPU (MNERF_FORCE_DEVICE=0, gloo transport)."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MNERF_FORCE_DEVICE="0", MNERF_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["value"] > 0 and j["steps"] == 1
    assert j["config"]["rays_per_step_per_gpu"] == 512 * 640 and "roofline" in j


def _bench(*flags, gpus):
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MNERF_FORCE_DEVICE="0", MNERF_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", str(gpus), "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline", "--no-secondary", *flags], env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_with_eight_ranks_on_one_gpu():
    """BASELINE config[3] as the driver will start it on an 8-GPU node - `bench.py --gpus 8`, both shard modes - with the eight
    ranks sharing this box's one GPU (MNERF_FORCE_DEVICE=0) over gloo: the REAL model in eight processes, one JSON line with
    n_gpus 8 and eight per-rank rows; the frame gathered from eight row bands has the bits of the frame one rank renders."""
    one = _bench(gpus=1)
    rows = _bench("--shard", "rows", gpus=8)
    views = _bench("--shard", "views", gpus=8)
    for j, scaling in ((rows, "strong"), (views, "weak")):
        assert j["n_gpus"] == 8 and j["scaling"] == scaling and j["value"] > 0
        assert len(j["config"]["per_rank_ms_per_step"]["ranks"]) == 8
    assert rows["config"]["frame_bits"] == one["config"]["frame_bits"]
    assert views["config"]["frame_bits"] != one["config"]["frame_bits"]  # eight different target poses


def _nccl_one_rank_worker(port, q):
    try:
        os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                          MNERF_DIST_INIT_ALWAYS="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        os.environ.pop("MNERF_DIST_BACKEND", None)
        from matchnerf_amd import dist as mdist
        rank, world, dev = mdist.init_from_env()
        assert torch.distributed.get_backend() == "nccl" and dev.type == "cuda"
        tile = torch.arange(11 * 5, dtype=torch.float32, device=dev).reshape(11, 5)
        full = mdist.gather_tiles(tile, always=True)                    # count exchange + all_gather_into_tensor on device tensors
        full2 = mdist.gather_tiles(tile, counts=[11], always=True)
        mdist.barrier(always=True)
        mx = mdist.max_over_ranks(3.25, dev, always=True)
        ok = torch.equal(full, tile) and torch.equal(full2, tile) and full.is_cuda and mx == 3.25
        torch.distributed.destroy_process_group()
        q.put((bool(ok), ""))
    except Exception as e:  # noqa: BLE001
        q.put((False, repr(e)))


def test_rccl_collectives_execute_on_one_gpu():
    """The nccl (= RCCL) branch of dist.py — device-tensor all_gather_into_tensor, all_reduce, barrier — has never run on
    the one-GPU boxes (every multi-rank test goes through gloo).  A ONE-rank RCCL group runs the very same calls."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_one_rank_worker, args=(_free_port(), q))
    p.start()
    ok, err = q.get(timeout=240)
    p.join(60)
    assert ok, err
