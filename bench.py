#!/usr/bin/env python
"""bench.py — rendered rays/sec of the MatchNeRF hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one full pass of the hot path over one synthetic batch of BASELINE config[1]:
``MatchNeRF.forward(batch, mode='test')`` for a 3-view 512x640 DTU-shaped scene with 64
samples/ray — the GMFlow encoder on the 3 source views plus all 327,680 target rays
(encoder INCLUDED in the timed region; render-only rate reported separately under
``config``).  Inputs are resident in HBM when the timed region starts.  Arithmetic is fp32
("f32": the 1e-4 RGB parity gate cannot be met in bf16): fp32 data, fp32
accumulation everywhere; the decoder's matrix products run, by MNERF_DECODER_MATH, as
  f16x3  (default) fp32 operands as two range-managed fp16 terms, 3 products per MAC on the fp16 MFMA,
  bf16x6 fp32 operands as three bf16 terms, 6 products per MAC on the bf16 MFMA,
  f32    the exact-f32 MFMA;
all three are checked against the reference to the same tolerances.  A fourth, opt-in `f16` (one fp16 product per
MAC: the reduced-precision FAST mode, RGB L-inf ~4e-3 against the fp32 path) is timed under config.other_decoder_math only.

N > 1: one process per GPU over RCCL.  Started WITHOUT a launcher (``python bench.py --gpus N``)
the script re-executes itself under ``torch.distributed.run`` with N ranks on 127.0.0.1; started
under a launcher it reads RANK / LOCAL_RANK / WORLD_SIZE from the environment.  Every rank encodes
the shared source views and renders a DIFFERENT target view (weak scaling: BASELINE config[3],
"target views sharded across GPUs"), then one all_gather returns every rank's [327680,5] tile to
all ranks.  value = all rays of all ranks / max-over-ranks time.

Also in the JSON line (N = 1):
  roofline      the dominant kernel (fused decoder), SURVEY.md §8(d): ALGORITHMIC FLOPs per launch
                (258,336 + 64 S per sample) / average launch duration measured with events on the launch
                stream inside the timed region, against the ceiling of the matrix path IN USE (dense 16-bit
                MFMA peak / products per MAC: 833 TFLOP/s for f16x3, 417 for bf16x6; 157.3 for the exact-f32
                MFMA); next to it the same rate against the 2.5 PFLOP/s dense 16-bit peak, the measured MFMA-busy
                fraction and HBM-side bytes per launch from the committed PMC passes (profiles/current/decoder_counters.json,
                ignored when the kernel sources changed since).
                roofline.cost_volume: the second kernel of the frame (26 %; round 6: on the matrix pipe): busy fractions of
                the units that bind it (texture-address unit / vector L1, vector ALU) from profiles/current/cost_volume_counters.json
                (hash-guarded like the decoder's) and the same per-launch counts over THIS run's launch time.
                roofline.frame: the step as a whole and its render kernels in rays/s against the matrix-path bound and
                the vector-L1 tap bound (measured tap bytes per ray); every fraction in the line is <= 1.
  config        secondary workloads timed outside the timed region: BASELINE config[2] (Blender-like
                800x800, 128 samples/ray), config[4] (10 source views, 512x640), config[1]'s frame at 256 samples per ray
                (decoder_pp_kernel<256>) and a full Coach.train_iteration (forward, backward, clipping, AdamW step) at 64
                and at the reference's training 128 samples per ray.
  cpu_baseline  the CPU oracle (a port of the reference path, pinned to it by goldens) timed on this
                host's cores: one encoder pass + one full 4096-ray chunk, best of 3.
"""
import argparse
import json
from math import log10 as math_log10
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

H, W, V, S, CHUNK = 512, 640, 3, 64, 4096
CPU_RAYS = 4096      # one full chunk of the reference's slicing loop (matchnerf.py:145-161, rand_rays_test)
CPU_THREADS_MAX = 16  # torch intra-op threads for the CPU baseline (more threads than this is
                      # slower on the small tensors of this path: 256 threads measured 14x slower)
F32_MFMA_PEAK_TFLOPS = 157.3    # /opt/skills/guides/MI355X_MICROARCH.md, chip-level table
DENSE16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 / fp16 (same table)
HBM_PEAK_BYTES = 8.0e12
MLP_FLOPS_PER_SAMPLE = 258336   # the Linear layers of SURVEY.md §8(d); they run as 3 (f16x3) / 6 (bf16x6) products per MAC
PRODUCTS_PER_MAC = {"f16x3": 3, "bf16x6": 6, "f32": 1, "f16": 1}
# ceiling of each matrix path in ALGORITHMIC TFLOP/s: the pipe's dense peak / products per MAC
PATH_CEILING_TFLOPS = {"f16x3": DENSE16_MFMA_PEAK_TFLOPS / 3, "bf16x6": DENSE16_MFMA_PEAK_TFLOPS / 6,
                       "f32": F32_MFMA_PEAK_TFLOPS, "f16": DENSE16_MFMA_PEAK_TFLOPS}
# the matrix pipe's rate as MEASURED on this pool's boxes (tools/exp/ubench/mfma_rates.hip, profiles/current/r6_mfma_rates.log:
# a loop of nothing but independent v_mfma_f32_32x32x16_f16, two waves per SIMD: 38.9 cycles per instruction at the nominal
# clock instead of 32 = 2070 TFLOP/s chip): reported NEXT TO the guide's peak, never instead of it
MEASURED_DENSE16_TFLOPS = 2070.0
SHADER_CLOCK_HZ = 2.4e9          # ibid. ("256 CU x 2.4 GHz"); profiled passes run 1.9-2.0 GHz, so clock-based fractions are lower bounds
L1_PEAK_BYTES = 64 * 256 * SHADER_CLOCK_HZ  # vector L1 -> registers: 64 B per clock and CU


def flops_per_sample(s):
    return 258336 + 64 * s


def measured_counters(kernel_prefix, filename="decoder_counters.json", hash_name="source_hash"):
    """PMC-derived figures of a kernel from the committed rocprofv3 passes (bench.py cannot run the profiler itself):
    for the decoder HBM-side bytes per launch and the MFMA-busy fraction, for the cost volume its issue mix and the busy
    fractions of the units that bind it — None when the file is missing, `stale` when it was collected on other kernel
    sources than the ones in the tree (build hash)."""
    from matchnerf_amd.csrc import build
    now = getattr(build, hash_name)()
    try:
        with open(os.path.join(REPO, "profiles", "current", filename)) as f:
            c = json.load(f)
    except Exception:  # noqa: BLE001
        return None
    if c.get("build_hash") != now or not str(c.get("kernel", "")).startswith(kernel_prefix):
        return dict(stale=True, build_hash_profiled=c.get("build_hash"), build_hash_now=now)
    return c


def build_model(device, n_views=V, n_samples=S):
    import torch  # noqa: F401
    from matchnerf_amd import options, synthetic as syn
    from matchnerf_amd.models import models_dict
    opt = options.load_options("configs/test.yaml", verbose=False)
    opt.device = str(device)
    opt.n_src_views = n_views
    opt.nerf.sample_intvs = n_samples
    opt.nerf.rand_rays_test = CHUNK
    model = models_dict[opt.model](opt).to(device).eval()
    weights = syn.seeded_state_dict(syn.state_dict_spec(n_src_views=n_views), 1)
    model.load_state_dict(syn.to_torch(weights, device))
    return opt, model, weights


def make_batch(device, target_shift, height=H, width=W, n_views=V, **scene_kw):
    """Synthetic DTU-shaped scene; ``target_shift`` moves the target camera (one view per rank)."""
    import torch
    from matchnerf_amd import synthetic as syn
    from matchnerf_amd.edict import EasyDict
    sc = syn.make_scene(height, width, n_views, seed=scene_kw.pop("seed", 0), **scene_kw)
    if target_shift:
        ext = sc["extrinsics"].copy()
        ext[0, -1, 0, 3] += 0.04 * target_shift  # slide the target along x (camera frame)
        sc["extrinsics"] = ext
    return sc, EasyDict({k: torch.from_numpy(v).to(device) for k, v in sc.items()})


def cpu_baseline(weights, scene, n_threads):
    """Oracle timed on the host: 1 encoder pass + one full chunk of CPU_RAYS rays (best of 3), extrapolated to
    a full frame (the reference's own loop structure, matchnerf.py:145-161)."""
    import torch
    from matchnerf_amd import synthetic as syn
    from oracle import matchnerf_oracle as O
    torch.set_num_threads(n_threads)
    cfg = O.OracleConfig(n_src_views=V, sample_intvs=S)
    sd = syn.to_torch(weights)
    b = {k: torch.from_numpy(v) for k, v in scene.items()}
    imgs = b["images"][0, :V]
    with torch.no_grad():
        t0 = time.perf_counter()
        feats = O.encode_pairs(cfg, sd, imgs)
        t_enc = time.perf_counter() - t0
        idx = torch.arange(100 * W, 100 * W + CPU_RAYS)
        t_chunks = []
        for _ in range(3):
            t0 = time.perf_counter()
            O.render_rays(cfg, sd, idx, b["extrinsics"][0, -1, :3], b["intrinsics"][0, -1], b["near_fars"][0, -1],
                          b["extrinsics"][0, :-1, :3], b["intrinsics"][0, :-1], b["near_fars"][0, :-1], imgs, feats)
            t_chunks.append(time.perf_counter() - t0)
    t_chunk = min(t_chunks)
    n_chunks = H * W / CPU_RAYS
    frame_s = t_enc + n_chunks * t_chunk
    return dict(value=round(H * W / frame_s, 2), unit="rays/s", cores=n_threads, kind="port",
                sample=f"oracle (torch-CPU port of the reference path): 1 encoder pass ({t_enc:.2f} s) + one full chunk of "
                       f"{CPU_RAYS} rays, best of 3 ({', '.join(f'{t:.2f}' for t in t_chunks)} s), extrapolated to a "
                       f"{H * W}-ray frame",
                render_only_rays_per_s=round(CPU_RAYS / t_chunk, 2))


def _per_frame(ksum, name, frames):
    return round(ksum[name]["total_ms"] / frames, 3) if name in ksum else None


def secondary_workload(device, name, n_views, n_samples, height, width, bg, **scene_kw):
    """One of the other BASELINE configs at full size, outside the timed region: frame time incl. encoder."""
    import torch
    from matchnerf_amd import hip
    opt, model, _ = build_model(device, n_views, n_samples)
    model.nerf_setbg_opaque = bg
    _, batch = make_batch(device, 0, height, width, n_views, **scene_kw)
    with torch.no_grad():
        out = model(batch, mode="test")  # warm-up (MIOpen find, weight packing)
        timer = hip.KernelTimer()
        model.kernel_timer = timer
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            out = model(batch, mode="test")
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 2 * 1e3
        model.kernel_timer = None
        t0 = time.perf_counter()
        model.get_img_feat(batch.images[:, :n_views], cur_n_src_views=n_views)
        torch.cuda.synchronize()
        enc_ms = (time.perf_counter() - t0) * 1e3
    k = timer.summary()
    ok = bool(torch.isfinite(out.rgb).all())
    del model, batch, out
    torch.cuda.empty_cache()
    n = height * width
    return {"workload": name, "rays_per_s": round(n / (ms * 1e-3), 1), "ms_per_frame": round(ms, 3),
            "encoder_ms": round(enc_ms, 3), "cost_volume_ms": _per_frame(k, "cost_volume", 2),
            "decoder_ms": _per_frame(k, "decoder", 2), "fused_ray_chunk_ms": _per_frame(k, "render_fused", 2), "finite": ok}


def train_step_workload(device, n_samples=S):
    """A `Coach.train_iteration` (coach.py:215-243) at BASELINE config[1]'s geometry: zero_grad, mode='train' forward on
    rand_rays_train = 1024 random rays of the 512x640 target with `n_samples` stratified samples per ray (configs/train.yaml
    inherits sample_intvs 128 from base.yaml:48; 64 is the bench's inference count), an L2 loss, backward through the HIP ray
    chunk (mnerf_composite_backward -> mnerf_decoder_backward -> mnerf_cost_volume_backward) and the encoder (HIP transformer
    layers; the CNN's backward on the ROCm libraries), gradient clipping of the encoder and an Adam step — so the timed
    iteration includes re-packing every weight stream the forward kernels read (packing.py).  Outside the bench's timed region;
    device events around the decoder's backward."""
    import torch
    from matchnerf_amd import hip
    opt, model, _ = build_model(device, V, n_samples)
    model.train()
    opt.nerf.rand_rays_train = 1024
    opt.nerf.sample_stratified = True
    _, batch = make_batch(device, 0)
    params = [p for p in model.parameters() if p.requires_grad]
    optim = torch.optim.AdamW([dict(params=model.feat_enc.parameters(), lr=1e-7), dict(params=model.nerf_dec.parameters(), lr=1e-7)],
                              weight_decay=1e-4)  # configs/train.yaml: AdamW, weight_decay 1e-4 (tiny rates: the timing scene stays put)
    spans = []
    inner = hip.decoder_backward

    def timed_backward(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = inner(*a, **kw)
        e1.record()
        spans.append((e0, e1))
        return r

    def iteration():
        optim.zero_grad(set_to_none=True)
        out = model(batch, mode="train")
        gt = batch.images[:, -1].reshape(1, 3, -1).permute(0, 2, 1)[:, out.ray_idx]
        loss = ((out.rgb - gt) ** 2).mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.feat_enc.parameters(), 1.0)
        optim.step()
        return loss

    hip.decoder_backward = timed_backward
    try:
        torch.manual_seed(0)
        for _ in range(2):
            iteration()  # warm-up: MIOpen immediate-mode solver pick (cudnn.benchmark is off: no exhaustive find), optimizer state
        spans.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_it = 3
        for _ in range(n_it):
            loss = iteration()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n_it * 1e3
    finally:
        hip.decoder_backward = inner
    dec_ms = sum(a.elapsed_time(b) for a, b in spans) / max(len(spans), 1)
    grads_ok = all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in params)
    del model, batch, optim
    torch.cuda.empty_cache()
    return {"workload": f"train_iteration: 1024 random rays x {n_samples} stratified samples of a 512x640 target, 3 source views; "
                        "zero_grad + forward + L2 loss + backward of encoder and decoder + encoder gradient clipping + Adam step "
                        "(weight streams re-packed every iteration)",
            "sample_intvs": n_samples, "ms_per_iteration": round(ms, 3), "decoder_backward_ms": round(dec_ms, 3),
            "decoder_backward": "mnerf_decoder_backward (HIP: fp32-grade split-bf16 MFMA GEMMs by default, exact f32 with "
                                "MNERF_GEMM_MATH=f32, + per-ray attention / LayerNorm kernel)",
            "loss": float(loss.detach()), "all_gradients_finite": grads_ok}


def respawn_under_launcher(args):
    """``python bench.py --gpus N`` without a launcher: start N ranks of this script with torch.distributed.run."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup)]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    if getattr(args, "shard", "views") != "views":
        cmd += ["--shard", args.shard]
    # HSA_ENABLE_IPC_MODE_LEGACY=0: the pool's host driver only supports dmabuf IPC; without it RCCL's buffer exchange
    # between the rank processes fails with `hipIpcGetMemHandle: invalid argument` (environment note of the GPU boxes;
    # already exported there — set here only when the caller's environment lacks it)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the config[2] / config[4] timings")
    ap.add_argument("--shard", choices=["views", "rows"], default="views",
                    help="N > 1: 'views' = every rank renders its own target view (weak scaling, the default); 'rows' = ONE "
                         "frame per step, row bands sharded over the ranks (strong scaling, dist.render_frame_sharded)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn_under_launcher(args))

    import torch
    from matchnerf_amd import dist as mdist
    from matchnerf_amd import hip
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path is HIP-only (no CPU fallback)")
    if "MNERF_FORCE_DEVICE" not in os.environ and torch.cuda.device_count() < int(os.environ.get("WORLD_SIZE", "1")):
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible to this process "
                         f"(one rank per GPU; set MNERF_FORCE_DEVICE=0 MNERF_DIST_BACKEND=gloo for a dry run on one GPU)")
    rank, world, device = mdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    hip.load()
    # No library convolution is left in the inference path (csrc/conv.hip); the train-step side measurement still runs the
    # encoder's backward on MIOpen, and with find mode on its exhaustive solver search took 99 s of GPU time outside the
    # timed region (round-3 verdict): immediate mode everywhere.
    torch.backends.cudnn.benchmark = False

    opt, model, weights = build_model(device)
    rows_mode = args.shard == "rows" and world > 1
    # views: every rank renders its own target pose; rows: ALL ranks render bands of the same frame (rank 0's pose)
    scene, batch = make_batch(device, target_shift=0 if rows_mode else rank)
    n_rays = H * W

    spans = []  # per step: (start, after the render, after the gather) events of this rank

    def step(timer=None):
        model.kernel_timer = timer
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        with torch.no_grad():
            if rows_mode:  # strong scaling: one frame, this rank renders its band of rows; the gather is inside
                out = mdist.render_frame_sharded(model, batch)
                ev[1].record()
                full = torch.cat([out.rgb[0], out.depth[0], out.opacity[0]], -1)
            else:
                out = model(batch, mode="test")
                tile = torch.cat([out.rgb[0], out.depth[0], out.opacity[0]], -1)  # [HW,5]
                ev[1].record()
                full = mdist.gather_tiles(tile)                                    # RCCL all_gather (N>1)
        ev[2].record()
        if timer is not None:
            spans.append(ev)
        return full

    for _ in range(args.warmup):
        step()
    timer = hip.KernelTimer()
    mdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full = step(timer)
    torch.cuda.synchronize()
    mdist.barrier()
    elapsed = mdist.max_over_ranks(time.perf_counter() - t0, device)
    assert full.shape == ((1 if rows_mode else world) * n_rays, 5) and bool(torch.isfinite(full).all())
    frame_bits = "%012x" % (int(full.contiguous().view(torch.int32).to(torch.int64).sum()) & 0xffffffffffff)
    # per-rank milliseconds per step (frame incl. encoder | of it the gather), collected on every rank
    mine = torch.tensor([sum(a.elapsed_time(c) for a, _, c in spans) / max(len(spans), 1),
                         sum(b.elapsed_time(c) for _, b, c in spans) / max(len(spans), 1)], device=device)
    per_rank = mdist.gather_tiles(mine[None]) if world > 1 else mine[None]
    per_rank = [[round(float(x), 3) for x in row] for row in per_rank.cpu()]

    ksum = timer.summary()
    model.kernel_timer = None
    # encoder-only rate (outside the timed region, rank-local)
    with torch.no_grad():
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            model.get_img_feat(batch.images[:, :V], cur_n_src_views=V)
        torch.cuda.synchronize()
        enc_ms = (time.perf_counter() - t1) / 3 * 1e3

    # secondary figures (N=1, outside the timed region): the same frame with the other decoder matrix paths, and the
    # largest RGB difference to the default path
    from matchnerf_amd.cond_nerf import decoder_math
    math = model.nerf_dec.math_for(S)
    other_math = {}
    if world == 1:
        rgb_default = full[:, :3].clone()
        # PSNR against the target image (the synthetic scene's pseudo ground truth; metrics.py:19-41 on the whole frame)
        gt = batch.images[0, -1].permute(1, 2, 0).reshape(-1, 3)
        psnr_of = lambda rgb: float(-10.0 * torch.log10(((rgb - gt) ** 2).mean()))
        psnr_default = psnr_of(rgb_default)
        keep = os.environ.get("MNERF_DECODER_MATH")
        try:
            for m in ("f16x3", "bf16x6", "f32", "f16"):  # "f16": the reduced-precision fast mode (outside the parity gate)
                if m == math:
                    continue
                os.environ["MNERF_DECODER_MATH"] = m
                step()  # re-packs the weight stream
                tm = hip.KernelTimer()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(2):
                    fullm = step(tm)
                torch.cuda.synchronize()
                msm = (time.perf_counter() - t1) / 2 * 1e3
                other_math[m] = {"rays_per_s": round(n_rays / (msm * 1e-3), 1), "ms_per_step": round(msm, 3),
                                 "decoder_ms_per_frame": _per_frame(tm.summary(), "decoder", 2),
                                 "fused_ray_chunk_ms_per_frame": _per_frame(tm.summary(), "render_fused", 2),
                                 "rgb_linf_vs_default_math": float((fullm[:, :3] - rgb_default).abs().max()),
                                 # north_star's "PSNR delta < 0.01 dB": against the target image, and - where a deviation
                                 # weighs most - the worst case for a render 30 dB from its ground truth (mse 1e-3)
                                 "psnr_delta_db": round(psnr_of(fullm[:, :3]) - psnr_default, 5),
                                 "psnr_delta_db_at_30dB": {
                                     "what": "for a render 30 dB from its ground truth (mse 1e-3): deviation uncorrelated with the "
                                             "residual | perfectly aligned with it (upper bound)",
                                     "uncorrelated": round(10.0 * math_log10(
                                         1.0 + float(((fullm[:, :3] - rgb_default) ** 2).mean()) / 1e-3), 5),
                                     "aligned_bound": round(10.0 * math_log10(
                                         (1e-3 ** 0.5 + float(((fullm[:, :3] - rgb_default) ** 2).mean().sqrt())) ** 2 / 1e-3), 5)}}
        except Exception as e:  # noqa: BLE001  (a side measurement must not cost the headline line)
            other_math["error"] = f"{type(e).__name__}: {e}"[:300]
        finally:
            if keep is None:
                os.environ.pop("MNERF_DECODER_MATH", None)
            else:
                os.environ["MNERF_DECODER_MATH"] = keep
            model.kernel_timer = None
        assert decoder_math() == math or S > 128

    # the same frame through the one-launch (fused) form of the ray chunk: built, bit-identical, slower
    fused_form = None
    if world == 1 and math == "f16x3":
        model.fused_render = True
        try:
            ffull = step()
            tf_ = hip.KernelTimer()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(2):
                ffull = step(tf_)
            torch.cuda.synchronize()
            msf = (time.perf_counter() - t1) / 2 * 1e3
            kf = tf_.summary()
            if "render_fused" in kf:
                fused_form = {"ms_per_step": round(msf, 3), "rays_per_s": round(n_rays / (msf * 1e-3), 1),
                              "fused_ray_chunk_ms_per_frame": round(kf["render_fused"]["total_ms"] / 2, 3),
                              # (the one-launch form shares decoder_kernel's bits; the default staged decoder, the
                              # ping-pong kernel, sums layer 5 in the other order: a few ulps, tests/test_hip_kernels.py)
                              "rgb_linf_vs_staged": float((ffull[:, :3] - full[:, :3]).abs().max())}
        except Exception as e:  # noqa: BLE001
            fused_form = {"error": f"{type(e).__name__}: {e}"[:300]}
        finally:
            model.fused_render, model.kernel_timer = False, None

    secondary = []
    if world == 1 and not args.no_secondary:
        del full
        torch.cuda.empty_cache()
        def guarded(what, fn, *a, **kw):  # a failing side measurement must not cost the headline line
            try:
                secondary.append(fn(*a, **kw))
            except Exception as e:  # noqa: BLE001
                secondary.append({"workload": what, "error": f"{type(e).__name__}: {e}"[:300]})
                torch.cuda.empty_cache()

        guarded("BASELINE config[2]", secondary_workload, device,
                "BASELINE config[2]: Blender-like 3-view 800x800, 128 samples/ray, white background, full frame incl. encoder",
                3, 128, 800, 800, True, seed=31, wide=True, focal_scale=1.389, near_far=(2.0, 6.0))
        guarded("BASELINE config[4]", secondary_workload, device,
                "BASELINE config[4]: 10 source views 512x640, 64 samples/ray (45 view pairs, 1.18 GB of feature maps), "
                "full frame incl. encoder", 10, 64, 512, 640, False, seed=32)
        guarded("config[1] at 256 samples per ray", secondary_workload, device,
                "BASELINE config[1]'s frame at 256 samples per ray (configs/test_video_own.yaml: sample_intvs 256; "
                "decoder_pp_kernel<256>), full frame incl. encoder", 3, 256, 512, 640, False)
        guarded("train_iteration, sample_intvs 64", train_step_workload, device, 64)
        guarded("train_iteration, sample_intvs 128 (configs/train.yaml)", train_step_workload, device, 128)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = (1 if rows_mode else world) * n_rays * args.steps / elapsed
        fused = "render_fused" in ksum
        dec = ksum["render_fused"] if fused else ksum["decoder"]
        cv_ms = None if fused else ksum["cost_volume"]["total_ms"] / args.steps
        launch_rays = dec["rays"] / dec["launches"]
        samples_launch = launch_rays * S
        flops_launch = samples_launch * flops_per_sample(S)
        secs = dec["avg_ms"] * 1e-3
        algorithmic = flops_launch / secs / 1e12
        ppm = PRODUCTS_PER_MAC[math]
        issued_launch = samples_launch * (ppm * MLP_FLOPS_PER_SAMPLE + (flops_per_sample(S) - MLP_FLOPS_PER_SAMPLE))
        render_ms = dec["total_ms"] / args.steps + (cv_ms or 0.0)
        render_rate = n_rays / (render_ms * 1e-3)
        counters = measured_counters("decoder_") if math == decoder_math() else None
        fresh = bool(counters) and not counters.get("stale")
        # ---- the cost volume (no matrix instructions): what binds it, from the committed counter passes + the live launch time
        cv_entry = None
        cvc = measured_counters("cost_volume", "cost_volume_counters.json", "cost_volume_source_hash")
        cv_fresh = bool(cvc) and not cvc.get("stale")
        if not fused:
            cvk = ksum["cost_volume"]
            cv_secs = cvk["avg_ms"] * 1e-3
            mm_form = bool(getattr(model, "cv_matrix_form", False)) and os.environ.get("MNERF_CV_MM", "1") != "0"
            cv_entry = {
                "kernel": ("cost_volume_mm_kernel (matrix form: per 8x4-pixel tile and depth index the 128-channel bilinear "
                           "interpolation of every (pair, scale) as v_mfma_f32_32x32x16_f16 over 4x4-texel chunks of a split-fp16 "
                           "operand image; dot products, cosines, colours, masks on the vector ALU)") if mm_form else
                          "cost_volume_lean_kernel<8,false,false> (epipolar register-quad walk, group cosines, colours, masks; 0 MFMA)",
                "bound": ("vector-ALU issue (0.67 busy in the profiled pass; matrix pipe 0.24, texture-address units 0.56): the 192 "
                          "dot-product FMAs per lane and unit + the per-depth set-up; operands are requested a unit ahead and the matrix "
                          "instructions of a channel tile are interleaved with the previous tile's dot products (DESIGN.md section 4)") if mm_form
                         else "vector-memory pipeline (texture-address unit / vector L1 -> registers) and vector ALU, co-bound",
                "avg_launch_ms": round(cvk["avg_ms"], 4), "launch_rays": int(cvk["rays"] / cvk["launches"]),
                "counters_source": cvc.get("source") if cv_fresh else
                "profiles/current/cost_volume_counters.json is stale or absent: re-run tools/profile_round.sh",
            }
            if cv_fresh and mm_form != str(cvc.get("kernel", "")).startswith("cost_volume_mm"):
                cv_fresh = False  # the committed counters belong to the other kernel
                cv_entry["counters_source"] = "profiles/current/cost_volume_counters.json was measured on the other cost-volume kernel"
            if cv_fresh and cvc.get("mfma_insts_per_launch"):
                cv_entry.update({"mfma_insts_per_launch": cvc.get("mfma_insts_per_launch"), "mfma_busy_frac_measured": cvc.get("mfma_busy_frac"),
                                 "issued_matrix_tflops": round(cvc["mfma_insts_per_launch"] * 2 * 32 * 32 * 16 / cv_secs / 1e12, 1)})
            if cv_fresh:
                valu, l1b = cvc.get("valu_insts_per_launch"), cvc.get("l1_bytes_per_launch")
                cv_entry.update({
                    "ta_busy_frac_measured": cvc.get("ta_busy_frac"), "valu_busy_frac_measured": cvc.get("valu_busy_frac"),
                    "what": "busy fractions from the profiled pass: TA_TA_BUSY_sum / (256 CUs x launch cycles), 4 x "
                            "SQ_ACTIVE_INST_VALU (quad-cycles) / (1024 SIMDs x launch cycles); the *_vs_clock forms divide the "
                            "profiled per-launch counts by THIS run's launch time at the nominal 2.4 GHz",
                    "valu_insts_per_launch": valu, "vmem_rd_insts_per_launch": cvc.get("vmem_rd_insts_per_launch"),
                    "valu_issue_frac_vs_clock": round(4 * valu / (cv_secs * 1024 * SHADER_CLOCK_HZ), 4) if valu else None,
                    "l1_bytes_per_launch": l1b,
                    "l1_achieved_TBps": round(l1b / cv_secs / 1e12, 2) if l1b else None, "l1_peak_TBps": round(L1_PEAK_BYTES / 1e12, 2),
                    "l1_frac_vs_clock": round(l1b / cv_secs / L1_PEAK_BYTES, 4) if l1b else None,
                    "traffic": cvc.get("hbm_bytes_per_launch"),
                    "traffic_algorithmic_bytes": int(cvk["rays"] / cvk["launches"] * S * 96),  # the rows it writes (the maps stay in cache)
                })
        # ---- the frame: rays/s against the bounds of the units the kernels actually run on
        f_ray = S * flops_per_sample(S)
        mfma_bound = PATH_CEILING_TFLOPS[math] * 1e12 / f_ray
        frame_entry = {
            "what": "rays/s of the timed step (full frame incl. encoder, per GPU) and of its render kernels alone against (i) the "
                    "ceiling of the matrix path in use / FLOPs per ray (SURVEY.md 8d: S x (258336 + 64 S)) and (ii) the vector-L1 "
                    "bound: 64 B/clk/CU x 256 CUs x 2.4 GHz / MEASURED tap bytes per ray of the cost volume (TCP accesses x 64 B, "
                    "not the no-reuse model)",
            "full_frame_rays_per_s_per_gpu": round(n_rays / (ms_per_step * 1e-3), 1),
            "render_only_rays_per_s": round(render_rate, 1),
            "flops_per_ray": f_ray, "mfma_path_bound_rays_per_s": round(mfma_bound, 1),
            "render_only_frac_of_mfma_bound": round(render_rate / mfma_bound, 4),
            "full_frame_frac_of_mfma_bound": round(n_rays / (ms_per_step * 1e-3) / mfma_bound, 4),
        }
        if cv_entry and cv_entry.get("l1_bytes_per_launch"):
            tap_bytes_ray = cv_entry["l1_bytes_per_launch"] / cv_entry["launch_rays"]
            l1_bound = L1_PEAK_BYTES / tap_bytes_ray
            frame_entry.update({"cost_volume_l1_bytes_per_ray": int(tap_bytes_ray), "l1_tap_bound_rays_per_s": round(l1_bound, 1),
                                "render_only_frac_of_l1_tap_bound": round(render_rate / l1_bound, 4),
                                # the two stages are serial on the same CUs: time per ray at each stage's bound adds up
                                "serial_bound_rays_per_s": round(1.0 / (1.0 / mfma_bound + 1.0 / l1_bound), 1),
                                "render_only_frac_of_serial_bound": round(render_rate * (1.0 / mfma_bound + 1.0 / l1_bound), 4)})
        line = {
            "metric": "rendered rays/sec (3-view, 64 samples/ray)", "value": round(value, 1), "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong" if rows_mode else "weak", "vs_baseline": None,
            "dtype": "f16" if math == "f16" else "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE config[1]: DTU-shape 3-view 512x640, 64 samples/ray, full frame "
                            "(327680 rays) per step incl. GMFlow encoder; fp32 parity mode",
                "decoder_math": {"f16x3": "f16x3: fp32 operands as 2 range-managed fp16 terms, 3 products per MAC on the "
                                          "fp16 MFMA, fp32 accumulate (DESIGN.md section 4)",
                                 "bf16x6": "bf16x6: fp32 operands as 3 bf16 terms, 6 products per MAC on the bf16 MFMA, "
                                           "fp32 accumulate",
                                 "f32": "f32: exact-f32 MFMA",
                                 "f16": "f16: REDUCED PRECISION fast mode - one fp16 product per MAC, fp32 accumulate; outside the "
                                        "1e-4 parity gate (not the default; a headline measured in it is not the parity-mode figure)"}[math],
                "other_decoder_math": other_math or None,
                "rays_per_step_per_gpu": n_rays, "kernel_launch_rays": int(launch_rays),
                # 48-bit sum over the bit patterns of the last step's gathered output [rays, rgb | depth | opacity]: equal sums
                # <=> (almost surely) equal frames; rows mode at any rank count must give the 1-GPU frame's sum
                "frame_bits": frame_bits,
                "parallelism": (f"row bands of one frame x{world} (strong scaling)" if rows_mode else
                                f"target views x{world}") if world > 1 else "single GPU",
                "per_rank_ms_per_step": {"what": "[step incl. encoder and gather, of it the gather] per rank, device events",
                                         "ranks": per_rank},
                "encoder_ms": round(enc_ms, 3), "render_kernels_ms_per_frame": round(render_ms, 3),
                "render_only_rays_per_s_per_gpu": round(render_rate, 1),
                "ray_chunk_form": ("fused: cost volume + decoder + compositing in ONE launch per 65536 rays, conditioning rows "
                                   "produced and consumed in LDS (no HBM hand-off)") if fused else
                                  "staged: cost volume -> [rays*S, cond_stride] rows in HBM -> decoder (two launches)",
                "cost_volume_ms_per_frame": None if fused else round(cv_ms, 3),
                "decoder_ms_per_frame": None if fused else round(dec["total_ms"] / args.steps, 3),
                "fused_ray_chunk_ms_per_frame": round(dec["total_ms"] / args.steps, 3) if fused else None,
                "fused_one_launch_form": fused_form,
                "secondary_workloads": secondary or None,
            },
            "roofline": {
                "bound": "mfma",
                "kernel": ("decoder_kernel<4,64,2,1> (ONE launch: cost volume + MLP + ray transformer + compositing)" if fused
                           else "decoder_pp_kernel<64> (split-fp16 MLP + ray transformer + compositing, ping-pong teams; "
                                "decoder_kernel<4,64,2,0> with MNERF_DECODER_PP=0)" if math == "f16x3" else
                                "decoder_kernel<4,64> (fused MLP + ray transformer + compositing)"),
                "achieved": round(algorithmic, 2), "peak": round(PATH_CEILING_TFLOPS[math], 1), "unit": "TFLOP/s",
                "frac": round(algorithmic / PATH_CEILING_TFLOPS[math], 4),
                "what": "ALGORITHMIC FLOPs (SURVEY.md 8d: 258336 + 64 S per sample) / measured launch time vs the ceiling "
                        "of the matrix path the kernel runs on: dense 16-bit MFMA peak / products per MAC (f16x3: "
                        "2500/3 = 833, bf16x6: 2500/6 = 417) or the f32-MFMA peak 157.3 (f32)",
                "math": math, "algorithmic_tflops": round(algorithmic, 2),
                "frac_of_dense_16bit_peak": round(algorithmic / DENSE16_MFMA_PEAK_TFLOPS, 4),
                "frac_of_measured_matrix_rate": (round(issued_launch / secs / 1e12 / MEASURED_DENSE16_TFLOPS, 4) if math != "f32" else None),
                "measured_matrix_rate_what": "the matrix FLOPs actually issued / launch time against 2070 TFLOP/s, the rate a loop of "
                                             "independent v_mfma_f32_32x32x16_f16 reaches on these boxes (profiles/current/"
                                             "r6_mfma_rates.log): the trunk's matrix phases run at that rate (DESIGN.md section 9)",
                "mfma_pipe_frac": round(issued_launch / secs / 1e12 / DENSE16_MFMA_PEAK_TFLOPS, 4) if math != "f32"
                else round(algorithmic / F32_MFMA_PEAK_TFLOPS, 4),
                "mfma_pipe_what": f"matrix FLOPs actually issued ({ppm} products per MLP MAC) / launch time vs the "
                                  f"{'2.5 PFLOP/s dense 16-bit' if math != 'f32' else '157.3 TFLOP/s f32'} MFMA peak",
                "mfma_busy_measured": counters.get("mfma_busy_frac") if fresh else None,
                "traffic": counters.get("hbm_bytes_per_launch") if fresh else None,
                "traffic_algorithmic_bytes": (int(launch_rays * 296) if fused else  # 8d: compulsory bytes per ray
                                              int(samples_launch * 96 + launch_rays * 20)),
                "counters_source": (counters.get("source") if fresh else
                                    ("profiles/current/decoder_counters.json is stale or absent: re-run tools/profile_round.sh" if
                                     counters is None or counters.get("stale") else None)),
                "avg_launch_ms": round(dec["avg_ms"], 4), "flops_per_launch": flops_launch,
                "issued_flops_per_launch": issued_launch,
                "cost_volume": cv_entry,
                "frame": frame_entry,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(weights, scene, min(os.cpu_count() or 1, CPU_THREADS_MAX))
        print(json.dumps(line), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
