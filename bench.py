#!/usr/bin/env python
"""bench.py — rendered rays/sec of the MatchNeRF hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full pass of the hot path over one synthetic batch of BASELINE config[1]:
``MatchNeRF.forward(batch, mode='test')`` for a 3-view 512x640 DTU-shaped scene with 64
samples/ray — the GMFlow encoder on the 3 source views plus all 327,680 target rays
(encoder INCLUDED in the timed region; render-only rate reported separately under
``config``).  Inputs are resident in HBM when the timed region starts.  Arithmetic is fp32
("f32": the 1e-4 RGB parity gate cannot be met in bf16): fp32 data, fp32
accumulation everywhere; the decoder's matrix products run either on the exact-f32 MFMA
(MNERF_DECODER_MATH=f32) or, by default, as fp32-grade products assembled from three bf16 terms
per operand on the bf16 MFMA ("bf16x6", DESIGN.md §5: error not above an fp32 FMA chain's).

N > 1: one process per GPU (RCCL); every rank encodes the shared source views and renders a
DIFFERENT target view (weak scaling: BASELINE config[3], "target views sharded across GPUs"),
then one all_gather returns every rank's [327680,5] tile to all ranks.  value = all rays of
all ranks / max-over-ranks time.

Also printed in the same JSON line:
  roofline     dominant kernel (fused decoder): FLOPs per launch / average launch duration measured
               with events on the launch stream inside the timed region.  f32 math: algorithmic
               FLOPs (258,336 + 64*S per sample) vs the 157.3 TFLOP/s f32-MFMA peak.
               bf16x6 math: the matrix FLOPs actually issued (6 bf16 products per algorithmic MLP
               MAC) vs the 2.5 PFLOP/s dense bf16 peak; the algorithmic rate is reported next to it.
  cpu_baseline the CPU oracle (a port of the reference path, pinned to it by goldens) timed
               on this host's cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

H, W, V, S, CHUNK = 512, 640, 3, 64, 4096
CPU_RAYS = 1024      # rays of the bounded CPU-baseline sample
CPU_THREADS_MAX = 16 # torch intra-op threads for the CPU baseline (more threads than this is
                     # slower on the small tensors of this path: 256 threads measured 14x slower)
F32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, chip-level table
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 (same table)
BF16_32X32X16_SUSTAINED_TFLOPS = 1860.0  # measured on MI355X: register-only stream of independent
                                         # v_mfma_f32_32x32x16_bf16 (tools/exp/bf16x6.hip, pure_kernel)
MLP_FLOPS_PER_SAMPLE = 258336  # the MLP part of SURVEY.md §8(d): runs as 6 bf16 products per MAC in bf16x6 math


def measured_traffic(launch_rays):
    """HBM-side bytes per decoder launch from the committed rocprofv3 PMC passes (bench.py cannot
    run the profiler itself); None when the launch size differs from the profiled one."""
    try:
        with open(os.path.join(REPO, "profiles", "decoder_traffic.json")) as f:
            t = json.load(f)
        return t["hbm_bytes_per_launch"] if t["launch_rays"] == launch_rays else None
    except Exception:  # noqa: BLE001
        return None


def flops_per_sample(s):
    return 258336 + 64 * s


def build_model(device):
    from matchnerf_amd import options, synthetic as syn
    from matchnerf_amd.models import models_dict
    opt = options.load_options("configs/test.yaml", verbose=False)
    opt.device = str(device)
    opt.n_src_views = V
    opt.nerf.sample_intvs = S
    opt.nerf.rand_rays_test = CHUNK
    model = models_dict[opt.model](opt).to(device).eval()
    weights = syn.seeded_state_dict(syn.state_dict_spec(n_src_views=V), 1)
    model.load_state_dict(syn.to_torch(weights, device))
    return opt, model, weights


def make_batch(device, target_shift):
    """Synthetic DTU-shaped scene; ``target_shift`` moves the target camera (one view per rank)."""
    from matchnerf_amd import synthetic as syn
    from matchnerf_amd.edict import EasyDict
    sc = syn.make_scene(H, W, V, seed=0)
    if target_shift:
        ext = sc["extrinsics"].copy()
        ext[0, -1, 0, 3] += 0.04 * target_shift  # slide the target along x (camera frame)
        sc["extrinsics"] = ext
    return sc, EasyDict({k: torch.from_numpy(v).to(device) for k, v in sc.items()})


def cpu_baseline(weights, scene, n_threads):
    """Oracle timed on the host: 1 encoder pass + 1 slice of CPU_RAYS rays, extrapolated to a
    full frame (the reference's own loop structure, matchnerf.py:145-161).  Bounded sample:
    ~10-30 s of CPU work."""
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
        t0 = time.perf_counter()
        O.render_rays(cfg, sd, idx, b["extrinsics"][0, -1, :3], b["intrinsics"][0, -1], b["near_fars"][0, -1],
                      b["extrinsics"][0, :-1, :3], b["intrinsics"][0, :-1], b["near_fars"][0, :-1], imgs, feats)
        t_chunk = time.perf_counter() - t0
    n_chunks = H * W / CPU_RAYS
    frame_s = t_enc + n_chunks * t_chunk
    return dict(value=round(H * W / frame_s, 2), unit="rays/s", cores=n_threads, kind="port",
                sample=f"oracle (torch-CPU port of the reference path): 1 encoder pass ({t_enc:.2f} s) + 1 slice of "
                       f"{CPU_RAYS} rays ({t_chunk:.2f} s), extrapolated to a {H * W}-ray frame",
                render_only_rays_per_s=round(CPU_RAYS / t_chunk, 2))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from matchnerf_amd import dist as mdist
    from matchnerf_amd import hip
    rank, world, device = mdist.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path is HIP-only (no CPU fallback)")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    hip.load()
    torch.backends.cudnn.benchmark = True  # MIOpen find mode for the backbone convolutions

    opt, model, weights = build_model(device)
    scene, batch = make_batch(device, target_shift=rank)
    n_rays = H * W

    def step(timer=None):
        model.kernel_timer = timer
        with torch.no_grad():
            out = model(batch, mode="test")
        tile = torch.cat([out.rgb[0], out.depth[0], out.opacity[0]], -1)  # [HW,5]
        full = mdist.gather_tiles(tile)                                    # RCCL all_gather (N>1)
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
    assert full.shape == (world * n_rays, 5) and bool(torch.isfinite(full).all())

    ksum = timer.summary()
    model.kernel_timer = None
    # encoder-only and render-only rates (outside the timed region, rank-local)
    with torch.no_grad():
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            feats = model.get_img_feat(batch.images[:, :V], cur_n_src_views=V)
        torch.cuda.synchronize()
        enc_ms = (time.perf_counter() - t1) / 3 * 1e3

    # secondary figure (N=1, outside the timed region): the same frame with the decoder's matrix products on
    # the exact-f32 MFMA instead of the default split-bf16 path, and the largest RGB difference between the two
    exact_f32 = None
    from matchnerf_amd.cond_nerf import decoder_math
    if world == 1 and decoder_math() == "bf16x6" and S <= 128:
        rgb_split = full[:, :3].clone()
        os.environ["MNERF_DECODER_MATH"] = "f32"
        try:
            step()  # re-packs the weight stream
            t32 = hip.KernelTimer()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(2):
                full32 = step(t32)
            torch.cuda.synchronize()
            ms32 = (time.perf_counter() - t1) / 2 * 1e3
            exact_f32 = {"rays_per_s": round(n_rays / (ms32 * 1e-3), 1), "ms_per_step": round(ms32, 3),
                         "decoder_ms_per_frame": round(t32.summary()["decoder"]["total_ms"] / 2, 3),
                         "rgb_linf_vs_default_math": float((full32[:, :3] - rgb_split).abs().max())}
        finally:
            os.environ["MNERF_DECODER_MATH"] = "bf16x6"
            model.kernel_timer = None

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * n_rays * args.steps / elapsed
        dec = ksum["decoder"]
        math = decoder_math() if S <= 128 else "f32"
        samples_launch = dec["rays"] / dec["launches"] * S
        flops_launch = samples_launch * flops_per_sample(S)
        algorithmic = flops_launch / (dec["avg_ms"] * 1e-3) / 1e12
        if math == "bf16x6":
            issued_launch = samples_launch * (6 * MLP_FLOPS_PER_SAMPLE + (flops_per_sample(S) - MLP_FLOPS_PER_SAMPLE))
            achieved, peak = issued_launch / (dec["avg_ms"] * 1e-3) / 1e12, BF16_MFMA_PEAK_TFLOPS
        else:
            issued_launch, achieved, peak = flops_launch, algorithmic, F32_MFMA_PEAK_TFLOPS
        render_ms = (dec["total_ms"] + ksum["cost_volume"]["total_ms"]) / args.steps
        line = {
            "metric": "rendered rays/sec (3-view, 64 samples/ray)", "value": round(value, 1), "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "BASELINE config[1]: DTU-shape 3-view 512x640, 64 samples/ray, full frame "
                            "(327680 rays) per step incl. GMFlow encoder; fp32 parity mode",
                "decoder_math": math if math == "f32" else
                "bf16x6: fp32 operands as 3 bf16 terms, 6 products per MAC on the bf16 MFMA, fp32 accumulate "
                "(error <= an fp32 FMA chain's; DESIGN.md section 4)",
                "exact_f32_mfma_path": exact_f32,
                "rays_per_step_per_gpu": n_rays, "kernel_launch_rays": int(dec["rays"] / dec["launches"]),
                "parallelism": f"target views x{world}" if world > 1 else "single GPU",
                "encoder_ms": round(enc_ms, 3), "render_kernels_ms_per_frame": round(render_ms, 3),
                "render_only_rays_per_s_per_gpu": round(n_rays / (render_ms * 1e-3), 1),
                "cost_volume_ms_per_frame": round(ksum["cost_volume"]["total_ms"] / args.steps, 3),
                "decoder_ms_per_frame": round(dec["total_ms"] / args.steps, 3),
            },
            "roofline": {"bound": "mfma", "kernel": "decoder_kernel<4> (fused MLP + ray transformer + compositing)",
                         "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4),
                         "math": math, "algorithmic_tflops": round(algorithmic, 2),
                         "instruction_ceiling": ({"tflops": BF16_32X32X16_SUSTAINED_TFLOPS,
                                                  "frac": round(achieved / BF16_32X32X16_SUSTAINED_TFLOPS, 4),
                                                  "what": "sustained rate of a register-only v_mfma_f32_32x32x16_bf16 "
                                                          "stream measured on this GPU model"} if math == "bf16x6" else None),
                         "issued_flops_per_launch": issued_launch,
                         "traffic": measured_traffic(int(dec["rays"] / dec["launches"])),
                         "avg_launch_ms": round(dec["avg_ms"], 4),
                         "flops_per_launch": flops_launch},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(weights, scene, min(os.cpu_count() or 1, CPU_THREADS_MAX))
        print(json.dumps(line), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
