"""Round-3 GPU probe for the co-resident fused-workgroup race.

Runs the one-launch ray chunk (decoder_kernel<4,64,2,1>) of the bench frame with TWO workgroups per CU (the
debug library built with -DMNERF_FUSED_DEBUG: MNERF_FDBG_LDS_KB=0 asks for the natural 68 KiB footprint) under a
set of debug flags, compares every launch with the staged form bit for bit, and for every wrong ray reports
  * whether the conditioning rows the trunk READ (dumped from its registers) differ from the stand-alone cost volume,
  * which samples' (rgb, sigma) differ,
  * which workgroup / CU / XCD / wave slot ran the tile.
usage: MNERF_LIB=matchnerf_amd/libmnerf_hip_fdbg.so python tools/exp/race_probe.py [frames]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from matchnerf_amd import camera, hip  # noqa: E402

dev = torch.device("cuda:0")
FRAMES = int(sys.argv[1]) if len(sys.argv) > 1 else 6
S = 64
CHUNK = 65536

opt, model, _ = bench.build_model(dev)
_, batch = bench.make_batch(dev, 0)
with torch.no_grad():
    feats = model.get_img_feat(batch.images[:, :3], cur_n_src_views=3)
tgt_pose, ref_poses = model.extract_poses(batch)
ref_images = batch.images[:, :3]
ref_host, images_cl = model._frame_ctx(ref_poses, ref_images)
tgt_ex, tgt_in, tgt_nf = model._tgt_host(tgt_pose)
sc = model._scene(0, ref_host, feats, images_cl)
dec = model._decoder(S, dev)
kinv, c2w = camera.target_ray_consts(tgt_ex[0], tgt_in[0], True)
H, W = bench.H, bench.W
n_rays = H * W
CS = dec.cond_stride


def rays_of(c):
    m = min(CHUNK, n_rays - c)
    return hip.make_rays(m, S, H, W, kinv, c2w, tgt_nf[0, 0], tgt_nf[0, 1], ray_begin=c, legacy=True), m


# ---- staged reference per launch (twice: the reference itself must be reproducible)
staged = []
for c in range(0, n_rays, CHUNK):
    rays, m = rays_of(c)
    cond = hip.cost_volume(sc, rays, CS).clone()
    a = hip.decoder_chunk(dec, sc.views[0], rays, cond, want_samples=True)
    b = hip.decoder_chunk(dec, sc.views[0], rays, hip.cost_volume(sc, rays, CS), want_samples=True)
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert torch.equal(x, y), "staged form not reproducible"
    staged.append((cond.reshape(m * S, CS), [t.clone() for t in a]))
print(f"staged reference: {len(staged)} launches, reproducible", flush=True)

rows = torch.zeros(CHUNK * S, 32, device=dev)
nv = torch.zeros(CHUNK * S, device=dev)
tile_info = torch.zeros(CHUNK // 2 * 4, dtype=torch.int32, device=dev)
rgbs = torch.zeros(CHUNK, S, 3, device=dev)
sig = torch.zeros(CHUNK, S, device=dev)
os.environ["MNERF_FDBG_ROWS"] = str(rows.data_ptr())
os.environ["MNERF_FDBG_NV"] = str(nv.data_ptr())
os.environ["MNERF_FDBG_TILE"] = str(tile_info.data_ptr())
os.environ["MNERF_FDBG_RGBS"] = str(rgbs.data_ptr())
os.environ["MNERF_FDBG_SIGMA"] = str(sig.data_ptr())


def hwid(x):
    x = int(x) & 0xffffffff
    return dict(wave=x & 15, simd=(x >> 4) & 3, pipe=(x >> 6) & 3, cu=(x >> 8) & 15, sh=(x >> 12) & 1, se=(x >> 13) & 7,
                raw=hex(x))


def run_config(name, lds_kb, flags, frames=FRAMES, verbose=6):
    os.environ["MNERF_FDBG_LDS_KB"] = str(lds_kb)
    os.environ["MNERF_FDBG_FLAGS"] = str(flags)
    bad_rays = bad_launches = 0
    shown = 0
    rows_bad_total = trunk_only_total = 0
    for f in range(frames):
        for li, c in enumerate(range(0, n_rays, CHUNK)):
            rays, m = rays_of(c)
            out = [torch.full((m, 3), -1.0, device=dev), torch.full((m,), -1.0, device=dev), torch.full((m,), -1.0, device=dev)]
            rows.zero_(), nv.zero_(), rgbs.zero_(), sig.zero_()
            hip.render_chunk(sc, dec, rays, None, *out, fused=True)
            torch.cuda.synchronize()
            cond_ref, (rgb_r, dep_r, opa_r, rgbs_r, sig_r) = staged[li]
            d = (out[0] != rgb_r).any(1) | (out[1] != dep_r) | (out[2] != opa_r)
            nb = int(d.sum())
            if nb == 0:
                continue
            bad_launches += 1
            bad_rays += nb
            # rows the trunk consumed vs the stand-alone cost volume (first CS columns; NaN-safe compare on bits)
            rr = rows[:m * S, :CS].view(torch.int32) != cond_ref.view(torch.int32)
            row_bad = rr.any(1).reshape(m, S)
            nvr = cond_ref[:, 19:22].sum(1) if CS == 24 else None
            samp_bad = ((rgbs[:m] != rgbs_r).any(2) | (sig[:m] != sig_r))
            for r in d.nonzero().flatten().tolist():
                rb = row_bad[r].nonzero().flatten().tolist()
                sb = samp_bad[r].nonzero().flatten().tolist()
                rows_bad_total += bool(rb)
                trunk_only_total += (not rb)
                if shown < verbose:
                    shown += 1
                    t = r // 2
                    ti = tile_info[t * 4:t * 4 + 4].tolist()
                    cols = rr.reshape(m, S, CS)[r].any(0).nonzero().flatten().tolist()
                    print(f"  [{name}] frame {f} launch {li} ray {r} (tile {t}, block {ti[0]}, hw {hwid(ti[1])}, xcc {ti[2] & 15}): "
                          f"|d rgb|={float((out[0][r] - rgb_r[r]).abs().max()):.2e}  rows differ at samples {rb[:12]} cols {cols[:12]}"
                          f"  per-sample outputs differ at {sb[:16]}{'..' if len(sb) > 16 else ''} ({len(sb)})", flush=True)
                    if rb:
                        j = rb[0]
                        got = rows[r * S + j, :CS].tolist()
                        ref = cond_ref[r * S + j].tolist()
                        print("      row got", [f"{v:.5f}" for v in got])
                        print("      row ref", [f"{v:.5f}" for v in ref], flush=True)
                    if nvr is not None:
                        dn = (nv[:m * S].reshape(m, S)[r] != nvr.reshape(m, S)[r]).nonzero().flatten().tolist()
                        if dn:
                            print(f"      n_valid differs at samples {dn[:8]}")
    print(f"{name}: lds={lds_kb or 68.25} KiB flags={flags}: {bad_rays} wrong rays in {bad_launches} launches over {frames} frames"
          f" (rows wrong for {rows_bad_total} rays, rows right but outputs wrong for {trunk_only_total})", flush=True)
    return bad_rays


PARTNER = os.environ.get("PROBE_PARTNER", "decoder")
outs_dec32 = []
dec32 = None
if PARTNER in ("f32dec", "bf16dec"):
    keep_default_pack = model._dec()._packed  # `dec` points into these tensors: keep them alive across the re-pack
    os.environ["MNERF_DECODER_MATH"] = {"f32dec": "f32", "bf16dec": "bf16x6"}[PARTNER]
    dec32 = model._decoder(S, dev)
    os.environ.pop("MNERF_DECODER_MATH")


if PARTNER.startswith("micro"):
    import ctypes as C_
    MICRO = C_.CDLL(os.path.join(ROOT, "tools", "exp", "liblockstep.so"))
    micro_sink = torch.zeros(16, device=dev)


CVDBG = None
if os.environ.get("PROBE_CVDBG"):
    CVDBG = torch.zeros(16 + 64 * 40, dtype=torch.int32, device=dev)
    os.environ["MNERF_CVDBG_PTR"] = str(CVDBG.data_ptr())


def concurrent_streams(rounds=6):
    """E1: the STAND-ALONE cost volume on one stream while the staged decoder (LDS-DMA weight pipeline) runs on another,
    so that workgroups of both kernels share CUs: are the cost volume's rows / the decoder's outputs still exact?"""
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    bad_cv = bad_dec = 0
    detail, forensic = [], []
    for it in range(rounds):
        for li, c in enumerate(range(0, n_rays, CHUNK)):
            rays, m = rays_of(c)
            cond_ref, (rgb_r, dep_r, opa_r, rgbs_r, sig_r) = staged[li]
            outs_cv, outs_dec = [], []
            torch.cuda.synchronize()
            for rep in range(3):  # interleave launches on the two streams
                with torch.cuda.stream(s1):
                    buf = torch.full((m * S * CS,), float("nan"), device=dev)  # a store that never happened shows as NaN
                    outs_cv.append(hip.cost_volume(sc, rays, CS, out=buf, stream=s1))
                with torch.cuda.stream(s2):
                    if PARTNER == "decoder":
                        outs_dec.append(hip.decoder_chunk(dec, sc.views[0], rays, cond_ref.reshape(-1), stream=s2))
                    elif PARTNER == "cv":   # the same kernel on the second stream
                        outs_cv.append(hip.cost_volume(sc, rays, CS, stream=s2))
                    elif PARTNER.startswith("micro"):  # tools/exp/lockstep.hip: one instruction kind in a loop
                        assert MICRO.mfma_partner_launch(int(PARTNER[5:]), 40000, C_.c_void_p(micro_sink.data_ptr()), 1024,
                                                         C_.c_void_p(s2.cuda_stream)) == 0
                    elif PARTNER in ("f32dec", "bf16dec"):  # another matrix path of the decoder (other instruction mix, also LDS-DMA)
                        outs_dec32.append(hip.decoder_chunk(dec32, sc.views[0], rays, cond_ref.reshape(-1), stream=s2))
            torch.cuda.synchronize()
            for o in outs_cv:
                o2 = o.reshape(m * S, CS)
                d = (o2.view(torch.int32) != cond_ref.view(torch.int32))
                nb = int(d.any(1).sum())
                if nb:
                    bad_cv += nb
                    rws = d.any(1).nonzero().flatten()[:6].tolist()
                    detail.append(("cv", it, li, [(r // S, r % S, d[r].nonzero().flatten().tolist()) for r in rws]))
                    for r in rws[:2]:   # where do the wrong values come from?  look for them in neighbouring reference rows
                        cols = d[r].nonzero().flatten()
                        got = o2[r, cols]
                        ray, j = r // S, r % S
                        hits = []
                        for dr in range(-48, 49):
                            for dj in range(-3, 4):
                                rr_, jj = ray + dr, j + dj
                                if (dr or dj) and 0 <= rr_ < m and 0 <= jj < S and torch.equal(cond_ref[rr_ * S + jj, cols], got):
                                    hits.append((dr, dj))
                        if len(forensic) < 24:
                            forensic.append((ray, j, cols.tolist()[:3], [f"{v:.5f}" for v in got.tolist()[:3]],
                                             [f"{v:.5f}" for v in cond_ref[r, cols].tolist()[:3]], "same as ref rows (dray,dsample): " + str(hits[:4])))
            for o in outs_dec:
                nb = int(((o[0] != rgb_r).any(1) | (o[1] != dep_r) | (o[2] != opa_r)).sum())
                bad_dec += nb
    flat = [x for d in detail for x in d[3]]
    slot_hist = [sum(1 for r, _, _ in flat if r % 4 == k) for k in range(4)]
    kinds = {}
    for _, _, cols in flat:
        k = "scale0" if cols == [0, 1] else ("scale1" if cols == list(range(2, 10)) else str(cols))
        kinds[k] = kinds.get(k, 0) + 1
    if CVDBG is not None:
        torch.cuda.synchronize()
        n = int(CVDBG[0])
        print(f"in-kernel duplicate-load probe: {n} events (taps whose two loads differ)")
        rec = CVDBG[16:16 + 40 * min(n, 64)].reshape(-1, 40)
        fl = rec[:, 9:37].view(torch.float32)
        for i in range(min(n, 16)):
            r = rec[i].tolist()
            print(f"   lane {r[0]} ray {r[1]} sample {r[2]} view {r[3]} differing taps mask {r[4]:04b} texels {r[5:9]} hw {hwid(r[38])} block {r[37]}")
            print("      first :", [f"{v:.5f}" for v in fl[i, :12].tolist()])
            print("      second:", [f"{v:.5f}" for v in fl[i, 12:24].tolist()], " weights", [f"{v:.3f}" for v in fl[i, 24:28].tolist()], flush=True)
    print(f"E1 concurrent streams (stand-alone cost volume || {PARTNER}) cv_variant={os.environ.get('MNERF_CV_VARIANT', '3')} lib={os.path.basename(hip.lib_path())}: "
          f"{bad_cv} wrong cost-volume rows, {bad_dec} wrong decoder rays over {rounds} rounds x 5 launches x 3; "
          f"sampled rows by ray%4 {slot_hist}, by kind {kinds}", flush=True)
    for d in detail[:4]:
        print("   ", d, flush=True)
    for f in forensic:
        print("    forensic (ray, sample, cols, got, ref):", f, flush=True)


def lockstep_probe():
    """E2: a self-checking kernel (tools/exp/lockstep.hip: every operation class evaluated twice, lanes whose two
    results differ are counted) alone and next to the staged decoder on another stream."""
    import ctypes as C
    lib = C.CDLL(os.path.join(ROOT, "tools", "exp", "liblockstep.so"))
    n_tex = 40960
    data = torch.empty(n_tex * 128, device=dev)
    vp = C.c_void_p
    lib.lockstep_fill_data(vp(data.data_ptr()), C.c_int64(data.numel()), vp(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    rays, m = rays_of(0)
    cond_ref = staged[0][0]
    names = ["global loads", "LDS write->read_b128", "ds_bpermute", "div+sqrt", "DPP row sums", "packed fma", "LDS atomics"]

    def run(sections, with_decoder, iters=1500, grid=2048):
        counts = torch.zeros(8 * 64, dtype=torch.int32, device=dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        torch.cuda.synchronize()
        ev[0].record()
        s1.wait_event(ev[0]), s2.wait_event(ev[0])
        ev[1].record(s1)
        for rep in range(6):  # launches alternate between the streams so that both queues always hold work
            if with_decoder:
                with torch.cuda.stream(s2):
                    hip.decoder_chunk(dec, sc.views[0], rays, cond_ref.reshape(-1), stream=s2)
                    hip.decoder_chunk(dec, sc.views[0], rays, cond_ref.reshape(-1), stream=s2)
            with torch.cuda.stream(s1):
                rc = lib.lockstep_launch(vp(data.data_ptr()), n_tex, iters, vp(counts.data_ptr()), C.c_uint(sections), grid,
                                         vp(s1.cuda_stream))
                assert rc == 0, rc
        ev[2].record(s1), ev[3].record(s2)
        torch.cuda.synchronize()
        c = counts[:7 * 64].reshape(7, 4, 16).sum(2).tolist()
        run.span = (round(ev[0].elapsed_time(ev[2]), 2), round(ev[0].elapsed_time(ev[3]), 2))
        return c

    for with_dec in (False, True):
        c = run(0x7f, with_dec)
        print(f"E2 lock-step self-check, all sections, {'next to the decoder' if with_dec else 'alone'}: mismatches by section and "
              f"lane quarter [0-15,16-31,32-47,48-63]:", {names[i]: c[i] for i in range(7) if sum(c[i])} or "none",
              f"(stream spans ms: lockstep {run.span[0]}, decoder {run.span[1]})", flush=True)
    for i in range(7):
        c = run(1 << i, True)
        print(f"E2 section '{names[i]}' only, next to the decoder: {c[i]} (spans {run.span})", flush=True)


def sentinel_probe():
    """E3: parked register patterns (tools/exp/lockstep.hip: sentinel_kernel) checked after idle / waiting phases, alone and
    next to the f16x3 decoder: a register that changed was written by another wave."""
    import ctypes as C
    lib = C.CDLL(os.path.join(ROOT, "tools", "exp", "liblockstep.so"))
    vp = C.c_void_p
    data = torch.rand(1 << 22, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    rays, m = rays_of(0)
    cond_ref = staged[0][0]
    idle = ["s_sleep", "parked on vmcnt", "parked on lgkmcnt", "s_nop trains"]
    for variant in (0, 1):
        for mode in range(4):
            for with_dec in (False, True):
                out_v = torch.zeros(128 * 64, dtype=torch.int32, device=dev)
                out_s = torch.zeros(128, dtype=torch.int32, device=dev)
                first = torch.zeros(64, dtype=torch.int32, device=dev)
                torch.cuda.synchronize()
                for rep in range(4):
                    if with_dec:
                        with torch.cuda.stream(s2):
                            hip.decoder_chunk(dec, sc.views[0], rays, cond_ref.reshape(-1), stream=s2)
                            hip.decoder_chunk(dec, sc.views[0], rays, cond_ref.reshape(-1), stream=s2)
                    with torch.cuda.stream(s1):
                        rc = lib.sentinel_launch(vp(data.data_ptr()), data.numel(), 4000, mode, vp(out_v.data_ptr()),
                                                 vp(out_s.data_ptr()), vp(first.data_ptr()), 1024, variant, vp(s1.cuda_stream))
                        assert rc == 0
                torch.cuda.synchronize()
                nv_bad, ns_bad = int(out_v.sum()), int(out_s.sum())
                msg = f"E3 sentinel variant {variant} ({'registers' if variant == 0 else 'forced SGPR parking in VGPR lanes'}), idle={idle[mode]}, " \
                      f"{'next to the f16x3 decoder' if with_dec else 'alone'}: {nv_bad} vector-register mismatches, {ns_bad} scalar mismatches"
                if nv_bad:
                    ov = out_v.reshape(128, 4, 16).sum(2)
                    regs = ov.sum(1).nonzero().flatten().tolist()
                    msg += f"; by lane quarter {ov.sum(0).tolist()}, first-bad regs {regs[:12]}"
                    f = first[8:8 + 6 * min(int(first[0]), 8)].reshape(-1, 6).tolist()
                    msg += "; samples (reg, lane, got, want, block, it): " + str([(a, b, hex(c & 0xffffffff), hex(d & 0xffffffff), e, g) for a, b, c, d, e, g in f[:4]])
                if ns_bad:
                    msg += f"; scalar slots {out_s.nonzero().flatten().tolist()[:16]}"
                print(msg, flush=True)


def mask_probe():
    """E4: VALU-written lane masks consumed by the scalar unit (v_cmp -> s_and_saveexec) in a victim kernel, next to a
    partner kernel that issues nothing but ONE kind of matrix instruction."""
    import ctypes as C
    lib = C.CDLL(os.path.join(ROOT, "tools", "exp", "liblockstep.so"))
    vp = C.c_void_p
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    sink = torch.zeros(16, device=dev)
    kinds = {-1: "no partner", 0: "v_mfma_f32_32x32x16_f16", 4: "v_mfma_f32_32x32x16_bf16", 2: "v_mfma_f32_16x16x32_f16",
             1: "v_mfma_f32_32x32x8_f16", 6: "v_mfma_f32_16x16x16_f16", 3: "v_mfma_f32_32x32x2_f32", 5: "v_pk_fma_f32 only"}
    for kind, name in kinds.items():
        counts = torch.zeros(3 * 64, dtype=torch.int32, device=dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        torch.cuda.synchronize()
        ev[0].record()
        s1.wait_event(ev[0]), s2.wait_event(ev[0])
        for rep in range(3):
            if kind >= 0:
                with torch.cuda.stream(s2):
                    assert lib.mfma_partner_launch(kind, 150000, vp(sink.data_ptr()), 1024, vp(s2.cuda_stream)) == 0
            with torch.cuda.stream(s1):
                assert lib.mask_victim_launch(30000, vp(counts.data_ptr()), 2048, vp(s1.cuda_stream)) == 0
                assert lib.mask_victim_launch(30000, vp(counts.data_ptr()), 2048, vp(s1.cuda_stream)) == 0
        ev[1].record(s1), ev[2].record(s2)
        torch.cuda.synchronize()
        c = counts.reshape(3, 4, 16).sum(2).tolist()
        print(f"E4 mask victim next to [{name}]: lanes with a wrong count, by quarter [0-15,16-31,32-47,48-63]: branch on a per-lane "
              f"predicate {c[0]}, branch on a slot-uniform compare {c[1]}, v_cndmask {c[2]}  (stream spans ms: victim "
              f"{ev[0].elapsed_time(ev[1]):.1f}, partner {ev[0].elapsed_time(ev[2]):.1f})", flush=True)


def pk_probe():
    """E6: chains of ONE packed-fp32 instruction form, evaluated twice, next to the f16x3 decoder / a pure-MFMA partner."""
    import ctypes as C
    lib = C.CDLL(os.path.join(ROOT, "tools", "exp", "liblockstep.so"))
    vp = C.c_void_p
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    sink = torch.zeros(16, device=dev)
    rays, m = rays_of(0)
    cond_ref = staged[0][0]
    forms = ["v_pk_fma_f32", "v_pk_fma_f32 op_sel_hi:[1,0,1]", "v_pk_mul_f32", "v_pk_add_f32", "v_fma_f32", "v_pk_mul_f32 op_sel_hi:[1,0]"]
    for partner in ("none", "f16x3 decoder", "micro v_mfma_f32_32x32x16_f16", "micro v_mfma_f32_32x32x2_f32"):
        for form, fname in enumerate(forms):
            counts = torch.zeros(64, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            for rep in range(4):
                with torch.cuda.stream(s2):
                    if partner == "f16x3 decoder":
                        hip.decoder_chunk(dec, sc.views[0], rays, cond_ref.reshape(-1), stream=s2)
                        hip.decoder_chunk(dec, sc.views[0], rays, cond_ref.reshape(-1), stream=s2)
                    elif partner.startswith("micro"):
                        lib.mfma_partner_launch(0 if "x16_f16" in partner else 3, 60000, vp(sink.data_ptr()), 1024, vp(s2.cuda_stream))
                with torch.cuda.stream(s1):
                    assert lib.pk_victim_launch(form, 3000, vp(counts.data_ptr()), 2048, vp(s1.cuda_stream)) == 0
            torch.cuda.synchronize()
            c = counts.reshape(4, 16).sum(1).tolist()
            print(f"E6 [{fname}] next to [{partner}]: mismatching chain pairs by lane quarter [0-15,16-31,32-47,48-63] = {c}", flush=True)


def encoder_probe(rounds=12):
    """E7: the encoder's kernels (split-fp16 convolutions, InstanceNorm, q|k|v, window attention, K7) as possible victims:
    feature maps computed while the f16x3 decoder runs on another stream against the maps computed alone, bit for bit."""
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    rays, m = rays_of(0)
    cond_ref = staged[0][0]
    with torch.no_grad():
        solo = [f.clone() for f in model.get_img_feat(batch.images[:, :3], cur_n_src_views=3)]
        again = model.get_img_feat(batch.images[:, :3], cur_n_src_views=3)
        torch.cuda.synchronize()
        print("E7 encoder alone, twice: bit-identical =", all(torch.equal(a, b) for a, b in zip(solo, again)), flush=True)
        diffs = [0, 0]
        for it in range(rounds):
            torch.cuda.synchronize()
            with torch.cuda.stream(s2):
                for _ in range(3):
                    hip.decoder_chunk(dec, sc.views[0], rays, cond_ref.reshape(-1), stream=s2)
            with torch.cuda.stream(s1):
                f = model.get_img_feat(batch.images[:, :3], cur_n_src_views=3)
            torch.cuda.synchronize()
            for k in range(2):
                diffs[k] += int((f[k].view(torch.int32) != solo[k].view(torch.int32)).sum())
    print(f"E7 encoder next to the f16x3 decoder, {rounds} rounds: differing feature values per scale {diffs} "
          f"(of {solo[0].numel()} / {solo[1].numel()} per round)", flush=True)


if os.environ.get("PROBE_E7"):
    encoder_probe()
if os.environ.get("PROBE_E6"):
    pk_probe()
if os.environ.get("PROBE_E4"):
    mask_probe()
if os.environ.get("PROBE_E3"):
    sentinel_probe()
if os.environ.get("PROBE_E2"):
    lockstep_probe()
if os.environ.get("PROBE_E1", "1") == "1":
    concurrent_streams()
if os.environ.get("PROBE_E1_ONLY"):
    sys.exit(0)

run_config("one WG/CU (84 KiB), baseline", 84, 0, frames=3)
run_config("two WG/CU", 0, 0)
run_config("two WG/CU, weight segments by plain loads + LDS stores (no LDS-DMA anywhere)", 0, 64, verbose=3)
run_config("two WG/CU, no LDS-DMA + seg0 after the walk", 0, 64 | 4, verbose=3)
run_config("two WG/CU + 2|4|8", 0, 14, verbose=2)
