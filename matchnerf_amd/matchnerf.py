"""MatchNeRF — drop-in module for the reference's ``models.matchnerf.MatchNeRF``.

Same constructor (``MatchNeRF(opts)``), same attributes (``feat_enc``, ``nerf_dec``,
``nerf_setbg_opaque``, ``n_src_views``), same ``forward(batch, mode, render_video,
render_path_mode)`` contract and the same ``state_dict`` keys as
/root/reference/models/matchnerf.py:13-325 — but the per-ray hot path runs in hand-written
HIP kernels for MI355X (``libmnerf_hip.so``, include/mnerf.h):

    encoder        GMFlow with the K6 shifted-window attention kernel        (gmflow.py)
    render chunk   K1+K2 cost volume -> K3+K4+K5 fused decoder/compositing   (mnerf_render_chunk)

What the reference does per 4096-ray chunk in ~100 eager ops (full-image ray grid, six
268 MB sampled-feature tensors, [R,4,S,S] attention scores ...) is two kernel launches here,
fed by pair-major channel-last feature maps that stay resident in HBM for the whole frame.

The HIP path is the ONLY path: there is no eager fallback.  Without the shared library or
without a GPU, rendering raises ``hip.MnerfError`` / ``RuntimeError``.
"""
import os

import numpy as np
import torch

from . import camera, hip
from .cond_nerf import CondNeRF
from .edict import EasyDict as edict
from .gmflow import GMFlow, pair_major_to_view_chunks

# rays per kernel launch when a full image is rendered.  The reference's
# ``nerf.rand_rays_{val,test}`` only bounds its temporaries (README.md:132); results are
# chunk-invariant (tests/test_hip_kernels.py), so larger launches are used here.
MAX_RAYS_PER_LAUNCH = int(os.environ.get("MNERF_MAX_RAYS_PER_LAUNCH", "65536"))
# a pose-table launch (render_poses) may carry more: frames of a video are rendered back to back anyway, and with 288 GB of HBM
# the conditioning hand-off of a quarter of a million rays (3 GB at 128 samples per ray) is no constraint.  What it buys is fewer,
# fuller launches: configs/demo_own.yaml's 24 frames of 256 x 160 at S = 128 go out as 4 launches of 6 poses instead of 24 x 2.
MAX_RAYS_PER_POSE_LAUNCH = int(os.environ.get("MNERF_MAX_RAYS_PER_POSE_LAUNCH", "262144"))
# rays per autograd node when gradients are required (bounds the temporaries of the re-evaluated backward)
GRAD_RAYS_PER_CALL = 8192


class MatchNeRF(torch.nn.Module):
    def __init__(self, opts):
        super().__init__()
        self.opts = opts
        self.nerf_setbg_opaque = False
        self.n_src_views = opts.n_src_views
        if self.n_src_views > hip.MNERF_MAX_VIEWS:
            raise NotImplementedError(f"n_src_views={self.n_src_views} > {hip.MNERF_MAX_VIEWS}")
        self.feat_enc = GMFlow(feature_channels=128, num_scales=1, num_head=1, attention_type="swin",
                               ffn_dim_expansion=4, feature_upsampler=opts.encoder.feature_upsampler,
                               upsample_factor=opts.encoder.upsample_factor,
                               num_transformer_layers=opts.encoder.num_transformer_layers,
                               device=opts.device).to(opts.device)
        self.nerf_dec = CondNeRF(opts).to(opts.device)
        if getattr(opts.encoder, "feature_sample_local_radius", 0):
            raise NotImplementedError("encoder.feature_sample_local_radius > 0 (gmflow/utils.py:136-162) is not "
                                      "built; every shipped config uses 0 (base.yaml:27)")
        self._ws = None
        self._enc_graphs = {}     # captured encoder passes (get_img_feat), keyed by input shape and weight versions
        self.encoder_graph = os.environ.get("MNERF_ENCODER_GRAPH", "0") == "1"  # opt-in: measured no gain (see _encoder_graph_replay)
        self._frame = None        # per-source-set launch context (host camera copies, RGBA images): see _frame_ctx
        self.kernel_timer = None  # hip.KernelTimer: per-kernel event timing (bench.py)
        self.fused_render = False  # True: ray chunks take the one-launch form where it exists (slower on MI355X: DESIGN.md)
        # video of small frames: several poses per launch through a pose table (mnerf_rays.pose_table).  OFF by default since
        # round 6: the table never paid (157.4 against 158.6 ms for demo_own.yaml's 24 frames, profiles/history/r5_video_demo_own.log)
        # and it keeps the cost volume on the segment walk, which the matrix form of the pose-by-pose launches now beats
        # (DESIGN.md section 4).  MNERF_POSE_BATCHING=1 brings it back; tests/test_pose_table_gpu.py keeps it bit-identical.
        self.pose_batching = os.environ.get("MNERF_POSE_BATCHING", "0") == "1"
        self.cv_matrix_form = os.environ.get("MNERF_CV_MM", "1") != "0"  # full-frame renders: the cost volume on the matrix pipe
        self._cv_ops = None

    def _dec(self):
        """the CondNeRF module, also when the reference's coach wrapped it in nn.DataParallel (coach.py:83-85)"""
        return getattr(self.nerf_dec, "module", self.nerf_dec)

    # ------------------------------------------------------------------ forward (matchnerf.py:32-73)
    def forward(self, batch, mode=None, render_video=False, render_path_mode="interpolate"):
        self._frame = None  # the launch context below never outlives one forward (see _frame_ctx)
        self._cv_ops = None
        ref_images = batch.images[:, :self.n_src_views]
        ref_feats_list = self.get_img_feat(ref_images, attn_splits_list=self.opts.encoder.attn_splits_list,
                                           cur_n_src_views=self.n_src_views)
        tgt_pose, ref_poses = self.extract_poses(batch)
        batch_size, _, _, img_h, img_w = ref_images.shape

        if render_video:
            assert mode in ["test", "val"], f"Do NOT render video in mode {mode}, change to either 'test' or 'val'."
            poses_paths = self.get_video_rendering_path(tgt_pose, ref_poses, render_path_mode,
                                                        self.opts.nerf.video_n_frames, batch)
        else:
            poses_paths = [tgt_pose]

        mode_rand_rays = getattr(self.opts.nerf, f"rand_rays_{mode}", 0)
        collected, frames_done = {}, {}
        if render_video and self.pose_batching:
            # small frames: as many poses per launch as fill one (mnerf_rays.pose_table); None where the kernels or the
            # frame size do not take a table -> the pose loop below
            frames = self.render_poses(self.opts, poses_paths, ref_poses=ref_poses, ref_images=ref_images,
                                       ref_feats_list=ref_feats_list)
            if frames is not None:
                for k, v in frames.items():  # [n_poses, B, N, C] -> the reference's frame-major [n_poses * B, N, C]
                    host = torch.empty((v.shape[0] * v.shape[1],) + tuple(v.shape[2:]), dtype=v.dtype, pin_memory=True)
                    host.copy_(v.reshape(host.shape), non_blocking=True)
                    collected[k] = host
                    frames_done[k] = v
                poses_paths = []
        for cur_tgt_pose in poses_paths:
            if mode_rand_rays and mode in ["train", "test-optim"]:
                batch.ray_idx = torch.randperm(img_h * img_w, device=ref_images.device)[:mode_rand_rays // batch_size]
                ret = self.render(self.opts, cur_tgt_pose, ray_idx=batch.ray_idx, mode=mode, ref_poses=ref_poses,
                                  ref_images=ref_images, ref_feats_list=ref_feats_list)
            elif mode_rand_rays:
                ret = self.render_by_slices(self.opts, cur_tgt_pose, mode=mode, ref_poses=ref_poses,
                                            ref_images=ref_images, ref_feats_list=ref_feats_list)
            else:
                ret = self.render(self.opts, cur_tgt_pose, mode=mode, ref_poses=ref_poses, ref_images=ref_images,
                                  ref_feats_list=ref_feats_list)
            if render_video:
                # Frames go to the host as the reference's do (matchnerf.py:62-70, per-frame .cpu()), but without a
                # host sync per frame: one pinned buffer per output holds all frames, each frame is an async copy
                # queued behind its own kernels, and the GPU starts the next pose while Python is still here.
                for k, v in ret.items():
                    v = v.detach()
                    if k not in collected:
                        collected[k] = torch.empty((len(poses_paths) * v.shape[0],) + tuple(v.shape[1:]), dtype=v.dtype,
                                                   pin_memory=True)
                    lo = len(frames_done.setdefault(k, [])) * v.shape[0]
                    collected[k][lo:lo + v.shape[0]].copy_(v, non_blocking=True)
                    frames_done[k].append(v)  # keep the device tensor alive until the copy has run
            else:
                for k, v in ret.items():
                    collected.setdefault(k, []).append(v)
        if render_video:
            torch.cuda.current_stream(ref_images.device).synchronize()
            for k, v in collected.items():
                batch[k] = v
        else:
            for k, v in collected.items():
                batch[k] = torch.cat(v, dim=0)
        return batch

    def extract_poses(self, batch):
        """matchnerf.py:75-86: target = LAST view, sources = all the others."""
        tgt_pose = dict(extrinsics=batch.extrinsics[:, -1, :3, :], intrinsics=batch.intrinsics[:, -1],
                        near_fars=batch.near_fars[:, -1])
        ref_poses = dict(extrinsics=batch.extrinsics[:, :-1, :3, :], intrinsics=batch.intrinsics[:, :-1],
                         near_fars=batch.near_fars[:, :-1])
        return tgt_pose, ref_poses

    # ------------------------------------------------------------------ encoder (matchnerf.py:183-207)
    def get_img_feat(self, imgs, attn_splits_list=None, cur_n_src_views=3):
        """-> list over scales of pair-major channel-last maps [B,P,2,h,w,128]
        (``gmflow.pair_major_to_view_chunks`` converts to the reference's [B,V,(V-1)*128,h,w])."""
        if attn_splits_list is None:
            attn_splits_list = self.opts.encoder.attn_splits_list
        imgs = imgs[:, :cur_n_src_views]
        if self.encoder_graph and imgs.is_cuda and not torch.is_grad_enabled() and not torch.cuda.is_current_stream_capturing():
            return self._encoder_graph_replay(imgs, attn_splits_list)
        return self.feat_enc(imgs=imgs, attn_splits_list=attn_splits_list, wo_self_attn=self.opts.encoder.wo_self_attn)

    def _encoder_graph_replay(self, imgs, attn_splits_list):
        """The inference encoder as ONE HIP graph launch (opt-in: MNERF_ENCODER_GRAPH=1 or .encoder_graph = True).  An encoder
        pass is ~100 short kernels (15 convolutions, 15 norms, 12 x (q|k|v, window attention, K7), layout glue); the idea was
        that launch gaps are a sizeable part of its 3.9 ms.  MEASURED (round 4, same box): frame 30.27-30.50 ms with the graph,
        30.32 without, encoder 3.85 vs 3.89 ms — the host enqueues far enough ahead that the kernels already run back to back;
        what is left of the 3.9 ms is the kernels themselves.  Kept because it is correct and tested, not because it pays.
        Every kernel of the pass is enqueue-only on the current stream (include/mnerf.h), so the
        pass is captured once per (input shape, attention splits, encoder weights) and replayed: the input is copied into the
        graph's static buffer, the outputs are CLONED out of it (78 MB at 3 views: ~40 us), so results never alias a later call.
        A weight update (load_state_dict, an optimizer step) changes the parameters' version counters and thereby the key.
        MNERF_ENCODER_GRAPH=0 (or .encoder_graph = False) keeps the eager pass; a failed capture falls back to it once and
        for all with a warning (the kernels are the same either way)."""
        enc = self.feat_enc
        splits = tuple(attn_splits_list) if isinstance(attn_splits_list, (list, tuple)) else (attn_splits_list,)
        wkey = tuple((int(p._version), int(p.data_ptr())) for p in enc.parameters())
        # what the captured kernels depend on besides the input: the attention arithmetic (MNERF_WA_MATH picks other kernels) and
        # the encoder's tuning knobs; a graph of other WEIGHTS is dead (its key can never match again): dropped, not kept
        shape_key = (tuple(imgs.shape), str(imgs.device), splits, bool(self.opts.encoder.wo_self_attn), hip.wa_math(),
                     os.environ.get("MNERF_WA_MIN4"), os.environ.get("MNERF_WA_XCD"))
        key = shape_key + (hash(wkey),)
        entry = self._enc_graphs.get(key)
        if entry is None:
            for stale in [k for k in self._enc_graphs if k[:-1] == shape_key]:
                del self._enc_graphs[stale]
            run = lambda x: enc(imgs=x, attn_splits_list=attn_splits_list, wo_self_attn=self.opts.encoder.wo_self_attn)
            static_in = imgs.clone()
            run(static_in)  # eager warm-up: weight packing, LDS attributes, cached index / position tensors
            torch.cuda.synchronize(imgs.device)
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    static_out = run(static_in)
            except Exception as e:  # noqa: BLE001
                import warnings
                warnings.warn(f"encoder graph capture failed ({type(e).__name__}: {e}); using the eager encoder pass")
                self.encoder_graph = False
                return run(imgs)
            if len(self._enc_graphs) >= 4:  # a handful of live shapes at most (each graph owns its activations)
                self._enc_graphs.pop(next(iter(self._enc_graphs)))
            entry = self._enc_graphs[key] = (graph, static_in, static_out)
        graph, static_in, static_out = entry
        static_in.copy_(imgs)
        graph.replay()
        return [o.clone() for o in static_out]

    # ------------------------------------------------------------------ C-ABI argument structs
    def _scene(self, b, ref_poses_host, ref_feats_list, images_cl, feats_b=None):
        """``feats_b``: per-scale maps [P,2,h,w,128] of batch element b (default: slices of
        ``ref_feats_list``)."""
        if feats_b is None:
            feats_b = [f[b] for f in ref_feats_list]
        sc = hip.Scene()
        sc.n_views, sc.n_scales = self.n_src_views, len(feats_b)
        groups = self.opts.encoder.cos_n_group
        groups = [groups] if isinstance(groups, int) else list(groups)
        assert len(groups) == len(feats_b), "cos_n_group needs one entry per feature scale"
        for s, f in enumerate(feats_b):
            assert f.is_contiguous()
            sc.fh[s], sc.fw[s] = f.shape[2], f.shape[3]
            sc.n_group[s] = groups[s]
            sc.feat[s] = f.data_ptr()
        sc.images = images_cl[b].data_ptr()
        ex, it, nf = ref_poses_host
        for v in range(self.n_src_views):
            sc.views[v] = hip.make_view(ex[b, v], it[b, v], nf[b, v, 0], nf[b, v, 1])
        return sc

    def _scene_mm(self, b, ref_poses_host, ref_feats_list, images_cl):
        """``_scene`` + the split-fp16 operand image of the feature maps (``hip.cost_volume_operands``, ABI v9) that the matrix
        form of the cost volume reads: built once per source set and batch element (two short launches, ~the bytes of the maps),
        shared by every pose and ray chunk rendered from it.  Keyed on the identity and version of the maps; the entry holds the
        maps themselves (see ``_frame_ctx`` for why) and ``forward`` drops it at its top.  MNERF_CV_MM=0 keeps the walk."""
        sc = self._scene(b, ref_poses_host, ref_feats_list, images_cl)
        if not self.cv_matrix_form:
            return sc
        key = tuple((f.data_ptr(), f._version, tuple(f.shape)) for f in ref_feats_list)
        if self._cv_ops is None or self._cv_ops[0] != key:
            self._cv_ops = (key, {}, list(ref_feats_list))
        ops = self._cv_ops[1]
        if b not in ops:
            ops[b] = hip.cost_volume_operands(sc, device=ref_feats_list[0].device)
        else:
            sc.feat_op = ops[b].data_ptr()
        return sc

    def _decoder(self, n_samples, device):
        return self._dec().decoder_struct(n_samples, device, self.nerf_setbg_opaque)

    def _workspace(self, n_floats, device):
        if self._ws is None or self._ws.numel() < n_floats or self._ws.device != device:
            self._ws = torch.empty(n_floats, device=device)
        return self._ws

    @staticmethod
    def _host(t):
        return t.detach().float().cpu().numpy()

    def _frame_ctx(self, ref_poses, ref_images):
        """Launch context of one source set, built once and shared by every target pose rendered from it (all
        frames of a video, all chunks of a frame): host copies of the source cameras (they travel by value in the
        kernel arguments) and the channel-last RGBA copy of the source images.  Keyed on the identity and version
        of the tensors, so in-place edits or a new batch rebuild it; the entry holds the keyed tensors themselves, so
        their storage cannot be freed and handed to a NEW batch at the same address while the entry lives (the caching
        allocator does exactly that in the reference's coach loop), and ``forward`` drops the entry at its top: a
        context is shared by the poses and chunks of ONE forward, never by two batches."""
        keyed = (ref_images, ref_poses["extrinsics"], ref_poses["intrinsics"], ref_poses["near_fars"])
        key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in keyed)
        if self._frame is None or self._frame[0] != key:
            b, v, _, h, w = ref_images.shape
            ref_host = (self._host(ref_poses["extrinsics"]), self._host(ref_poses["intrinsics"]),
                        self._host(ref_poses["near_fars"]))
            images_cl = torch.zeros(b, v, h, w, 4, device=ref_images.device)
            images_cl[..., :3] = ref_images.detach().permute(0, 1, 3, 4, 2)
            self._frame = (key, ref_host, images_cl, keyed)
        return self._frame[1], self._frame[2]

    def _tgt_host(self, tgt_pose):
        """(extrinsics, intrinsics, near_fars) of the target pose on the host; poses generated on the host
        (video paths) carry their numpy originals under "_host" and cost no device sync."""
        if "_host" in tgt_pose:
            return tgt_pose["_host"]
        return tuple(self._host(tgt_pose[k]) for k in ("extrinsics", "intrinsics", "near_fars"))

    # ------------------------------------------------------------------ render (matchnerf.py:88-143)
    def render(self, opt, tgt_pose=None, ray_idx=None, mode=None, ref_poses=None, ref_images=None,
               ref_feats_list=None, ray_range=None):
        """Rays of one target pose -> edict(rgb [B,N,3], depth [B,N,1], opacity [B,N,1]).
        ``ray_idx`` (LongTensor [N], shared by the batch) selects pixels; ``ray_range`` = (first pixel, count) a contiguous run of
        pixels (a band of rows: dist.render_frame_sharded) that keeps the kernels of the full-frame path; neither = full image."""
        if tgt_pose is None:
            raise Exception("Must provide tgt_pose.")
        if not ref_images.is_cuda:
            raise RuntimeError("MatchNeRF.render: the HIP render path needs CUDA tensors (no CPU fallback)")
        batch_size, _, _, img_h, img_w = ref_images.shape
        device = ref_images.device
        n_samples = int(opt.nerf.sample_intvs)
        legacy = bool(opt.nerf.legacy_coord)
        assert ray_idx is None or ray_range is None
        pix0, n_rays = (0, img_h * img_w) if ray_range is None else (int(ray_range[0]), int(ray_range[1]))
        if ray_idx is not None:
            n_rays = int(ray_idx.numel())
        idx32 = None if ray_idx is None else ray_idx.to(device=device, dtype=torch.int32).contiguous()
        stratified = mode == "train" and bool(opt.nerf.sample_stratified)

        dec_mod = self._dec()
        needs_grad = torch.is_grad_enabled() and (any(f.requires_grad for f in ref_feats_list) or
                                                  any(p.requires_grad for p in dec_mod.parameters()))
        ref_host, images_cl = self._frame_ctx(ref_poses, ref_images)
        tgt_ex, tgt_in, tgt_nf = self._tgt_host(tgt_pose)
        if needs_grad:
            if ray_range is not None:
                ray_idx = torch.arange(pix0, pix0 + n_rays, device=device)
            return self._render_with_grad(opt, ref_host, (tgt_ex, tgt_in, tgt_nf), ray_idx, stratified, ref_images,
                                          ref_feats_list, images_cl, n_rays, n_samples, img_h, img_w)
        dec = self._decoder(n_samples, device)
        chunk = min(n_rays, MAX_RAYS_PER_LAUNCH)
        ws = None  # [rays*S, cond_stride] hand-off buffer of the staged form
        rgb = torch.empty(batch_size, n_rays, 3, device=device)
        depth = torch.empty(batch_size, n_rays, 1, device=device)
        opacity = torch.empty(batch_size, n_rays, 1, device=device)
        for b in range(batch_size):
            sc = self._scene(b, ref_host, ref_feats_list, images_cl) if idx32 is not None else \
                self._scene_mm(b, ref_host, ref_feats_list, images_cl)
            kinv, c2w = camera.target_ray_consts(tgt_ex[b], tgt_in[b], legacy)
            strat = torch.rand(n_rays, n_samples, device=device) if stratified else None
            for c in range(0, n_rays, chunk):
                m = min(chunk, n_rays - c)
                rays = hip.make_rays(
                    m, n_samples, img_h, img_w, kinv, c2w, tgt_nf[b, 0], tgt_nf[b, 1], ray_begin=pix0 + c, legacy=legacy,
                    depth_inverse=(opt.nerf.depth.param == "inverse"),
                    ray_idx_ptr=None if idx32 is None else idx32[c:].data_ptr(),
                    strat_u_ptr=None if strat is None else strat[c:].data_ptr())
                fused = self.fused_render and hip.render_is_fused(sc, dec, rays)
                if ws is None and not fused:
                    ws = self._workspace(hip.render_workspace_bytes(chunk, n_samples, dec.cond_stride) // 4, device)
                hip.render_chunk(sc, dec, rays, ws, rgb[b, c:c + m], depth[b, c:c + m], opacity[b, c:c + m],
                                 timer=self.kernel_timer, fused=fused)
        return edict(rgb=rgb, depth=depth, opacity=opacity)

    def render_poses(self, opt, poses, ref_poses=None, ref_images=None, ref_feats_list=None):
        """Full frames of SEVERAL target poses of one source set (the video loop of matchnerf.py:42-71) with as many poses per
        launch as fit MAX_RAYS_PER_POSE_LAUNCH rays: the poses' camera constants travel as a table in HBM (mnerf_rays.pose_table,
        include/mnerf.h) instead of by value in the kernel arguments.  A 128 x 160 frame is 20 480 rays = 80 decoder tiles, a
        third of the 256 workgroups the decoder keeps resident; three poses per launch fill them.  Results are bit-identical
        to ``render`` pose by pose (tests/test_model_gpu.py).
        -> edict(rgb [n_poses,B,N,3], depth [n_poses,B,N,1], opacity [n_poses,B,N,1]) on the device, or None when a table does
        not apply (a single frame exceeds half a pose launch, H*W not a multiple of 64, more than 5 source views, sample_intvs >
        128, a non-default decoder form: ``hip.render_takes_pose_table``)."""
        batch_size, _, _, img_h, img_w = ref_images.shape
        device = ref_images.device
        n_pix = img_h * img_w
        per_launch = MAX_RAYS_PER_POSE_LAUNCH // n_pix
        if per_launch < 2 or len(poses) < 2 or not ref_images.is_cuda or self.fused_render:
            return None
        n_samples = int(opt.nerf.sample_intvs)
        # the hand-off rows of a launch are rays x S x cond_stride floats (3.2 GB for 262 144 rays at S = 128): never more than
        # a quarter of what the device has free, so that a smaller card renders with fewer poses per launch instead of failing
        if ref_images.is_cuda:
            free, _ = torch.cuda.mem_get_info(device)
            row_bytes = n_pix * n_samples * 4 * hip.MNERF_COND_STRIDE_MAX
            per_launch = max(1, min(per_launch, int(free // 4 // max(row_bytes, 1))))
            if per_launch < 2:
                return None
        legacy = bool(opt.nerf.legacy_coord)
        ref_host, images_cl = self._frame_ctx(ref_poses, ref_images)
        dec = self._decoder(n_samples, device)
        scenes = [self._scene(b, ref_host, ref_feats_list, images_cl) for b in range(batch_size)]
        if not all(hip.render_takes_pose_table(sc, dec, n_samples, n_pix) for sc in scenes):
            return None
        n_poses = len(poses)
        hosts = [self._tgt_host(p) for p in poses]
        rows = np.zeros((batch_size, n_poses, hip.MNERF_POSE_FLOATS), np.float32)
        for b in range(batch_size):
            rows[b] = hip.pose_table_rows([camera.target_ray_consts(ex[b], it[b], legacy) + (nf[b, 0], nf[b, 1])
                                           for ex, it, nf in hosts])
        table = torch.from_numpy(rows).to(device)  # one small copy per video; stream-ordered before the launches
        rgb = torch.empty(n_poses, batch_size, n_pix, 3, device=device)
        depth = torch.empty(n_poses, batch_size, n_pix, 1, device=device)
        opacity = torch.empty(n_poses, batch_size, n_pix, 1, device=device)
        # outputs of a launch are [poses of the group] x [pixels]: a contiguous scratch block per launch, then one strided copy
        # into the frame-major result (the batch dimension sits between pose and pixel there)
        ws = self._workspace(hip.render_workspace_bytes(per_launch * n_pix, n_samples, dec.cond_stride) // 4, device)
        contiguous_out = batch_size == 1
        if not contiguous_out:
            tmp = [torch.empty(per_launch * n_pix, c, device=device) for c in (3, 1, 1)]
        for b in range(batch_size):
            kinv, c2w = rows[b, 0, :9], rows[b, 0, 9:21]
            for p0 in range(0, n_poses, per_launch):
                g = min(per_launch, n_poses - p0)
                rays = hip.make_rays(g * n_pix, n_samples, img_h, img_w, kinv, c2w, rows[b, 0, 21], rows[b, 0, 22],
                                     ray_begin=p0 * n_pix, legacy=legacy, depth_inverse=(opt.nerf.depth.param == "inverse"),
                                     pose_table_ptr=table[b].data_ptr(), rays_per_pose=n_pix)
                if contiguous_out:
                    outs = (rgb[p0:p0 + g, 0].reshape(-1, 3), depth[p0:p0 + g, 0].reshape(-1, 1),
                            opacity[p0:p0 + g, 0].reshape(-1, 1))
                else:
                    outs = tuple(t[:g * n_pix] for t in tmp)
                hip.render_chunk(scenes[b], dec, rays, ws, *outs, timer=self.kernel_timer, fused=False)
                if not contiguous_out:
                    for dst, src in zip((rgb, depth, opacity), outs):
                        dst[p0:p0 + g, b] = src.view(g, n_pix, -1)
        return edict(rgb=rgb, depth=depth, opacity=opacity)

    def _render_with_grad(self, opt, ref_host, tgt_host, ray_idx, stratified, ref_images, ref_feats_list, images_cl,
                          n_rays, n_samples, img_h, img_w):
        """Training path (matchnerf_amd/autograd.py): forward through the HIP kernels; backward = HIP kernels end to end for the
        ray chunk (mnerf_composite_backward -> mnerf_decoder_backward -> mnerf_cost_volume_backward).  Rays go through in
        chunks of GRAD_RAYS_PER_CALL so that a full-image call under autograd keeps the decoder backward's workspace (11.2 KB
        per sample) bounded."""
        from . import autograd as ag
        device = ref_images.device
        legacy = bool(opt.nerf.legacy_coord)
        tgt_ex, tgt_in, tgt_nf = tgt_host
        batch_size = ref_images.shape[0]
        idx_all = torch.arange(n_rays, device=device) if ray_idx is None else ray_idx.to(device)
        dec_mod = self._dec()
        outs = []
        for b in range(batch_size):
            strat_all = torch.rand(n_rays, n_samples, device=device) if stratified else None
            kinv, c2w = camera.target_ray_consts(tgt_ex[b], tgt_in[b], legacy)
            parts = []
            for c in range(0, n_rays, GRAD_RAYS_PER_CALL):
                idx = idx_all[c:c + GRAD_RAYS_PER_CALL]
                idx32 = idx.to(torch.int32).contiguous()
                strat = None if strat_all is None else strat_all[c:c + GRAD_RAYS_PER_CALL].contiguous()
                m = int(idx.numel())

                def make_rays(strat=strat, kinv=kinv, c2w=c2w, idx32=idx32, m=m, b=b):
                    r = hip.make_rays(m, n_samples, img_h, img_w, kinv, c2w, tgt_nf[b, 0], tgt_nf[b, 1], legacy=legacy,
                                      depth_inverse=(opt.nerf.depth.param == "inverse"), ray_idx_ptr=idx32.data_ptr(),
                                      strat_u_ptr=None if strat is None else strat.data_ptr())
                    return r, (idx32, strat)  # the struct holds raw pointers: keep the tensors alive with it

                launch = ag.RayChunkLaunch(
                    opt, dec_mod, make_scene=lambda feats, b=b: self._scene(b, ref_host, None, images_cl, feats_b=feats),
                    make_rays=make_rays, make_decoder=lambda: self._decoder(n_samples, device),
                    view0_extr=ref_host[0][b, 0], kinv=kinv, c2w=c2w, ray_idx=idx, width=img_w, n_views=self.n_src_views,
                    setbg_opaque=bool(self.nerf_setbg_opaque))
                parts.append(ag.render_ray_chunk(launch, [f[b] for f in ref_feats_list]))
            outs.append([torch.cat([p[k] for p in parts], 0) for k in range(3)])
        return edict(rgb=torch.stack([o[0] for o in outs], 0), depth=torch.stack([o[1] for o in outs], 0),
                     opacity=torch.stack([o[2] for o in outs], 0))

    def render_by_slices(self, opt, tgt_pose, mode=None, ref_poses=None, ref_images=None, ref_feats_list=None):
        """matchnerf.py:145-161.  The reference loops over ``rand_rays_<mode>``-sized slices to
        bound memory; the fused kernels have no such temporaries, so the full image is rendered
        in launches of up to MAX_RAYS_PER_LAUNCH rays (identical results, see module docstring)."""
        assert ref_images is not None, "Must provide the reference images for MatchNeRF."
        return self.render(opt, tgt_pose, ray_idx=None, mode=mode, ref_poses=ref_poses, ref_images=ref_images,
                           ref_feats_list=ref_feats_list)

    # ------------------------------------------------------------------ API-compat helpers
    def sample_depth(self, opt, batch_size, num_rays, near_far, legacy=False, mode="train"):
        """matchnerf.py:163-181 (torch; the kernels evaluate the same expression in-register)."""
        s = opt.nerf.sample_intvs
        dev = near_far.device
        depth_min, depth_max = near_far[:, :1], near_far[:, 1:]
        if mode == "train" and opt.nerf.sample_stratified:
            t = torch.rand(batch_size, num_rays, s, 1, device=dev)
        else:
            t = (0.0 if legacy else 0.5) * torch.ones(batch_size, num_rays, s, 1, device=dev)
        t = t + torch.arange(s, device=dev)[None, None, :, None].float()
        d = t / ((s - 1) if legacy else s) * (depth_max - depth_min).reshape(batch_size, 1, 1, 1) \
            + depth_min.reshape(batch_size, 1, 1, 1)
        return dict(metric=d, inverse=1 / (d + 1e-8))[opt.nerf.depth.param]

    def query_cond_info(self, point_samples, ref_poses, ref_images, ref_feats_list, tgt_pose=None, ray_idx=None):
        """matchnerf.py:209-293 through the K1+K2 kernel.  The kernel rebuilds the sample points
        from the target pose, so ``tgt_pose`` (+ optional ``ray_idx``) replaces ``point_samples``
        (kept in the signature for call compatibility; only its shape is used)."""
        if tgt_pose is None:
            raise NotImplementedError("query_cond_info needs tgt_pose: the HIP kernel regenerates the 3D samples")
        b_n, n_rays, n_samples = point_samples.shape[:3]
        _, v, _, img_h, img_w = ref_images.shape
        device = ref_images.device
        ref_host, images_cl = self._frame_ctx(ref_poses, ref_images)
        dc = self._dec().cond_dim
        stride = ((dc + 1 + 7) // 8) * 8
        sum_g = dc - 4 * v
        out = torch.empty(b_n, n_rays * n_samples, stride, device=device)
        idx32 = None if ray_idx is None else ray_idx.to(device=device, dtype=torch.int32).contiguous()
        for b in range(b_n):
            sc = self._scene(b, ref_host, ref_feats_list, images_cl)
            t_ex, t_in, t_nf = self._tgt_host(tgt_pose)
            kinv, c2w = camera.target_ray_consts(t_ex[b], t_in[b], bool(self.opts.nerf.legacy_coord))
            nf = t_nf[b]
            rays = hip.make_rays(n_rays, n_samples, img_h, img_w, kinv, c2w, nf[0], nf[1],
                                 legacy=bool(self.opts.nerf.legacy_coord),
                                 depth_inverse=(self.opts.nerf.depth.param == "inverse"),
                                 ray_idx_ptr=None if idx32 is None else idx32.data_ptr())
            hip.cost_volume(sc, rays, stride, out=out[b])
        out = out.reshape(b_n, n_rays, n_samples, stride)
        return dict(feat_info=out[..., :sum_g], color_info=out[..., sum_g:sum_g + 3 * v],
                    mask_info=out[..., sum_g + 3 * v:dc])

    def get_img_feat_view_chunks(self, imgs):
        """``get_img_feat`` in the reference's return format [B,V,(V-1)*128,h,w] per scale."""
        return [pair_major_to_view_chunks(f) for f in self.get_img_feat(imgs, cur_n_src_views=self.n_src_views)]

    # ------------------------------------------------------------------ video paths (matchnerf.py:295-325)
    def get_video_rendering_path(self, tgt_pose, ref_poses, mode, n_frames=30, batch=None):
        from . import video
        device = tgt_pose["extrinsics"].device
        per_batch = []
        for bi, src in enumerate(ref_poses["extrinsics"]):
            if mode == "interpolate":
                sq = torch.eye(4).repeat(src.shape[0], 1, 1)
                sq[:, :3] = src.detach().cpu()
                c2ws = sq.double().inverse()[:, :3].float().numpy()
                path = video.interpolate_render_path(c2ws, n_frames)
            elif mode == "spiral":
                assert batch is not None, "Must provide all c2ws and near_far for getting spiral rendering path."
                c2ws_all = batch["c2ws_all"][bi].detach().cpu().numpy()
                nf = tgt_pose["near_fars"][bi].detach().cpu().numpy().tolist()
                path = video.spiral_render_path(c2ws_all, nf, rads_scale=getattr(self.opts.nerf, "video_rads_scale", 0.1),
                                                n_views=n_frames)
            else:
                raise Exception(f"Unknown video rendering path mode {mode}")
            per_batch.append(torch.tensor(np.asarray(path)).inverse()[:, :3].to(torch.float32))
        paths_host = torch.stack(per_batch, 0)                       # [B, n_frames, 3, 4] on the host
        paths = paths_host.to(device)
        _, t_in, t_nf = self._tgt_host(tgt_pose)                     # one device->host copy for the whole path
        paths_np = paths_host.numpy()
        # the reference indexes frames 0..n_frames-1 of the path (matchnerf.py:319-323); "_host" carries the numpy
        # originals so that render() needs no device->host copy per frame
        return [dict(extrinsics=paths[:, i], intrinsics=tgt_pose["intrinsics"].clone().detach(),
                     near_fars=tgt_pose["near_fars"].clone().detach(), _host=(paths_np[:, i], t_in, t_nf))
                for i in range(n_frames)]
