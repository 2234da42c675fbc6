"""Device-side packing of the GMFlow transformer's weights into the split-fp16 MFMA operand streams — the training path's twin
of the numpy packers in gmflow.py (`pack_qkv`, `pack_encoder_block`), bit-identical to them (tests/test_packing_cpu.py).

Two forms: per stream (`pack_qkv`, `pack_encoder_block`: a handful of device ops per fragment) and, what training uses, the
whole transformer in one go (`TransformerPacker`: ONE concatenation of all weights, ONE gather through an index built once, the
fp16 split — about ten launches and one device->host copy per optimizer step for all 24 streams of the 12 layers).

Why: the training forward of every transformer layer runs the inference kernels, which read host-packed weight streams cached
on the parameters' `_version`.  After each `optimizer.step()` all 12 layers re-packed on the HOST: `.detach().cpu().numpy()` on
every weight (a blocking device sync each), numpy fragment building for the [1024,256] / [128,1024] FFN weights, a pageable
host->device copy — dozens of syncs per training step that the torch-Linear path did not have (advisor, round 4).  Here a
stream is ONE gather of the weight (index tensor built once per layer shape and kept on the device), a power-of-two scale and
the fp16 hi | lo split, all as device ops on the caller's stream; the only host round trip is the per-tensor scale exponent the
kernels take as an integer argument, and `FeatureTransformer.refresh_packs` fetches the exponents of ALL stale layers with
one device->host copy per step.

A fragment stream [T, nmb, 2, 64, 8] fp16 (cond_nerf._fragments_h): element (t, m, part, lane, j) = hi | lo part of
2^ew * W[32 m + lane % 32, cols[t, lane // 32, j]] (zero outside W, or where cols < 0)."""
import numpy as np
import torch

from . import cond_nerf as CN

_INDEX_CACHE = {}
_DECODER_PLAN = {}       # (decoder shape, device) -> the index tensors of DecoderPacker
_TRANSFORMER_INDEX = {}  # (layer kinds, device) -> (int32 index [units, 512], per-layer unit spans, units)


def fragment_index(n_out, n_in, cols, nmb, device):
    """int64 [T, nmb, 64, 8] on `device`: positions in the zero-extended weight ext [(nmb*32) x (n_in+1)] (flattened) that the
    fragment stream gathers — the addressing of cond_nerf._fragments_h, cached per (shape, cols, device)."""
    cols = np.asarray(cols)
    key = (int(n_out), int(n_in), int(nmb), cols.shape, cols.tobytes(), str(device))
    if key not in _INDEX_CACHE:
        c = np.where(cols < 0, n_in, cols)
        lane = np.arange(64)
        col = c[:, lane >> 5, :]                                                      # [T, 64, 8]
        row = (lane & 31)[None, None, :, None] + 32 * np.arange(nmb)[None, :, None, None]
        t_n = c.shape[0]
        flat = np.broadcast_to(row, (t_n, nmb, 64, 8)) * (n_in + 1) + np.broadcast_to(col[:, None], (t_n, nmb, 64, 8))
        _INDEX_CACHE[key] = torch.from_numpy(np.ascontiguousarray(flat)).to(device)
    return _INDEX_CACHE[key]


def exponent_of_absmax(m):
    """cond_nerf.f16_weight_exponent for a host float: ew with max|w| * 2^ew in [2^13, 2^14); 0 for zero / non-finite."""
    if m == 0.0 or not np.isfinite(m):
        return 0
    return int(CN.F16_TARGET_EXP - np.frexp(np.float32(m))[1])


def weight_exponents(tensors):
    """Scale exponents of several weight tensors with ONE device->host copy."""
    if not tensors:
        return []
    amax = torch.stack([t.detach().abs().max() for t in tensors]).tolist()
    return [exponent_of_absmax(float(m)) for m in amax]


def fragments_h(weight, cols, nmb, ew):
    """fp16 [T, nmb, 2, 64, 8] on weight's device; `ew` a host integer.  Same bits as cond_nerf._fragments_h."""
    w = weight.detach().to(torch.float32)
    n_out, n_in = w.shape
    ext = w.new_zeros(nmb * 32, n_in + 1)
    ext[:n_out, :n_in] = w * float(2.0 ** int(ew))  # exact (a power of two as a host scalar; torch.ldexp goes through pow() on the device)
    g = ext.reshape(-1)[fragment_index(n_out, n_in, cols, nmb, w.device)]
    hi = g.to(torch.float16)
    lo = (g - hi.to(torch.float32)).to(torch.float16)
    return torch.stack([hi, lo], 2)


def _as_float_words(halfs):
    return torch.cat([h.reshape(-1) for h in halfs]).view(torch.float32)


_NATURAL = np.arange(128).reshape(8, 2, 8)


def pack_qkv(wq, wk, wv, ews=None):
    """gmflow.pack_qkv on the weights' device -> (wstream float32 words, (ew_q, ew_k, ew_v))."""
    from .gmflow import K_ROW_ORDER
    rows = torch.from_numpy(K_ROW_ORDER).to(wk.device)
    ws = [wq.detach(), wk.detach()[rows], wv.detach()]
    if ews is None:
        ews = weight_exponents(ws)
    return _as_float_words([fragments_h(w, _NATURAL, 4, e) for w, e in zip(ws, ews)]), tuple(int(e) for e in ews)


def pack_encoder_block(merge_w, mlp0_w=None, mlp2_w=None, ews=None):
    """gmflow.pack_encoder_block on the weights' device -> (wstream float32 words, (ew_merge, ew_w1, ew_w2))."""
    from .gmflow import EB_CHUNK, EB_SEG_FLOATS
    acc_order = CN._reg_cols16(4)
    ffn = mlp0_w is not None
    if ews is None:
        e = weight_exponents([merge_w] + ([mlp0_w, mlp2_w] if ffn else []))
        ews = (e[0], e[1], e[2]) if ffn else (e[0], 0, 0)
    parts = [fragments_h(merge_w, _NATURAL, 4, ews[0])]
    if ffn:
        assert tuple(mlp0_w.shape) == (1024, 256) and tuple(mlp2_w.shape) == (128, 1024)
        cols1 = np.concatenate([_NATURAL, 128 + acc_order], 0)
        for c in range(1024 // EB_CHUNK):
            parts.append(fragments_h(mlp0_w[EB_CHUNK * c:EB_CHUNK * (c + 1)], cols1, 4, ews[1]))
            parts.append(fragments_h(mlp2_w[:, EB_CHUNK * c:EB_CHUNK * (c + 1)], acc_order, 4, ews[2]))
    ws = _as_float_words(parts)
    assert ws.numel() % EB_SEG_FLOATS == 0
    return ws, tuple(int(e) for e in ews)


class TransformerPacker:
    """All operand streams of a gmflow.FeatureTransformer (12 layers: q|k|v streams + post-attention streams) from one gather.
    The index (int32, ~3.1 M entries for the shipped 6-block transformer) maps every fragment element to its position in the
    concatenation of the layers' scaled weight matrices (one trailing zero serves the padding); it depends on the layer shapes
    only and is built once per device."""

    def __init__(self, ft, device):
        self.device = torch.device(device)
        self.layers = [l for blk in ft.layers for l in (blk.self_attn, blk.cross_attn_ffn)]
        self.tensors = []
        for layer in self.layers:
            self.tensors += layer._qkv_params() + layer._block_weights()
        # the index depends on the layers' SHAPES only: shared by every transformer of that shape on the device (nn.DataParallel
        # makes a new replica object per forward; rebuilding 3 M index entries each time would cost half a second)
        sig = (tuple(bool(l.no_ffn) for l in self.layers), str(self.device))
        if sig not in _TRANSFORMER_INDEX:
            _TRANSFORMER_INDEX[sig] = self._build_index()
        self.index, self.spans, self.n_units = _TRANSFORMER_INDEX[sig]

    def _build_index(self):
        from .gmflow import EB_CHUNK, K_ROW_ORDER
        acc_order = CN._reg_cols16(4)
        cols1 = np.concatenate([_NATURAL, 128 + acc_order], 0)
        rows128 = np.arange(128)
        spans, chunks, off, pos = [], [], 0, 0

        def add_tensor(t):
            nonlocal off
            base, off = off, off + t.numel()
            return base

        def frag(base, n_cols_total, rows, cols):
            """[T * 4 units, 512] flat positions of one fragment: stream row r of block m <- weight row rows[32 m + r]"""
            c = np.asarray(cols)
            lane = np.arange(64)
            col = c[:, lane >> 5, :]                                                     # [T, 64, 8]
            row = rows[(lane & 31)[None, None, :, None] + 32 * np.arange(4)[None, :, None, None]]  # [1, 4, 64, 1]
            flat = base + np.broadcast_to(row, (c.shape[0], 4, 64, 8)) * n_cols_total + np.broadcast_to(col[:, None], (c.shape[0], 4, 64, 8))
            return flat.reshape(-1, 512)

        for layer in self.layers:
            start = pos
            for i, w in enumerate(layer._qkv_params()):
                chunks.append(frag(add_tensor(w), 128, K_ROW_ORDER if i == 1 else rows128, _NATURAL))
                pos += chunks[-1].shape[0]
            qkv_span = (start, pos)
            start = pos
            chunks.append(frag(add_tensor(layer.merge.weight), 128, rows128, _NATURAL))
            pos += chunks[-1].shape[0]
            if not layer.no_ffn:
                b0, b2 = add_tensor(layer.mlp[0].weight), add_tensor(layer.mlp[2].weight)
                assert tuple(layer.mlp[0].weight.shape) == (1024, 256) and tuple(layer.mlp[2].weight.shape) == (128, 1024)
                for c in range(1024 // EB_CHUNK):
                    chunks.append(frag(b0, 256, EB_CHUNK * c + rows128, cols1))
                    chunks.append(frag(b2, 1024, rows128, EB_CHUNK * c + acc_order))
                    pos += chunks[-2].shape[0] + chunks[-1].shape[0]
            spans.append((qkv_span, (start, pos)))
        idx = np.concatenate(chunks, 0)
        assert idx.max() < off < 2 ** 31
        return torch.from_numpy(idx.astype(np.int32)).to(self.device), spans, pos

    def matches(self, ft):
        layers = [l for blk in ft.layers for l in (blk.self_attn, blk.cross_attn_ffn)]
        return len(layers) == len(self.layers) and all(a is b for a, b in zip(layers, self.layers))

    def pack(self):
        """-> per layer ((qkv stream, (ew_q, ew_k, ew_v)), (block stream, (ew_merge, ew_w1, ew_w2))); streams are float32-word
        views of one fresh buffer.  One device->host copy (the exponents)."""
        # the parameter OBJECTS are collected again on every pack: a layer whose Parameter was replaced since construction
        # (layer.q_proj.weight = nn.Parameter(...), parametrize, weight tying) must not be packed from the old tensor
        self.tensors = [t for layer in self.layers for t in layer._qkv_params() + layer._block_weights()]
        ts = [t.detach() for t in self.tensors]
        ews = [exponent_of_absmax(float(m)) for m in torch.stack(torch._foreach_norm(ts, float("inf"))).tolist()]
        scaled = torch._foreach_mul(ts, [float(2.0 ** e) for e in ews])
        flat = torch.cat([t.reshape(-1) for t in scaled])
        g = flat[self.index.long()]                                   # [units, 512]
        hi = g.to(torch.float16)
        lo = (g - hi.to(torch.float32)).to(torch.float16)
        words = torch.stack([hi, lo], 1).reshape(-1).view(torch.float32)  # [units][hi | lo][512] halves = 512 words per unit
        out, k = [], 0
        for layer, (qs, bs) in zip(self.layers, self.spans):
            e_qkv = tuple(ews[k:k + 3])
            k += 3
            if layer.no_ffn:
                e_blk, k = (ews[k], 0, 0), k + 1
            else:
                e_blk, k = (ews[k], ews[k + 1], ews[k + 2]), k + 3
            out.append(((words[qs[0] * 512:qs[1] * 512], e_qkv), (words[bs[0] * 512:bs[1] * 512], e_blk)))
        return out


_CONV_INDEX = {}  # (conv shapes, device) -> (int32 index [units, 512], spans)


class ConvPacker:
    """The split-fp16 weight streams of a list of nn.Conv2d (gmflow.pack_conv / pack_conv_stem) for the TRAINING path, from one gather:
    per convolution the forward stream W[co, tap * c_in + c] and, for stride-1 convolutions whose data gradient is itself a
    convolution the forward kernel builds (c_in 64 / 96 / 128), the stream of the flipped, transposed filter
    W'[ci, tap' * c_out + co] = W[co, ci, k-1-ky', k-1-kx'] - ``mnerf_conv2d`` over dY with it IS dX.  One device->host copy per
    pack (the exponents); the index depends on the shapes only."""

    def __init__(self, convs, device):
        self.device = torch.device(device)
        self.convs = list(convs)
        sig = (tuple((c.out_channels, c.in_channels, c.kernel_size[0], c.stride[0]) for c in self.convs), str(self.device))
        if sig not in _CONV_INDEX:
            _CONV_INDEX[sig] = self._build_index()
        self.index, self.spans = _CONV_INDEX[sig]

    @staticmethod
    def has_backward_stream(conv):
        return conv.stride[0] == 1 and conv.kernel_size[0] in (1, 3) and conv.in_channels in (64, 96, 128) and conv.out_channels % 32 == 0

    def _build_index(self):
        chunks, spans, off, pos = [], [], 0, 0
        lane = np.arange(64)

        def frag(flat_of, n_rows, n_cols, cols):
            """[T * nmb units, 512]: positions of the fragment elements; flat_of(row, col) -> position in the concatenation (or the
            zero slot, marked -1, for padding)"""
            nmb = n_rows // 32
            c = np.asarray(cols)                                           # [T, 2, 8]
            col = np.broadcast_to(c[:, lane >> 5, :][:, None], (c.shape[0], nmb, 64, 8))
            row = np.broadcast_to((lane & 31)[None, None, :, None] + 32 * np.arange(nmb)[None, :, None, None], col.shape)
            flat = np.where(col < n_cols, flat_of(row, np.minimum(col, n_cols - 1)), -1)
            return flat.reshape(-1, 512)

        for conv in self.convs:
            co, ci, k = conv.out_channels, conv.in_channels, conv.kernel_size[0]
            base, off = off, off + co * ci * k * k

            def fwd(row, col, base=base, ci=ci, k=k):      # row = co, col = tap * ci + c  ->  W[co][c][ky][kx]
                tap, c = col // ci, col % ci
                return base + ((row * ci + c) * k + tap // k) * k + tap % k

            def bwd(row, col, base=base, ci=ci, co=co, k=k):  # row = ci, col = tap' * co + o  ->  W[o][ci][k-1-ky'][k-1-kx']
                tap, o = col // co, col % co
                return base + ((o * ci + row) * k + (k - 1 - tap // k)) * k + (k - 1 - tap % k)

            n_f = k * k * ci
            steps_f = (n_f + 15) // 16
            chunks.append(frag(fwd, co, n_f, np.arange(steps_f * 16).reshape(-1, 2, 8)))
            f_span = (pos, pos + chunks[-1].shape[0])
            pos = f_span[1]
            b_span = None
            if self.has_backward_stream(conv):
                n_b = k * k * co
                chunks.append(frag(bwd, ci, n_b, np.arange(n_b).reshape(-1, 2, 8)))
                b_span = (pos, pos + chunks[-1].shape[0])
                pos = b_span[1]
            spans.append((f_span, b_span))
        idx = np.concatenate(chunks, 0)
        idx = np.where(idx < 0, off, idx)  # the zero slot behind the last weight
        assert idx.max() <= off < 2 ** 31
        return torch.from_numpy(idx.astype(np.int32)).to(self.device), spans

    def pack(self):
        """-> per convolution (forward stream, backward stream or None, ew): float32-word views of one fresh buffer"""
        ts = [c.weight.detach() for c in self.convs]
        ews = [exponent_of_absmax(float(m)) for m in torch.stack(torch._foreach_norm(ts, float("inf"))).tolist()]
        scaled = torch._foreach_mul(ts, [float(2.0 ** e) for e in ews])
        flat = torch.cat([t.reshape(-1) for t in scaled] + [ts[0].new_zeros(1)])
        g = flat[self.index.long()]
        hi = g.to(torch.float16)
        lo = (g - hi.to(torch.float32)).to(torch.float16)
        words = torch.stack([hi, lo], 1).reshape(-1).view(torch.float32)
        return [(words[f[0] * 512:f[1] * 512], None if b is None else words[b[0] * 512:b[1] * 512], e) for (f, b), e in zip(self.spans, ews)]


class DecoderPacker:
    """The conditional-NeRF decoder's split-fp16 weight stream (cond_nerf.pack_wstream_h) assembled on the device, with NO
    device->host copy at all: this stream carries its scale exponents in the stage headers (float [128] = 2^-ew, [129] = ew), so
    they are computed where the weights are (frexp of the tensor's absmax).  Built once per decoder shape: index maps from the
    numpy packer's own helpers, evaluated on position-valued stand-ins for the parameters (every fp32 word of the stream that is
    not a fragment half is a plain copy of one parameter element: the bias headers and the resident tail segment)."""

    def __init__(self, dec, n_views, cos_n_group, L_3D, legacy, device, prefix="nerf_dec."):
        self.device = torch.device(device)
        sd = {prefix + k: v for k, v in dec.state_dict().items()}
        self.names = [k for k in sd if not k.endswith("num_batches_tracked")]
        self._dec_ref, self._prefix = dec, prefix
        self.params = [dict(dec.named_parameters())[k[len(prefix):]] for k in self.names]
        # the plan depends on the decoder's shape only: shared by every decoder of that shape on the device (DataParallel replicas)
        key = (int(n_views), tuple(int(g) for g in cos_n_group), int(L_3D), bool(legacy), str(self.device),
               tuple((k, tuple(p.shape)) for k, p in zip(self.names, self.params)))
        if key not in _DECODER_PLAN:
            _DECODER_PLAN[key] = self._build_plan(n_views, cos_n_group, L_3D, legacy, prefix)
        (self.zero_pos, self.cond_dim, self.cond_stride, self.total, self.frag_src, self.frag_dst, self.frag_tensor, self.word_src,
         self.word_dst, self.hdr_dst, self.hdr_tensor) = _DECODER_PLAN[key]

    def _build_plan(self, n_views, cos_n_group, L_3D, legacy, prefix):
        base, pos_sd = {}, {}
        off = 0
        for k, p in zip(self.names, self.params):
            base[k] = off
            pos_sd[k] = (off + 1 + np.arange(p.numel(), dtype=np.float64)).reshape(tuple(p.shape)).astype(np.float32)
            off += p.numel()
        assert off + 1 < 2 ** 24, "position-valued stand-ins must be exact in float32"
        self.zero_pos = off  # flat = cat(params) + [0]

        def to_pos(a):  # position + 1 (0 = padding) -> flat index
            a = np.asarray(a).astype(np.int64)
            return np.where(a == 0, self.zero_pos, a - 1)

        cond_dim = int(sum(cos_n_group)) + 4 * n_views
        self.cond_dim, self.cond_stride = cond_dim, ((cond_dim + 1 + 7) // 8) * 8
        d_enc = 3 + 6 * L_3D
        tf = (cond_dim + 15) // 16
        film_cols = np.arange(16 * tf).reshape(tf, 2, 8)
        film_cols = np.where(film_cols < cond_dim, film_cols, CN.ZERO)
        e_cols, h_cols = CN._enc_cols16(L_3D, legacy), CN._reg_cols16(4)
        dir_cols = np.full((1, 2, 8), CN.ZERO, np.int64)
        dir_cols[0, 0, :3] = [128, 129, 130]
        w5 = prefix + "pts_linears.5.weight"
        # stage -> (weight tensor, first column of the slice, columns in the slice, bias tensor or None, cols, scale tensor)
        stages = {"film": (prefix + "pts_bias.weight", 0, None, prefix + "pts_bias.bias", film_cols),
                  "l0": (prefix + "pts_linears.0.weight", 0, None, prefix + "pts_linears.0.bias", e_cols),
                  "l5e": (w5, 0, d_enc, prefix + "pts_linears.5.bias", e_cols),
                  "l5h": (w5, d_enc, 128, None, h_cols),
                  "feature": (prefix + "feature_linear.weight", 0, None, prefix + "feature_linear.bias", h_cols),
                  "views": (prefix + "views_linears.0.weight", 0, None, prefix + "views_linears.0.bias", np.concatenate([h_cols, dir_cols], 0)),
                  "rgb": (prefix + "rgb_linear.weight", 0, None, prefix + "rgb_linear.bias", CN._reg_cols16(2)),
                  "alpha": (prefix + "alpha_linear.0.weight", 0, None, prefix + "alpha_linear.0.bias", h_cols)}
        for i in range(1, 5):
            stages[f"l{i}"] = (prefix + f"pts_linears.{i}.weight", 0, None, prefix + f"pts_linears.{i}.bias", h_cols)
        segs, total = CN.decoder_schedule_h(cond_dim, L_3D)
        self.total = total
        tensor_id = {k: i for i, k in enumerate(self.names)}
        frag_src, frag_dst, frag_tensor = [], [], []
        word_src, word_dst, hdr_dst, hdr_tensor = [], [], [], []
        frag_cache = {}
        for name, first, steps, m, off_w, fl, hdr in segs:
            if name == "tail":
                continue
            wname, c0, ncols, bname, cols = stages[name]
            wt = pos_sd[wname]
            n_out, n_in_total = wt.shape
            n_in = n_in_total - c0 if ncols is None else ncols
            if name not in frag_cache:
                c = np.where(np.asarray(cols) < 0, -1, np.asarray(cols))
                lane = np.arange(64)
                col = c[:, lane >> 5, :]                                                       # [T, 64, 8]
                row = (lane & 31)[None, None, :, None] + 32 * np.arange(m)[None, :, None, None]  # [1, m, 64, 1]
                t_n = c.shape[0]
                colb = np.broadcast_to(col[:, None], (t_n, m, 64, 8))
                rowb = np.broadcast_to(row, (t_n, m, 64, 8))
                valid = (rowb < n_out) & (colb >= 0) & (colb < n_in)
                flat = base[wname] + rowb * n_in_total + c0 + np.where(valid, colb, 0)
                frag_cache[name] = np.where(valid, flat, self.zero_pos).reshape(t_n * m, 512)
            if hdr:
                b = np.zeros(0, np.float32) if bname is None else pos_sd[bname]
                word_src.append(to_pos(CN._bias_fragment(b if bname is not None else None, m)[:128]))
                word_dst.append(off_w + np.arange(128))
                hdr_dst.append(off_w + 128)
                hdr_tensor.append(tensor_id[wname])
                off_w += CN.FRAG_FLOATS
            units = frag_cache[name][first * m:(first + steps) * m]
            frag_src.append(units)
            frag_dst.append(off_w + 512 * np.arange(units.shape[0])[:, None] + np.arange(512)[None, :])
            frag_tensor.append(np.full(units.shape[0], tensor_id[wname]))
        # the resident tail segment: plain fp32 copies, taken from the fp32-format packer run on the stand-ins
        tail_words = ((CN.TAIL_FLOATS + 255) // 256) * 256
        f32_stream = CN.pack_wstream(pos_sd, n_views, cos_n_group, L_3D, legacy, prefix)[0]
        word_src.append(to_pos(f32_stream[-tail_words:]))
        word_dst.append(segs[-1][4] + np.arange(tail_words))
        assert segs[-1][0] == "tail"
        dev = self.device
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a).astype(np.int64)).to(dev)
        return (self.zero_pos, self.cond_dim, self.cond_stride, self.total, t(np.concatenate(frag_src, 0)), t(np.concatenate(frag_dst, 0)),
                t(np.concatenate(frag_tensor)), t(np.concatenate(word_src)), t(np.concatenate(word_dst)),
                torch.tensor(hdr_dst, dtype=torch.int64, device=dev), torch.tensor(hdr_tensor, dtype=torch.int64, device=dev))

    def pack(self):
        """-> wstream (float32 words [total]) on the parameters' device; same bits as cond_nerf.pack_wstream_h."""
        # (the Parameter objects are looked up again on every pack: see TransformerPacker.pack)
        named = dict(self._dec_ref.named_parameters())
        self.params = [named[k[len(self._prefix):]] for k in self.names]
        ts = [p.detach().to(torch.float32) for p in self.params]
        amax = torch.stack(torch._foreach_norm(ts, float("inf")))
        _, e = torch.frexp(amax)
        ew = torch.where((amax == 0) | ~torch.isfinite(amax), torch.zeros_like(e), CN.F16_TARGET_EXP - e).to(torch.int32)
        scale = ((ew + 127) << 23).view(torch.float32)          # 2^ew, exact
        inv = ((127 - ew) << 23).view(torch.float32)            # 2^-ew
        flat = torch.cat([t.reshape(-1) for t in ts] + [amax.new_zeros(1)])
        g = flat[self.frag_src] * scale[self.frag_tensor][:, None]
        hi = g.to(torch.float16)
        lo = (g - hi.to(torch.float32)).to(torch.float16)
        out = flat.new_zeros(self.total)
        out[self.frag_dst.reshape(-1)] = torch.stack([hi, lo], 1).reshape(-1).view(torch.float32)
        out[self.word_dst] = flat[self.word_src]
        out[self.hdr_dst] = inv[self.hdr_tensor]
        out[self.hdr_dst + 1] = ew[self.hdr_tensor].to(torch.float32)
        return out
