"""GMFlow pair-wise feature encoder of MatchNeRF, MI355X-native host orchestration.

Mirrors the parameter structure (and therefore the ``state_dict`` keys — SURVEY.md Appendix B)
of /root/reference/models/gmflow/{gmflow,backbone,transformer,superres,position}.py, with a
forward pass designed around what the render kernels consume:

* tokens are channel-last ``[sequence, h*w, 128]`` from the backbone output onwards, so the
  transformer's result *is* the pair-major channel-last feature map ``[P,2,h,w,128]`` that
  ``mnerf_cost_volume`` samples (include/mnerf.h) — no NCHW round trip, no per-view
  ``torch.cat`` regrouping (reference: matchnerf.py:192-205);
* the window sine position tile depends only on the token position, so it is added once per
  VIEW before pairs are formed (reference adds it per pair member, gmflow/utils.py:68-88);
* the six swin self/cross attention layers call the flash-style HIP kernel
  ``mnerf_window_attention`` (K6): roll, window split/merge and the shift mask are index
  arithmetic in the kernel, the [24,1280,1280] score tensor is never materialised
  (reference: transformer.py:46-105);
* the backbone and up-sampler convolutions after the 7x7 stem are split-fp16 implicit GEMMs (``mnerf_conv2d``), every
  InstanceNorm + activation (+ residual add) one kernel (``mnerf_instance_norm``), everything after the attention
  inside a transformer layer one kernel (``mnerf_encoder_block``), the q|k|v projections one kernel
  (``mnerf_qkv_projection``): no library GEMM or convolution is left in the inference path.  Under autograd the
  reference's op chain (MIOpen, rocBLAS) is used throughout.

There is no CPU path: ``forward`` needs the HIP library and a GPU tensor.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip  # noqa: F401  (the encoder has no path without the HIP library)
from .autograd import window_attention
from .camera import pair_list


def _fused_norm(x):
    """the fused InstanceNorm kernel serves inference on the GPU; under autograd the torch op chain is kept"""
    return x.is_cuda and not torch.is_grad_enabled()


def _train_hip(x):
    """Training path (autograd on) of the CNN on this library's kernels - forward, data and weight gradients of every convolution and the
    InstanceNorm backward (csrc/conv_backward.hip, instance_norm.hip; forward and stride-1 data gradients on the split-fp16
    convolution of inference, csrc/conv.hip) - instead of torch's op chain (MIOpen convolutions, ATen norms).
    ``MNERF_TRAIN_CNN=torch`` restores that chain, ``=f32`` keeps every convolution on the exact-f32 kernels."""
    import os
    return x.is_cuda and torch.is_grad_enabled() and os.environ.get("MNERF_TRAIN_CNN", "hip") != "torch"


def _train_packs(owner, convs, device):
    """Attach this step's split-fp16 weight streams (packing.ConvPacker: one gather + one device->host copy for all of them) to the
    convolutions of ``owner``'s training path; re-packed when a parameter changed (``MNERF_TRAIN_CNN=f32``: none - exact-f32
    kernels for the forward too)."""
    import os
    if os.environ.get("MNERF_TRAIN_CNN", "hip") == "f32":
        for c in convs:
            c._mnerf_train_pack = None
        return
    from . import packing
    key = (tuple((int(c.weight._version), int(c.weight.data_ptr())) for c in convs), str(device))
    if getattr(owner, "_train_pack_key", None) != key:
        pk = getattr(owner, "_train_packer", None)
        if pk is None or pk.device != torch.device(device) or len(pk.convs) != len(convs) or any(a is not b for a, b in zip(pk.convs, convs)):
            pk = owner._train_packer = packing.ConvPacker(convs, device)
        owner._train_streams = pk.pack()
        owner._train_pack_key = key
    # fresh (zeroed) absmax regions for this forward / backward pair: one fill for all convolutions
    regs = hip.absmax_regions(2 * len(convs), device).reshape(len(convs), 2, -1)
    for i, (c, st) in enumerate(zip(convs, owner._train_streams)):
        c._mnerf_train_pack = (st[0], st[1], st[2], regs[i])


def _conv_out(conv, x):
    """library convolution result as a plain NCHW-contiguous fp32 tensor (what ``mnerf_instance_norm`` reads plane by plane)"""
    return conv(x).contiguous()


def pack_conv(weight):
    """Conv2d weight [c_out, c_in, k, k] -> (wstream float32 words, ew) for ``mnerf_conv2d`` (csrc/conv.hip): the
    implicit-GEMM matrix W[out, tap * c_in + c] (tap = ky * k + kx) as split-fp16 A-operand fragments, K16-step s =
    16 consecutive columns, unit (s, 32-row block) = [hi | lo] x 64 lanes x 8 (the decoder's unit layout,
    cond_nerf._fragments_h)."""
    import numpy as np
    from . import cond_nerf as CN
    w = (weight.detach().cpu().numpy() if torch.is_tensor(weight) else np.asarray(weight)).astype(np.float32)
    c_out, c_in, kh, kw = w.shape
    assert kh == kw and c_in % 32 == 0 and c_out % 32 == 0, w.shape
    mat = np.ascontiguousarray(w.transpose(0, 2, 3, 1).reshape(c_out, kh * kw * c_in))
    ew = CN.f16_weight_exponent(mat)
    cols = np.arange(mat.shape[1]).reshape(-1, 2, 8)
    halfs = CN._fragments_h(mat, cols, c_out // 32, ew)            # [steps, blocks, 2, 64, 8] fp16
    return halfs.reshape(-1).view(np.float32).copy(), int(ew)


def pack_conv_stem(weight):
    """The stem's weight [64, 3, 7, 7] -> (wstream, ew) for ``mnerf_conv_stem``: matrix [64, 3 tap + c], 147 columns
    padded with zeros to ten K16-steps."""
    import numpy as np
    from . import cond_nerf as CN
    w = (weight.detach().cpu().numpy() if torch.is_tensor(weight) else np.asarray(weight)).astype(np.float32)
    assert w.shape == (64, 3, 7, 7), w.shape
    mat = np.zeros((64, 160), np.float32)
    mat[:, :147] = w.transpose(0, 2, 3, 1).reshape(64, 147)
    ew = CN.f16_weight_exponent(mat)
    halfs = CN._fragments_h(mat, np.arange(160).reshape(10, 2, 8), 2, ew)
    return halfs.reshape(-1).view(np.float32).copy(), int(ew)


def _hip_conv(conv, x, in_absmax, **kw):
    """``conv`` (nn.Conv2d with a shape conv.hip builds) through ``mnerf_conv2d``; the packed weights are cached on the
    module and re-packed when the parameter changed."""
    ps = [conv.weight] + ([conv.bias] if conv.bias is not None else [])
    key = (tuple(int(p._version) for p in ps), tuple(int(p.data_ptr()) for p in ps), str(x.device))
    if getattr(conv, "_mnerf_pack", None) is None or conv._mnerf_pack[0] != key:
        ws, ew = pack_conv(conv.weight)
        bias = conv.bias.detach().float().contiguous().to(x.device) if conv.bias is not None else None
        conv._mnerf_pack = (key, torch.from_numpy(ws).to(x.device), bias, ew)
    _, ws, bias, ew = conv._mnerf_pack
    return hip.conv2d(x, ws, bias, conv.in_channels, conv.out_channels, conv.kernel_size[0], conv.stride[0], ew, in_absmax, **kw)


class ResidualBlock(nn.Module):
    """backbone.py:6-36 (InstanceNorm2d without affine => no parameters for the norms)."""

    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, 3, stride=stride, padding=1, bias=False)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.stride = stride
        if stride != 1 or in_planes != planes:
            # index 0 carries the parameters; the InstanceNorm that follows it in the reference
            # (nn.Sequential(conv, norm3)) has none and is applied functionally below
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, 1, stride=stride))
        else:
            self.downsample = None

    def forward_fused(self, x, x_absmax, scal):
        """Inference: split-fp16 MFMA convolutions (conv.hip) and one norm + activation (+ residual) kernel each
        (instance_norm.hip).  ``x_absmax``: absmax region filled by x's producer; ``scal``: two zeroed regions for this
        block's own intermediate / output maxima.  Returns (out, region holding max|out|)."""
        y = _hip_conv(self.conv1, x, x_absmax)
        hip.instance_norm(y, relu_inner=True, out=y, out_absmax=scal[0])
        y = _hip_conv(self.conv2, y, scal[0])
        if self.downsample is not None:
            x = _hip_conv(self.downsample[0], x, x_absmax)
            hip.instance_norm(x, out=x)
        hip.instance_norm(y, residual=x, relu_inner=True, relu_outer=True, out=y, out_absmax=scal[1])
        return y, scal[1]

    def forward(self, x):
        if _train_hip(x):
            from . import autograd as AG
            y = AG.instance_norm(AG.conv2d(self.conv1, x), relu=True)
            y = AG.instance_norm(AG.conv2d(self.conv2, y), relu=True)
            if self.downsample is not None:
                x = AG.instance_norm(AG.conv2d(self.downsample[0], x))
            return F.relu(x + y)
        y = F.relu(F.instance_norm(self.conv1(x)))
        y = F.relu(F.instance_norm(self.conv2(y)))
        if self.downsample is not None:
            x = F.instance_norm(self.downsample(x))
        return F.relu(x + y)


class CNNEncoder(nn.Module):
    """backbone.py:39-122 with num_output_scales=1: stride-8, 128-channel features."""

    def __init__(self, output_dim=128):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.layer1 = nn.Sequential(ResidualBlock(64, 64, 1), ResidualBlock(64, 64, 1))
        self.layer2 = nn.Sequential(ResidualBlock(64, 96, 2), ResidualBlock(96, 96, 1))
        self.layer3 = nn.Sequential(ResidualBlock(96, 128, 2), ResidualBlock(128, 128, 1))
        self.conv2 = nn.Conv2d(128, output_dim, 1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x, tokens_plus=None):
        """``tokens_plus`` (inference): a [h*w, C] tile; the result is then the transformer's channel-last tokens
        [N,h,w,C] with the tile added - layout change and position embedding are the last convolution's epilogue."""
        if _fused_norm(x):
            # every convolution (stem: mnerf_conv_stem, the other 14: mnerf_conv2d) and all 15 norms are HIP kernels
            scal = hip.absmax_regions(14, x.device)  # max|.| of every convolution input (one fill kernel)
            x = x.contiguous()
            hip.absmax(x, scal[13])
            key = (int(self.conv1.weight._version), int(self.conv1.weight.data_ptr()), str(x.device))
            if getattr(self, "_stem_pack", None) is None or self._stem_pack[0] != key:
                ws, ew = pack_conv_stem(self.conv1.weight)
                self._stem_pack = (key, torch.from_numpy(ws).to(x.device), ew)
            x = hip.conv_stem(x, self._stem_pack[1], self._stem_pack[2], scal[13])
            hip.instance_norm(x, relu_inner=True, out=x, out_absmax=scal[0])
            amax, k = scal[0], 1
            for layer in (self.layer1, self.layer2, self.layer3):
                for blk in layer:
                    x, amax = blk.forward_fused(x, amax, scal[k:k + 2])
                    k += 2
            if tokens_plus is not None:
                return _hip_conv(self.conv2, x, amax, out_layout=hip.CONV_OUT_CHANNEL_LAST, add_channel_last=tokens_plus)
            return _hip_conv(self.conv2, x, amax)
        if _train_hip(x):
            from . import autograd as AG
            _train_packs(self, [m for m in self.modules() if isinstance(m, nn.Conv2d)], x.device)
            x = AG.instance_norm(AG.conv2d(self.conv1, x), relu=True)
            x = AG.conv2d(self.conv2, self.layer3(self.layer2(self.layer1(x))))
        else:
            x = F.relu(F.instance_norm(self.conv1(x)))
            x = self.conv2(self.layer3(self.layer2(self.layer1(x))))
        return x if tokens_plus is None else x.permute(0, 2, 3, 1) + tokens_plus.reshape(x.shape[2], x.shape[3], -1)


EB_SEG_FLOATS = 32 * 256  # one weight segment of the encoder block kernel: 4 K16-steps x 4 blocks x [hi | lo] x 1 KiB
EB_CHUNK = 128            # hidden units of the FFN produced and consumed at a time


def pack_encoder_block(merge_w, mlp0_w=None, mlp2_w=None):
    """Weights of one transformer layer's post-attention chain -> (wstream float32 words, (ew_merge, ew_w1, ew_w2)) in the
    split-fp16 MFMA A-fragment layout consumed by ``mnerf_encoder_block`` (csrc/encoder_block.hip; same unit layout as
    the decoder's stream, cond_nerf.pack_wstream_h).  Segment order: merge (2 segments: input features 0-63, 64-127 in
    natural order), then for every 128-unit hidden chunk c: mlp.0 rows [128c, 128c+128) (4 segments: the 128 source
    features in natural order, then the 128 message features in accumulator order) and mlp.2 columns [128c, 128c+128)
    (2 segments, accumulator order)."""
    import numpy as np
    from . import cond_nerf as CN

    def npf(w):
        return (w.detach().cpu().numpy() if torch.is_tensor(w) else np.asarray(w)).astype(np.float32)

    natural = np.arange(128).reshape(8, 2, 8)          # K16-step t, half h, j -> feature 16 t + 8 h + j
    acc_order = CN._reg_cols16(4)                      # accumulator-register order of a 128-row stage
    merge_w = npf(merge_w)
    ew_m = CN.f16_weight_exponent(merge_w)
    parts = [CN._fragments_h(merge_w, natural, 4, ew_m)]                      # [8, 4, 2, 64, 8]
    ews = [ew_m, 0, 0]
    if mlp0_w is not None:
        w1, w2 = npf(mlp0_w), npf(mlp2_w)
        assert w1.shape == (1024, 256) and w2.shape == (128, 1024)
        ews[1], ews[2] = CN.f16_weight_exponent(w1), CN.f16_weight_exponent(w2)
        cols1 = np.concatenate([natural, 128 + acc_order], 0)                # 16 steps over cat[source, message]
        for c in range(1024 // EB_CHUNK):
            parts.append(CN._fragments_h(w1[EB_CHUNK * c:EB_CHUNK * (c + 1)], cols1, 4, ews[1]))
            parts.append(CN._fragments_h(w2[:, EB_CHUNK * c:EB_CHUNK * (c + 1)], acc_order, 4, ews[2]))
    halfs = np.concatenate([p.reshape(-1) for p in parts])                    # fp16, [steps][4][2][64][8] per stage
    ws = halfs.view(np.float32).copy()
    assert ws.size % EB_SEG_FLOATS == 0
    return ws, tuple(int(e) for e in ews)


def _k_row_order():
    """Row permutation of Wk for ``mnerf_qkv_projection`` / ``mnerf_qkv_window_images``: accumulator row 32 m + n of the
    transposed chain is register r = (n & 3) + 4 (n >> 3) of lane half (n >> 2) & 1; it is given output channel
    16 t + 8 half + j with 8 t + j = 16 m + r, so that a lane's registers 8t..8t+7 are the eight channels it supplies to
    K16-step t of the attention's S^T = K Q^T."""
    import numpy as np
    rows = np.zeros(128, np.int64)
    for m in range(4):
        for n in range(32):
            r, half = (n & 3) + 4 * (n >> 3), (n >> 2) & 1
            q = 16 * m + r
            rows[32 * m + n] = 16 * (q >> 3) + 8 * half + (q & 7)
    assert sorted(rows.tolist()) == list(range(128))
    return rows


K_ROW_ORDER = _k_row_order()


def pack_qkv(wq, wk, wv):
    """q/k/v projection weights [128,128] -> (wstream float32 words, (ew_q, ew_k, ew_v)) for ``mnerf_qkv_projection``
    (csrc/qkv.hip): three matrices of 8 K16-steps x 4 row blocks, input features in natural order, each with its own
    power-of-two scale."""
    import numpy as np
    from . import cond_nerf as CN
    natural = np.arange(128).reshape(8, 2, 8)
    parts, ews = [], []
    for i, w in enumerate((wq, wk, wv)):
        w = (w.detach().cpu().numpy() if torch.is_tensor(w) else np.asarray(w)).astype(np.float32)
        assert w.shape == (128, 128), w.shape
        if i == 1:
            w = w[K_ROW_ORDER]  # accumulator registers 8t..8t+7 of a lane = the attention's K16-step t (csrc/qkv.hip)
        ews.append(CN.f16_weight_exponent(w))
        parts.append(CN._fragments_h(w, natural, 4, ews[-1]).reshape(-1))
    return np.concatenate(parts).view(np.float32).copy(), tuple(int(e) for e in ews)


class TransformerLayer(nn.Module):
    """transformer.py:108-185 (single head, swin windows).  Inference runs the q|k|v projections as one kernel
    (``mnerf_qkv_projection``), the window-attention kernel (K6) and ONE kernel for everything after it (K7, ``mnerf_encoder_block``: merge, LayerNorm,
    concatenation, FFN with exact GELU, LayerNorm, residual); under autograd the reference's op chain is used."""

    def __init__(self, d_model=128, no_ffn=False, ffn_dim_expansion=4):
        super().__init__()
        self.no_ffn = no_ffn
        self.q_proj = nn.Linear(d_model, d_model, bias=False)
        self.k_proj = nn.Linear(d_model, d_model, bias=False)
        self.v_proj = nn.Linear(d_model, d_model, bias=False)
        self.merge = nn.Linear(d_model, d_model, bias=False)
        self.norm1 = nn.LayerNorm(d_model)
        if not no_ffn:
            self.mlp = nn.Sequential(nn.Linear(2 * d_model, 2 * d_model * ffn_dim_expansion, bias=False), nn.GELU(),
                                     nn.Linear(2 * d_model * ffn_dim_expansion, d_model, bias=False))
            self.norm2 = nn.LayerNorm(d_model)

    def _block_params(self):
        ps = [self.merge.weight, self.norm1.weight, self.norm1.bias]
        if not self.no_ffn:
            ps += [self.mlp[0].weight, self.mlp[2].weight, self.norm2.weight, self.norm2.bias]
        return ps

    def _qkv_params(self):
        return [self.q_proj.weight, self.k_proj.weight, self.v_proj.weight]

    @staticmethod
    def _pack_key(ps, device):
        return (tuple(int(p._version) for p in ps), tuple(int(p.data_ptr()) for p in ps), str(device))

    def _block_weights(self):
        return [self.merge.weight] + ([] if self.no_ffn else [self.mlp[0].weight, self.mlp[2].weight])

    def _packed_block(self, device, ews=None):
        """(wstream, ln, ews) of the post-attention chain for the K7 kernel; re-packed when a parameter changed.  Parameters on
        a GPU are packed THERE (packing.py: a gather + the fp16 split as device ops; `ews`: the scale exponents if the caller
        already fetched them — FeatureTransformer.refresh_packs does that for all layers with one device->host copy), host
        parameters through the numpy packer."""
        ps = self._block_params()
        key = self._pack_key(ps, device)
        if getattr(self, "_blk", None) is None or self._blk[0] != key:
            w0, w2 = (None, None) if self.no_ffn else (self.mlp[0].weight, self.mlp[2].weight)
            if self.merge.weight.is_cuda:
                from . import packing
                ws, ews = packing.pack_encoder_block(self.merge.weight, w0, w2, ews)
                ws = ws.to(device)
            else:
                ws, ews = pack_encoder_block(self.merge.weight, w0, w2)
                ws = torch.from_numpy(ws).to(device)
            n2 = (self.norm1 if self.no_ffn else self.norm2)
            ln = torch.stack([self.norm1.weight, self.norm1.bias, n2.weight, n2.bias], 0).detach().float().contiguous()
            self._blk = (key, ws, ln.to(device), ews)
        return self._blk[1:]

    def _packed_qkv(self, device, ews=None):
        ps = self._qkv_params()
        key = self._pack_key(ps, device)
        if getattr(self, "_qkv", None) is None or self._qkv[0] != key:
            if ps[0].is_cuda:
                from . import packing
                ws, ews = packing.pack_qkv(*ps, ews=ews)
                ws = ws.to(device)
            else:
                ws, ews = pack_qkv(*ps)
                ws = torch.from_numpy(ws).to(device)
            self._qkv = (key, ws, ews)
        return self._qkv[1:]

    def stale_packs(self, device):
        """which of ('qkv', 'block') would be re-packed by the next forward on `device`"""
        out = []
        if getattr(self, "_qkv", None) is None or self._qkv[0] != self._pack_key(self._qkv_params(), device):
            out.append("qkv")
        if getattr(self, "_blk", None) is None or self._blk[0] != self._pack_key(self._block_params(), device):
            out.append("block")
        return out

    _warned_eval_grad = False

    def forward(self, source, target, h, w, splits, shifted, kv_swap=False):
        """``kv_swap`` (inference only): ``target`` is given un-swapped and read with its batch halves exchanged"""
        # ONE decision per layer: if anything this layer touches needs a gradient, the whole layer takes the autograd
        # path (with partial freezing, e.g. merge frozen but mlp / norm2 trainable, the fused kernels would leave the
        # trainable parameters without a gradient and without an error)
        needs_grad = torch.is_grad_enabled() and (source.requires_grad or target.requires_grad or
                                                  any(p.requires_grad for p in self.parameters()))
        if needs_grad and not self.training and not TransformerLayer._warned_eval_grad:
            TransformerLayer._warned_eval_grad = True
            import warnings
            warnings.warn("GMFlow transformer layer in eval() mode with autograd enabled: it records an autograd graph (saved "
                          "activations, the training form of the layer). Run inference under torch.no_grad().")
        if source.is_cuda and not needs_grad:
            ws, ews = self._packed_qkv(source.device)
            if hip.wa_math() == hip.WA_PRESPLIT_F16:
                # one launch: q rows + K / V written straight as the window attention's operand images
                q, images = hip.qkv_window_images(ws, ews, source.contiguous(), target.contiguous(), kv_swap, h, w, splits,
                                                  shifted)
                msg = hip.window_attention_images(q, images, h, w, splits, shifted)
            else:  # another attention arithmetic was asked for (MNERF_WA_MATH): q, k, v as tensors
                q, k, v = hip.qkv_projection(ws, ews, source.contiguous(), target.contiguous(), kv_swap)
                msg = hip.window_attention(q, k, v, h, w, splits, shifted)
        else:
            if kv_swap:
                half = target.shape[0] // 2
                target = torch.cat([target[half:], target[:half]], 0)
            if source.is_cuda and os.environ.get("MNERF_ENC_BACKWARD", "hip") != "torch":
                # training: the layer as ONE autograd node - fused HIP kernels forward (the inference kernels), HIP backward
                # (mnerf_encoder_layer_backward / mnerf_window_attention_backward / mnerf_qkv_backward): autograd.py
                from .autograd import transformer_layer
                return transformer_layer(self, source, target, h, w, splits, shifted)
            q = self.q_proj(source)
            k = self.k_proj(target)
            v = self.v_proj(target)
            msg = window_attention(q, k, v, h, w, splits, shifted)  # HIP forward; torch re-evaluation backward
        if not needs_grad:
            ws, ln, ews = self._packed_block(source.device)
            b, n, c = source.shape
            out = hip.encoder_block(msg.reshape(b * n, c), source.reshape(b * n, c).contiguous(), ws, ln, not self.no_ffn, ews)
            return out.reshape(b, n, c)
        msg = self.norm1(self.merge(msg))
        if not self.no_ffn:
            msg = self.norm2(self.mlp(torch.cat([source, msg], dim=-1)))
        return source + msg


class TransformerBlock(nn.Module):
    """transformer.py:188-247."""

    def __init__(self, d_model=128, ffn_dim_expansion=4):
        super().__init__()
        self.self_attn = TransformerLayer(d_model, no_ffn=True, ffn_dim_expansion=ffn_dim_expansion)
        self.cross_attn_ffn = TransformerLayer(d_model, no_ffn=False, ffn_dim_expansion=ffn_dim_expansion)


class FeatureTransformer(nn.Module):
    """transformer.py:250-339: both directions of every pair batched as [f0;f1] vs [f1;f0]."""

    def __init__(self, num_layers=6, d_model=128, ffn_dim_expansion=4):
        super().__init__()
        self.layers = nn.ModuleList([TransformerBlock(d_model, ffn_dim_expansion) for _ in range(num_layers)])
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def refresh_packs(self, device):
        """Re-pack the operand streams of ALL layers when any layer's parameters changed (an optimizer step bumps every one of
        them): packing.TransformerPacker — one concatenation, one gather, the fp16 split, about ten launches and ONE
        device->host copy (the kernels take each stream's power-of-two scale as an integer argument) for the whole
        transformer.  Returns the number of streams re-packed (0: nothing was stale)."""
        layers = [l for blk in self.layers for l in (blk.self_attn, blk.cross_attn_ffn)]
        device = torch.device(device)
        if not any(l.stale_packs(device) for l in layers) or any(p.device != device for l in layers for p in l._qkv_params()):
            return 0
        from . import packing
        pk = getattr(self, "_packer", None)
        if pk is None or pk.device != device or not pk.matches(self):
            pk = self._packer = packing.TransformerPacker(self, device)
        for layer, ((q_ws, q_ews), (b_ws, b_ews)) in zip(layers, pk.pack()):
            layer._qkv = (layer._pack_key(layer._qkv_params(), device), q_ws, q_ews)
            n2 = layer.norm1 if layer.no_ffn else layer.norm2
            ln = torch.stack([layer.norm1.weight, layer.norm1.bias, n2.weight, n2.bias], 0).detach().float().contiguous()
            layer._blk = (layer._pack_key(layer._block_params(), device), b_ws, ln, b_ews)
        return 2 * len(layers)

    def forward(self, src, n_pairs, h, w, splits, wo_self_attn=False):
        """src [2P, h*w, C] (first P = pair member a, last P = member b) -> same shape."""
        assert src.shape[0] == 2 * n_pairs
        if src.is_cuda:
            self.refresh_packs(src.device)
        for i, blk in enumerate(self.layers):
            shifted = (i % 2 == 1) and splits > 1
            blk_in = src  # the cross attention's keys / values come from the block INPUT of the other pair member:
            if not wo_self_attn:  # its batch halves exchanged (transformer.py:317-335) - an index, not a torch.cat
                src = blk.self_attn(src, src, h, w, splits, shifted)
            src = blk.cross_attn_ffn(src, blk_in, h, w, splits, shifted, kv_swap=True)
        return src


class UpSampler(nn.Module):
    """superres.py:5-38."""

    def __init__(self, n_feat=128, upsample_factor=2):
        super().__init__()
        self.n_blocks = int(math.log2(upsample_factor))
        self.conv_ls = nn.ModuleList([nn.Conv2d(n_feat, n_feat, 3, 1, 1) for _ in range(self.n_blocks)])
        self.conv_l2rs = nn.ModuleList([nn.Conv2d(n_feat, n_feat, 3, 1, 1) for _ in range(self.n_blocks + 1)])

    def forward_tokens(self, x_cl, pair_major=False):
        """Inference on channel-last tokens [N,h,w,C] (what the transformer leaves): the convolutions read them as they
        are, the nearest up-sampling in front of ``conv_ls`` is index arithmetic inside the convolution, LeakyReLU and the
        bilinear up-sampling + add of the other branch are epilogues (csrc/conv.hip).  Returns NCHW [N,C,2^b h,2^b w] like
        ``forward``, or with ``pair_major`` the layout the cost volume reads, [N/2, 2, H, W, C] (first / second batch half
        = a / b member of every pair), written by the last convolution itself."""
        scal = hip.absmax_regions(self.n_blocks + 1, x_cl.device)
        hip.absmax(x_cl, scal[0])
        right = _hip_conv(self.conv_l2rs[0], x_cl, scal[0], channels_last=True)
        left, left_cl = x_cl, True
        for i in range(self.n_blocks):
            left = _hip_conv(self.conv_ls[i], left, scal[i], channels_last=left_cl, upsample2x=True, leaky=0.2,
                             out_absmax=scal[i + 1])
            left_cl = False
            # right = up_bilinear(right) + conv(left): the up-sampling and the add are the convolution's epilogue
            last = pair_major and i == self.n_blocks - 1
            right = _hip_conv(self.conv_l2rs[i + 1], left, scal[i + 1], add_bilinear2x=right,
                              out_layout=hip.CONV_OUT_PAIR_MAJOR if last else hip.CONV_OUT_NCHW)
        return right

    def forward(self, x):
        if _train_hip(x):
            from . import autograd as AG
            _train_packs(self, list(self.conv_ls) + list(self.conv_l2rs), x.device)
            conv = AG.conv2d
        else:
            conv = lambda m, t: m(t)  # noqa: E731
        right = conv(self.conv_l2rs[0], x)
        left = x
        for i in range(self.n_blocks):
            left = F.leaky_relu(conv(self.conv_ls[i], F.interpolate(left, scale_factor=2.0, mode="nearest")), 0.2)
            if conv is not None and _train_hip(x):
                right = AG.upsample_bilinear2x_add(right, conv(self.conv_l2rs[i + 1], left))
            else:
                right = F.interpolate(right, scale_factor=2, mode="bilinear", align_corners=False) + conv(self.conv_l2rs[i + 1], left)
        return right


_SINE_CACHE = {}
_PAIR_INDEX_CACHE = {}


def _pair_index_tensors(v, device):
    """view indices (a | b) of the V (V - 1) / 2 pairs as device tensors, built once per (V, device): a host list -> device
    copy per forward is a pageable-memory transfer, which a stream capture (the encoder graph, matchnerf.py) does not allow"""
    key = (int(v), str(device))
    if key not in _PAIR_INDEX_CACHE:
        pairs = pair_list(v)
        _PAIR_INDEX_CACHE[key] = (torch.tensor([a for a, _ in pairs], device=device),
                                  torch.tensor([bb for _, bb in pairs], device=device))
    return _PAIR_INDEX_CACHE[key]


def sine_position_tokens(h, w, channels, device):
    """DETR sine embedding of an (h,w) window (position.py:26-47) as tokens [h*w, C]."""
    key = (h, w, channels, str(device))
    if key not in _SINE_CACHE:
        npf = channels // 2
        y = (torch.arange(1, h + 1, dtype=torch.float32) / (h + 1e-6) * (2 * math.pi))[:, None].expand(h, w)
        x = (torch.arange(1, w + 1, dtype=torch.float32) / (w + 1e-6) * (2 * math.pi))[None, :].expand(h, w)
        i = torch.arange(npf, dtype=torch.float32)
        dim_t = 10000.0 ** (2 * torch.div(i, 2, rounding_mode="trunc") / npf)

        def emb(t):
            a = t[..., None] / dim_t
            return torch.stack([a[..., 0::2].sin(), a[..., 1::2].cos()], -1).flatten(-2)

        _SINE_CACHE[key] = torch.cat([emb(y), emb(x)], -1).to(device)  # [h,w,C]
    return _SINE_CACHE[key]


class GMFlow(nn.Module):
    """Constructor keeps the reference's keyword surface (gmflow.py:12-45); children are named
    ``backbone`` / ``transformer`` / ``featup_net`` as ``restore_checkpoint`` and
    ``load_gmflow_checkpoint`` expect (misc/utils.py:160-205)."""

    def __init__(self, num_scales=1, upsample_factor=2, feature_channels=128, attention_type="swin",
                 num_transformer_layers=6, ffn_dim_expansion=4, num_head=1, feature_upsampler="network",
                 device=None, **kwargs):
        super().__init__()
        if num_scales != 1 or num_head != 1 or attention_type != "swin" or feature_channels != 128:
            raise NotImplementedError("MatchNeRF instantiates GMFlow(num_scales=1, num_head=1, 'swin', 128 ch) "
                                      "(models/matchnerf.py:21-26); other shapes are not built")
        if feature_upsampler != "network":
            raise NotImplementedError("keep_raw_feats requires the network up-sampler (gmflow.py:117)")
        self.feature_channels = feature_channels
        self.upsample_factor = upsample_factor
        self.backbone = CNNEncoder(output_dim=feature_channels)
        self.transformer = FeatureTransformer(num_transformer_layers, feature_channels, ffn_dim_expansion)
        self.featup_net = UpSampler(feature_channels, upsample_factor)
        self.register_buffer("_mean", torch.tensor([0.485, 0.456, 0.406]).reshape(1, 3, 1, 1), persistent=False)
        self.register_buffer("_std", torch.tensor([0.229, 0.224, 0.225]).reshape(1, 3, 1, 1), persistent=False)

    def forward(self, imgs, attn_splits_list=None, wo_self_attn=False, **kwargs):
        """imgs [B,V,3,H,W] in [0,1] -> list over scales (1/8 raw, 1/4 up-sampled) of
        pair-major channel-last maps [B, P, 2, h_s, w_s, 128] (gmflow.py:91-150)."""
        b, v, c, hh, ww = imgs.shape
        splits = attn_splits_list[0] if isinstance(attn_splits_list, (list, tuple)) else (attn_splits_list or 1)
        x = imgs.reshape(b * v, c, hh, ww)
        if hh == 756 and ww == 1008:  # IBRNet setting, gmflow.py:100-103
            x = F.interpolate(x, size=(768, 1024), mode="bilinear", align_corners=True)
        def _down8(n):  # three stride-2 convolutions (7x7 pad 3, 3x3 pad 1, 3x3 pad 1): ceil-like sizes
            for _ in range(3):
                n = (n - 1) // 2 + 1
            return n
        h, w, ch = _down8(x.shape[2]), _down8(x.shape[3]), self.feature_channels
        if h % splits or w % splits:
            raise ValueError(f"feature map {h}x{w} is not divisible by attn_splits={splits}")
        pe = sine_position_tokens(h // splits, w // splits, ch, x.device).repeat(splits, splits, 1).reshape(h * w, ch).contiguous()
        # [BV,h,w,C] tokens with the window position tile added (by the backbone's last convolution at inference)
        tok = self.backbone((x - self._mean) / self._std, tokens_plus=pe)
        if tuple(tok.shape[1:]) != (h, w, ch):
            raise ValueError(f"backbone output {tuple(tok.shape[1:])} != expected {(h, w, ch)} for a {hh}x{ww} input")
        tok = tok.reshape(b, v, h * w, ch)
        pairs = pair_list(v)
        ia, ib = _pair_index_tensors(v, tok.device)
        p_n = len(pairs)
        outs0, outs1 = [], []
        for bi in range(b):
            src = torch.cat([tok[bi, ia], tok[bi, ib]], 0).contiguous()       # [2P, hw, C]
            src = self.transformer(src, p_n, h, w, splits, wo_self_attn)
            outs0.append(torch.stack([src[:p_n], src[p_n:]], 1).reshape(p_n, 2, h, w, ch))
            if _fused_norm(src) and self.featup_net.n_blocks >= 1:
                outs1.append(self.featup_net.forward_tokens(src.reshape(2 * p_n, h, w, ch), pair_major=True))
            else:
                up = self.featup_net(src.reshape(2 * p_n, h, w, ch).permute(0, 3, 1, 2))  # [2P,C,2h,2w]
                up = up.permute(0, 2, 3, 1)
                outs1.append(torch.stack([up[:p_n], up[p_n:]], 1).contiguous())
        if b == 1:  # the usual case: no batch copy
            return [outs0[0].unsqueeze(0).contiguous(), outs1[0].unsqueeze(0).contiguous()]
        return [torch.stack(outs0, 0).contiguous(), torch.stack(outs1, 0).contiguous()]


def pair_major_to_view_chunks(feat_pm):
    """[B,P,2,h,w,C] -> the reference's per-view layout [B,V,(V-1)*C,h,w] (matchnerf.py:192-205);
    for callers that want ``get_img_feat``'s original return format."""
    b, p_n = feat_pm.shape[:2]
    v = int(round((1 + math.sqrt(1 + 8 * p_n)) / 2))
    per_view = [[] for _ in range(v)]
    for p, (a, bb) in enumerate(pair_list(v)):
        per_view[a].append(feat_pm[:, p, 0])
        per_view[bb].append(feat_pm[:, p, 1])
    return torch.stack([torch.cat(chunks, -1) for chunks in per_view], 1).permute(0, 1, 4, 2, 3)
