"""ctypes binding of libmnerf_hip.so (include/mnerf.h) — the only door to the HIP kernels.

There is deliberately NO fallback: if the shared library is missing or a call fails, a
``MnerfError`` is raised.  The product path never routes through the CPU oracle.
Tensors are passed as raw device pointers (``tensor.data_ptr()``) plus explicit sizes.  Every
launch runs under ``torch.cuda.device(<device of the tensors>)`` on that device's current torch
stream (the library launches on the calling thread's current HIP device), so the module works on
any ``--gpu_ids`` / LOCAL_RANK, like the pure-torch reference.  The binding itself needs no GPU
(``load()`` works on a CPU-only box; that is what the ``-m "not gpu"`` ABI tests exercise).
"""
import contextlib
import ctypes as C
import os

import numpy as np

MNERF_ABI_VERSION = 9
MNERF_POSE_FLOATS = 24  # floats of one row of mnerf_rays.pose_table
MNERF_OK, MNERF_E_NULL, MNERF_E_RANGE, MNERF_E_UNSUPPORTED, MNERF_E_ALIGN = 0, -1, -2, -3, -4  # include/mnerf.h
MNERF_MAX_VIEWS = 16
MNERF_COND_STRIDE_MAX, MNERF_COND_STRIDE_MAX_F32 = 96, 64
SMALL_FIXED = 32  # floats of the `small` parameter block (LayerNorm weight|bias) before the ray-posenc table

_LIB = None
_LIB_PATH = os.environ.get("MNERF_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmnerf_hip.so")

EXPORTS = ("mnerf_abi_version", "mnerf_last_error", "mnerf_struct_size", "mnerf_ray_samples", "mnerf_composite", "mnerf_cost_volume",
           "mnerf_cost_volume_operand_bytes", "mnerf_cost_volume_operands",
           "mnerf_composite_backward", "mnerf_cost_volume_backward", "mnerf_decoder_backward", "mnerf_decoder_backward_workspace_bytes", "mnerf_debug_set_knob",
           "mnerf_decoder_wstream_floats", "mnerf_decoder_chunk", "mnerf_decoder_samples", "mnerf_render_workspace_bytes",
           "mnerf_render_chunk", "mnerf_render_chunk_fused", "mnerf_render_chunk_is_fused", "mnerf_render_takes_pose_table", "mnerf_window_attention",
           "mnerf_window_attention_presplit", "mnerf_window_attention_workspace_bytes", "mnerf_window_attention_backward", "mnerf_window_attention_backward_workspace_bytes", "mnerf_qkv_projection", "mnerf_qkv_wstream_floats", "mnerf_qkv_window_images", "mnerf_window_attention_images", "mnerf_instance_norm", "mnerf_instance_norm_backward", "mnerf_upsample_bilinear2x", "mnerf_upsample_bilinear2x_backward", "mnerf_conv2d", "mnerf_conv_wstream_floats", "mnerf_conv_stem", "mnerf_conv_stem_wstream_floats", "mnerf_absmax", "mnerf_conv2d_backward_data", "mnerf_conv2d_backward_weight", "mnerf_conv2d_backward_weight_workspace_bytes", "mnerf_conv2d_backward_weight_f16x3", "mnerf_conv2d_forward_f32", "mnerf_conv_stem_backward_weight", "mnerf_conv_stem_backward_weight_workspace_bytes", "mnerf_encoder_block", "mnerf_encoder_block_wstream_floats",
           "mnerf_encoder_layer_backward", "mnerf_encoder_layer_backward_workspace_bytes", "mnerf_qkv_backward", "mnerf_debug_gemm",
           "mnerf_window_attention_presplit_stats", "mnerf_window_attention_backward_stats", "mnerf_encoder_block_save", "mnerf_encoder_layer_backward_saved")


class MnerfError(RuntimeError):
    pass


class View(C.Structure):
    _fields_ = [("extr", C.c_float * 12), ("intr", C.c_float * 9), ("near_", C.c_float), ("far_", C.c_float)]


class Rays(C.Structure):
    _fields_ = [("n_rays", C.c_int32), ("n_samples", C.c_int32), ("ray_begin", C.c_int32),
                ("legacy_coord", C.c_int32), ("depth_inverse", C.c_int32), ("height", C.c_int32),
                ("width", C.c_int32), ("ray_idx", C.c_void_p), ("strat_u", C.c_void_p),
                ("kinv", C.c_float * 9), ("c2w", C.c_float * 12), ("near_", C.c_float), ("far_", C.c_float),
                ("pose_table", C.c_void_p), ("rays_per_pose", C.c_int32), ("pad_", C.c_int32)]


class Scene(C.Structure):
    _fields_ = [("n_views", C.c_int32), ("n_scales", C.c_int32), ("fh", C.c_int32 * 2), ("fw", C.c_int32 * 2),
                ("n_group", C.c_int32 * 2), ("feat", C.c_void_p * 2), ("images", C.c_void_p),
                ("views", View * MNERF_MAX_VIEWS), ("feat_op", C.c_void_p)]


class Decoder(C.Structure):
    _fields_ = [("wstream", C.c_void_p), ("wstream_floats", C.c_int64), ("small_", C.c_void_p),
                ("n_views", C.c_int32), ("cond_dim", C.c_int32), ("cond_stride", C.c_int32),
                ("L_3D", C.c_int32), ("raytrans_posenc", C.c_int32), ("raytrans_elu", C.c_int32),
                ("density_maskfill", C.c_int32), ("wo_render_interval", C.c_int32),
                ("setbg_opaque", C.c_int32), ("wstream_format", C.c_int32)]


# parameter tensors of the decoder in mnerf_decoder_train's order (include/mnerf.h, MNERF_DT_*): names relative to the CondNeRF module
DEC_TRAIN_TENSORS = tuple(f"pts_linears.{i}.{k}" for i in range(6) for k in ("weight", "bias")) + (
    "pts_bias.weight", "pts_bias.bias", "alpha_linear.0.weight", "alpha_linear.0.bias", "ray_attention.w_qs.weight",
    "ray_attention.w_ks.weight", "ray_attention.w_vs.weight", "ray_attention.fc.weight", "ray_attention.layer_norm.weight",
    "ray_attention.layer_norm.bias", "out_alpha_linear.0.weight", "out_alpha_linear.0.bias", "out_alpha_linear.2.weight",
    "out_alpha_linear.2.bias", "feature_linear.weight", "feature_linear.bias", "views_linears.0.weight", "views_linears.0.bias",
    "rgb_linear.weight", "rgb_linear.bias")


class DecoderTrain(C.Structure):
    _fields_ = [("n_views", C.c_int32), ("cond_dim", C.c_int32), ("n_trunk", C.c_int32), ("net_width", C.c_int32),
                ("skip_layer", C.c_int32), ("L_3D", C.c_int32), ("legacy_coord", C.c_int32), ("raytrans_elu", C.c_int32),
                ("raytrans_posenc", C.c_int32), ("density_maskfill", C.c_int32), ("raytrans_table", C.c_void_p),
                ("w", C.c_void_p * len(DEC_TRAIN_TENSORS)), ("g", C.c_void_p * len(DEC_TRAIN_TENSORS))]


class EncoderLayer(C.Structure):
    _fields_ = [("wstream", C.c_void_p), ("wstream_floats", C.c_int64), ("ln", C.c_void_p), ("ffn", C.c_int32),
                ("ew_merge", C.c_int32), ("ew_w1", C.c_int32), ("ew_w2", C.c_int32)]


class EncoderLayerTrain(C.Structure):
    """mnerf_encoder_layer_train: parameters of a transformer layer after the attention in torch's layouts + their gradients"""
    _fields_ = [("ffn", C.c_int32), ("pad_", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("w_merge", "ln1_w", "ln1_b", "w_mlp0", "w_mlp2", "ln2_w", "ln2_b",
                                          "g_w_merge", "g_ln1_w", "g_ln1_b", "g_w_mlp0", "g_w_mlp2", "g_ln2_w", "g_ln2_b")]


WSTREAM_F32, WSTREAM_BF16X3, WSTREAM_F16X2, WSTREAM_F16X1 = 0, 1, 2, 3
WA_SPLIT_BF16, WA_EXACT_F32, WA_SPLIT_F16 = 0, 1, 2
WA_PRESPLIT_F16 = 3  # host-side selector only: routed to mnerf_window_attention_presplit
ABSMAX_FLOATS = 64 * 32  # floats of one absmax region (MNERF_ABSMAX_FLOATS)
CONV_OUT_NCHW, CONV_OUT_CHANNEL_LAST, CONV_OUT_PAIR_MAJOR = 0, 1, 2


class ConvLayer(C.Structure):
    """struct mnerf_conv (include/mnerf.h)"""
    _fields_ = [("wstream", C.c_void_p), ("wstream_floats", C.c_int64), ("bias", C.c_void_p), ("c_in", C.c_int32),
                ("c_out", C.c_int32), ("ksize", C.c_int32), ("stride", C.c_int32), ("ew", C.c_int32),
                ("leaky_slope", C.c_float)]


def lib_path():
    return _LIB_PATH


def load():
    """dlopen the library (once) and declare signatures.  Raises MnerfError if absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # PyTorch-ROCm bundles its own libamdhip64.so.7; it must be resident BEFORE this library is
    # dlopen'ed so that both bind to ONE HIP runtime (device pointers and streams are shared).
    # Loading /opt/rocm's copy first leaves the process with a runtime torch cannot use.
    import torch  # noqa: F401
    if not os.path.exists(_LIB_PATH):
        raise MnerfError(
            f"{_LIB_PATH} not found: build it with `python -m matchnerf_amd.csrc.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the render path.")
    try:
        lib = C.CDLL(_LIB_PATH)
    except OSError as e:  # noqa: PERF203
        raise MnerfError(f"cannot load {_LIB_PATH}: {e}") from e
    vp, i32, i64, fp = C.c_void_p, C.c_int32, C.c_int64, C.c_void_p
    lib.mnerf_abi_version.restype = C.c_int
    lib.mnerf_abi_version.argtypes = []
    lib.mnerf_last_error.restype = C.c_char_p
    lib.mnerf_last_error.argtypes = []
    lib.mnerf_struct_size.restype = i64
    lib.mnerf_struct_size.argtypes = [i32]
    lib.mnerf_ray_samples.restype = C.c_int
    lib.mnerf_ray_samples.argtypes = [C.POINTER(Rays), C.POINTER(View), fp, fp, fp, vp]
    lib.mnerf_composite.restype = C.c_int
    lib.mnerf_composite.argtypes = [i32, i32, fp, fp, fp, fp, i32, i32, fp, fp, fp, fp, vp]
    lib.mnerf_composite_backward.restype = C.c_int
    lib.mnerf_composite_backward.argtypes = [i32, i32, fp, fp, fp, fp, i32, i32, fp, fp, fp, fp, fp, vp]
    lib.mnerf_cost_volume_backward.restype = C.c_int
    lib.mnerf_cost_volume_backward.argtypes = [C.POINTER(Scene), C.POINTER(Rays), i32, fp, fp, fp, vp]
    lib.mnerf_cost_volume.restype = C.c_int
    lib.mnerf_cost_volume.argtypes = [C.POINTER(Scene), C.POINTER(Rays), i32, fp, vp]
    lib.mnerf_cost_volume_operand_bytes.restype = i64
    lib.mnerf_cost_volume_operand_bytes.argtypes = [C.POINTER(Scene)]
    lib.mnerf_cost_volume_operands.restype = C.c_int
    lib.mnerf_cost_volume_operands.argtypes = [C.POINTER(Scene), vp, vp]
    lib.mnerf_debug_set_knob.restype = C.c_int
    lib.mnerf_debug_set_knob.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int)]
    lib.mnerf_decoder_backward_workspace_bytes.restype = i64
    lib.mnerf_decoder_backward_workspace_bytes.argtypes = [i32, i32]
    lib.mnerf_decoder_backward.restype = C.c_int
    lib.mnerf_decoder_backward.argtypes = [C.POINTER(DecoderTrain), i32, i32, fp, fp, fp, i32, fp, fp, fp, vp, vp]
    lib.mnerf_decoder_wstream_floats.restype = i64
    lib.mnerf_decoder_wstream_floats.argtypes = [i32, i32, i32, i32]
    lib.mnerf_decoder_chunk.restype = C.c_int
    lib.mnerf_decoder_chunk.argtypes = [C.POINTER(Decoder), C.POINTER(View), C.POINTER(Rays), fp, fp, fp, fp, fp, fp, vp]
    lib.mnerf_decoder_samples.restype = C.c_int
    lib.mnerf_decoder_samples.argtypes = [C.POINTER(Decoder), i32, i32, i32, fp, fp, fp, fp, fp, vp]
    lib.mnerf_render_workspace_bytes.restype = i64
    lib.mnerf_render_workspace_bytes.argtypes = [i32, i32, i32]
    lib.mnerf_render_chunk_is_fused.restype = i32
    lib.mnerf_render_chunk_is_fused.argtypes = [C.POINTER(Scene), C.POINTER(Decoder), C.POINTER(Rays)]
    lib.mnerf_render_takes_pose_table.restype = i32
    lib.mnerf_render_takes_pose_table.argtypes = [C.POINTER(Scene), C.POINTER(Decoder), i32, i32]
    lib.mnerf_render_chunk_fused.restype = C.c_int
    lib.mnerf_render_chunk_fused.argtypes = [C.POINTER(Scene), C.POINTER(Decoder), C.POINTER(Rays), fp, fp, fp, vp]
    lib.mnerf_render_chunk.restype = C.c_int
    lib.mnerf_render_chunk.argtypes = [C.POINTER(Scene), C.POINTER(Decoder), C.POINTER(Rays), vp, fp, fp, fp, vp]
    lib.mnerf_window_attention.restype = C.c_int
    lib.mnerf_window_attention.argtypes = [fp, fp, fp, fp, i32, i32, i32, i32, i32, i32, vp]
    lib.mnerf_window_attention_workspace_bytes.restype = C.c_size_t
    lib.mnerf_window_attention_workspace_bytes.argtypes = [i32, i32, i32, i32]
    lib.mnerf_window_attention_presplit.restype = C.c_int
    lib.mnerf_window_attention_presplit.argtypes = [fp, fp, fp, fp, i32, i32, i32, i32, i32, vp, C.c_size_t, vp]
    lib.mnerf_window_attention_backward_workspace_bytes.restype = i64
    lib.mnerf_window_attention_backward_workspace_bytes.argtypes = [i32, i32, i32]
    lib.mnerf_window_attention_backward.restype = C.c_int
    lib.mnerf_window_attention_backward.argtypes = [fp, fp, fp, fp, fp, fp, fp, fp, i32, i32, i32, i32, i32, vp, C.c_size_t, vp]
    lib.mnerf_qkv_wstream_floats.restype = i64
    lib.mnerf_qkv_wstream_floats.argtypes = []
    lib.mnerf_qkv_projection.restype = C.c_int
    lib.mnerf_qkv_projection.argtypes = [fp, C.POINTER(C.c_int32), fp, fp, i32, fp, fp, fp, i32, i32, vp]
    lib.mnerf_qkv_window_images.restype = C.c_int
    lib.mnerf_qkv_window_images.argtypes = [fp, C.POINTER(C.c_int32), fp, fp, i32, fp, vp, C.c_size_t, i32, i32, i32, i32, i32, vp]
    lib.mnerf_window_attention_images.restype = C.c_int
    lib.mnerf_window_attention_images.argtypes = [fp, fp, i32, i32, i32, i32, i32, vp, C.c_size_t, vp]
    lib.mnerf_instance_norm.restype = C.c_int
    lib.mnerf_instance_norm.argtypes = [fp, fp, fp, i64, i64, C.c_float, i32, i32, fp, vp]
    lib.mnerf_upsample_bilinear2x.restype = C.c_int
    lib.mnerf_upsample_bilinear2x.argtypes = [fp, fp, fp, i64, i32, i32, vp]
    lib.mnerf_upsample_bilinear2x_backward.restype = C.c_int
    lib.mnerf_upsample_bilinear2x_backward.argtypes = [fp, fp, i64, i32, i32, vp]
    lib.mnerf_instance_norm_backward.restype = C.c_int
    lib.mnerf_instance_norm_backward.argtypes = [fp, fp, fp, i64, i64, C.c_float, i32, vp]
    lib.mnerf_conv_wstream_floats.restype = i64
    lib.mnerf_conv_wstream_floats.argtypes = [i32, i32, i32]
    lib.mnerf_conv2d.restype = C.c_int
    lib.mnerf_conv2d.argtypes = [C.POINTER(ConvLayer), fp, i32, i32, fp, fp, fp, fp, i32, fp, i32, i32, i32, vp]
    lib.mnerf_conv2d_backward_data.restype = C.c_int
    lib.mnerf_conv2d_backward_data.argtypes = [fp, fp, fp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.mnerf_conv2d_backward_weight_workspace_bytes.restype = C.c_size_t
    lib.mnerf_conv2d_backward_weight_workspace_bytes.argtypes = [i32, i32, i32, i32, i32, i32, i32]
    lib.mnerf_conv2d_backward_weight.restype = C.c_int
    lib.mnerf_conv2d_backward_weight.argtypes = [fp, fp, fp, vp, C.c_size_t, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.mnerf_conv2d_backward_weight_f16x3.restype = C.c_int
    lib.mnerf_conv2d_backward_weight_f16x3.argtypes = [fp, fp, fp, fp, fp, vp, C.c_size_t, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.mnerf_conv2d_forward_f32.restype = C.c_int
    lib.mnerf_conv2d_forward_f32.argtypes = [fp, fp, fp, fp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.mnerf_conv_stem_backward_weight_workspace_bytes.restype = C.c_size_t
    lib.mnerf_conv_stem_backward_weight_workspace_bytes.argtypes = [i32, i32, i32]
    lib.mnerf_conv_stem_backward_weight.restype = C.c_int
    lib.mnerf_conv_stem_backward_weight.argtypes = [fp, fp, fp, vp, C.c_size_t, i32, i32, i32, vp]
    lib.mnerf_conv_stem_wstream_floats.restype = i64
    lib.mnerf_conv_stem_wstream_floats.argtypes = []
    lib.mnerf_conv_stem.restype = C.c_int
    lib.mnerf_conv_stem.argtypes = [fp, i32, fp, fp, fp, i32, i32, i32, vp]
    lib.mnerf_absmax.restype = C.c_int
    lib.mnerf_absmax.argtypes = [fp, i64, fp, vp]
    lib.mnerf_encoder_block_wstream_floats.restype = i64
    lib.mnerf_encoder_block_wstream_floats.argtypes = [i32]
    lib.mnerf_encoder_block.restype = C.c_int
    lib.mnerf_encoder_block.argtypes = [C.POINTER(EncoderLayer), fp, fp, fp, i32, vp]
    lib.mnerf_encoder_layer_backward_workspace_bytes.restype = i64
    lib.mnerf_encoder_layer_backward_workspace_bytes.argtypes = [i32]
    lib.mnerf_encoder_layer_backward.restype = C.c_int
    lib.mnerf_encoder_layer_backward.argtypes = [C.POINTER(EncoderLayerTrain), fp, fp, fp, fp, fp, i32, vp, vp]
    lib.mnerf_qkv_backward.restype = C.c_int
    lib.mnerf_qkv_backward.argtypes = [fp] * 13 + [i32, vp]
    ver = lib.mnerf_abi_version()
    if ver != MNERF_ABI_VERSION:
        raise MnerfError(f"libmnerf_hip.so ABI {ver} != binding ABI {MNERF_ABI_VERSION}")
    for which, st in enumerate((View, Rays, Scene, Decoder, EncoderLayer, ConvLayer, DecoderTrain, EncoderLayerTrain)):
        if lib.mnerf_struct_size(which) != C.sizeof(st):
            raise MnerfError(f"struct {st.__name__}: library says {lib.mnerf_struct_size(which)} bytes, "
                             f"ctypes mirror has {C.sizeof(st)}")
    _LIB = lib
    return lib


@contextlib.contextmanager
def knob(name, value):
    """Tests / diagnosis: run the block with one tuning knob of the library changed (mnerf_debug_set_knob)."""
    lib = load()
    old = C.c_int(0)
    check(lib.mnerf_debug_set_knob(name.encode(), int(value), C.byref(old)), "mnerf_debug_set_knob")
    try:
        yield
    finally:
        lib.mnerf_debug_set_knob(name.encode(), old.value, None)


def check(rc, what):
    if rc != 0:
        msg = load().mnerf_last_error().decode(errors="replace")
        raise MnerfError(f"{what} failed (rc={rc}): {msg}")


# ----------------------------------------------------------------------- struct builders


def _fill(arr, values):
    flat = np.asarray(values, dtype=np.float32).reshape(-1)
    assert flat.size == len(arr), (flat.size, len(arr))
    for i, x in enumerate(flat):
        arr[i] = float(x)


def make_view(extr34, intr33, near, far):
    v = View()
    _fill(v.extr, extr34)
    _fill(v.intr, intr33)
    v.near_, v.far_ = float(near), float(far)
    return v


def make_rays(n_rays, n_samples, height, width, kinv, c2w, near, far, ray_begin=0, legacy=True,
              depth_inverse=False, ray_idx_ptr=None, strat_u_ptr=None, pose_table_ptr=None, rays_per_pose=0):
    """``pose_table_ptr`` / ``rays_per_pose``: several target poses in one launch (include/mnerf.h: POSE TABLE; rows from
    ``pose_table_rows``); kinv / c2w / near / far are then ignored (pass those of any pose)."""
    r = Rays()
    r.pose_table = pose_table_ptr
    r.rays_per_pose = int(rays_per_pose) if pose_table_ptr else 0
    r.n_rays, r.n_samples, r.ray_begin = int(n_rays), int(n_samples), int(ray_begin)
    r.legacy_coord, r.depth_inverse = int(bool(legacy)), int(bool(depth_inverse))
    r.height, r.width = int(height), int(width)
    r.ray_idx = ray_idx_ptr
    r.strat_u = strat_u_ptr
    _fill(r.kinv, kinv)
    _fill(r.c2w, c2w)
    r.near_, r.far_ = float(near), float(far)
    return r


def pose_table_rows(poses):
    """[(kinv [3,3], c2w [3,4], near, far), ...] -> float32 [n, MNERF_POSE_FLOATS] rows of mnerf_rays.pose_table"""
    rows = np.zeros((len(poses), MNERF_POSE_FLOATS), np.float32)
    for i, (kinv, c2w, near, far) in enumerate(poses):
        rows[i, :9] = np.asarray(kinv, np.float32).reshape(-1)
        rows[i, 9:21] = np.asarray(c2w, np.float32).reshape(-1)
        rows[i, 21], rows[i, 22] = near, far
    return rows


@contextlib.contextmanager
def _on(device, stream=None):
    """Make ``device`` the current HIP device for the launch and yield the stream handle to launch on
    (``stream`` or that device's current torch stream)."""
    import torch
    device = torch.device(device)
    if device.type != "cuda":
        raise MnerfError(f"the HIP kernels need CUDA tensors, got device {device} (there is no CPU fallback)")
    with torch.cuda.device(device):
        st = stream if stream is not None else torch.cuda.current_stream(device)
        yield C.c_void_p(st.cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _f32c(t, name):
    import torch
    if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
        raise MnerfError(f"{name}: expected a contiguous float32 CUDA tensor, got {t.dtype} "
                         f"contig={t.is_contiguous()} device={t.device}")
    return t


def _current_device():
    import torch
    if not torch.cuda.is_available():
        raise MnerfError("the HIP kernels need a GPU (there is no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


# ----------------------------------------------------------------------- op wrappers


def ray_samples(rays, view, device=None, stream=None):
    """a8-a10 (camera.py:255-286, 351-379) -> pts [R,S,3], ndc [R,S,3], depth [R,S]."""
    import torch
    lib = load()
    device = torch.device(device) if device is not None else _current_device()
    r, s = rays.n_rays, rays.n_samples
    pts = torch.empty(r, s, 3, device=device)
    ndc = torch.empty(r, s, 3, device=device)
    depth = torch.empty(r, s, device=device)
    with _on(device, stream) as st:
        check(lib.mnerf_ray_samples(C.byref(rays), C.byref(view), _ptr(pts), _ptr(ndc), _ptr(depth), st),
              "mnerf_ray_samples")
    return pts, ndc, depth


def composite(rgb_s, sigma, depth_s, ray_len=None, wo_render_interval=True, setbg_opaque=False, want_prob=False,
              stream=None):
    """K5 (nerf.py:101-124). rgb_s [R,S,3], sigma [R,S], depth_s [R,S] -> rgb [R,3], depth [R], opacity [R]
    (+ prob [R,S] with ``want_prob``)."""
    import torch
    lib = load()
    r, s = sigma.shape
    _f32c(rgb_s, "rgb_s"), _f32c(sigma, "sigma"), _f32c(depth_s, "depth_s")
    dev = sigma.device
    rgb = torch.empty(r, 3, device=dev)
    depth = torch.empty(r, device=dev)
    opacity = torch.empty(r, device=dev)
    prob = torch.empty(r, s, device=dev) if want_prob else None
    with _on(dev, stream) as st:
        check(lib.mnerf_composite(r, s, _ptr(rgb_s), _ptr(sigma), _ptr(depth_s), _ptr(ray_len),
                                  int(wo_render_interval), int(setbg_opaque), _ptr(rgb), _ptr(depth),
                                  _ptr(opacity), _ptr(prob), st), "mnerf_composite")
    if want_prob:
        return rgb, depth, opacity, prob
    return rgb, depth, opacity


def composite_backward(rgb_s, sigma, depth_s, g_rgb, g_depth=None, g_opacity=None, ray_len=None,
                       wo_render_interval=True, setbg_opaque=False, stream=None):
    """K5 backward: gradients of (rgb [R,3], depth [R], opacity [R]) -> (g_rgb_s [R,S,3], g_sigma [R,S])."""
    import torch
    lib = load()
    r, s = sigma.shape
    for t, n in ((rgb_s, "rgb_s"), (sigma, "sigma"), (depth_s, "depth_s"), (g_rgb, "g_rgb")):
        _f32c(t, n)
    g_rgb_s = torch.empty(r, s, 3, device=sigma.device)
    g_sigma = torch.empty(r, s, device=sigma.device)
    with _on(sigma.device, stream) as st:
        check(lib.mnerf_composite_backward(r, s, _ptr(rgb_s), _ptr(sigma), _ptr(depth_s), _ptr(ray_len),
                                           int(wo_render_interval), int(setbg_opaque), _ptr(g_rgb), _ptr(g_depth),
                                           _ptr(g_opacity), _ptr(g_rgb_s), _ptr(g_sigma), st), "mnerf_composite_backward")
    return g_rgb_s, g_sigma


def cost_volume_backward(scene, rays, cond_stride, g_cond, g_feats, stream=None):
    """K1+K2 backward: scatter-adds the gradient of the conditioning rows [n_rays*S, cond_stride] into ``g_feats``
    (list of 1 or 2 zero-initialised tensors shaped like the scene's feature maps)."""
    lib = load()
    _f32c(g_cond, "g_cond")
    for g in g_feats:
        _f32c(g, "g_feat")
    with _on(g_cond.device, stream) as st:
        check(lib.mnerf_cost_volume_backward(C.byref(scene), C.byref(rays), int(cond_stride), _ptr(g_cond),
                                             _ptr(g_feats[0]), _ptr(g_feats[1]) if len(g_feats) > 1 else None, st),
              "mnerf_cost_volume_backward")
    return g_feats


def decoder_backward(opt, params, n_views, x_ndc, dirs, cond, cond_stride, g_rgb_s, g_sigma, want_g_cond=True, raytrans_table=None,
                     grads=None, stream=None):
    """K3+K4 backward (cond_nerf.py:52-100 and ray_transformer.py:29-79 under autograd): ``params`` maps DEC_TRAIN_TENSORS names to
    fp32 CUDA tensors in torch's layouts; returns (g_cond [N, cond_stride] or None, {name: gradient}).  ``grads`` may hold
    tensors to accumulate into (missing names get fresh zero tensors; a name mapped to None is skipped)."""
    import torch
    lib = load()
    r, s = g_sigma.shape
    n = r * s
    dev = g_sigma.device
    for t, nm in ((x_ndc, "x_ndc"), (dirs, "dirs"), (cond, "cond"), (g_rgb_s, "g_rgb_s"), (g_sigma, "g_sigma")):
        _f32c(t, nm)
    d = DecoderTrain()
    dc = sum(opt.encoder.cos_n_group) + 4 * n_views
    skip = list(opt.decoder.skip)
    if len(skip) > 1:
        raise NotImplementedError("decoder_backward: one skip connection (opt.decoder.skip has %d)" % len(skip))
    d.n_views, d.cond_dim, d.n_trunk, d.net_width = n_views, dc, int(opt.decoder.net_depth), int(opt.decoder.net_width)
    d.skip_layer = int(skip[0]) if skip else -1
    d.L_3D = int(opt.decoder.posenc.L_3D) if opt.decoder.posenc else 0
    d.legacy_coord = int(bool(opt.nerf.legacy_coord))
    d.raytrans_elu = int(opt.decoder.raytrans_act == "ELU")
    d.raytrans_posenc = int(bool(opt.decoder.raytrans_posenc))
    d.density_maskfill = int(bool(opt.decoder.density_maskfill))
    keep = []
    if d.raytrans_posenc:
        tab = raytrans_table.to(dev).float().contiguous()
        keep.append(tab)
        d.raytrans_table = _ptr(tab)
    out = {}
    for k, name in enumerate(DEC_TRAIN_TENSORS):
        w = _f32c(params[name], name)
        keep.append(w)
        d.w[k] = _ptr(w)
        if grads is not None and name in grads:
            gk = grads[name]
        else:
            gk = torch.zeros_like(w)
        if gk is not None:
            _f32c(gk, "grad " + name)
            out[name] = gk
        d.g[k] = _ptr(gk)
    g_cond = torch.zeros(n, cond_stride, device=dev) if want_g_cond else None
    ws = _grow_only_workspace(dev, lib.mnerf_decoder_backward_workspace_bytes(r, s) // 4, stream)
    with _on(dev, stream) as st:
        check(lib.mnerf_decoder_backward(C.byref(d), r, s, _ptr(x_ndc), _ptr(dirs), _ptr(cond), int(cond_stride), _ptr(g_rgb_s),
                                         _ptr(g_sigma), _ptr(g_cond), _ptr(ws), st), "mnerf_decoder_backward")
    return g_cond, out


_BWD_WS = {}


def _grow_only_workspace(dev, n_floats, stream=None):
    """Scratch of the backward kernels, kept per (device, stream) and only ever grown: one allocation per process instead of one
    torch.empty per chunk and batch element.  SHARED by mnerf_decoder_backward (10.4 KB per sample: 0.7 GB for 1 024 rays x 64
    samples) and mnerf_encoder_layer_backward (about 0.5 GB at the DTU shape): the kernels of a call are enqueued on one stream
    in order and the next call's kernels follow them on that stream, so reuse needs no extra synchronisation.
    An explicit `stream` is not torch's current stream: the buffer is then allocated UNDER that stream and every use is recorded
    on it (Tensor.record_stream), so a buffer dropped when the scratch grows is not handed to another stream's allocation while
    kernels on `stream` still use it.  `release_workspaces()` frees the cache (it also goes with torch.cuda.empty_cache() once
    released); MNERF_BWD_WS_CAP_MB caps what is KEPT between calls (a larger request is served by a one-off buffer)."""
    import torch
    cur = torch.cuda.current_stream(dev)
    st = stream if stream is not None else cur
    key = (torch.device(dev), st.cuda_stream)
    buf = _BWD_WS.get(key)
    if buf is None or buf.numel() < n_floats:
        with torch.cuda.stream(st):
            buf = torch.empty(n_floats, device=dev)
        cap = float(os.environ.get("MNERF_BWD_WS_CAP_MB", "0") or 0)
        if cap <= 0 or n_floats * 4 <= cap * 2 ** 20:
            _BWD_WS[key] = buf
    if st.cuda_stream != cur.cuda_stream:
        buf.record_stream(st)
    return buf


def release_workspaces():
    """Drop the cached backward scratch buffers (they return to torch's caching allocator; torch.cuda.empty_cache() then
    gives the memory back to the driver)."""
    _BWD_WS.clear()


def cost_volume(scene, rays, cond_stride, out=None, device=None, stream=None):
    """K1+K2 (matchnerf.py:209-293) -> cond [n_rays*S, cond_stride].  The launch device is ``out``'s, else
    ``device``, else the current one (it must be where the scene's maps live)."""
    import torch
    lib = load()
    n = rays.n_rays * rays.n_samples
    if out is None:
        out = torch.empty(n, cond_stride, device=torch.device(device) if device is not None else _current_device())
    with _on(out.device, stream) as st:
        check(lib.mnerf_cost_volume(C.byref(scene), C.byref(rays), int(cond_stride), _ptr(out), st),
              "mnerf_cost_volume")
    return out


def cost_volume_operand_bytes(scene):
    """Bytes of the operand image of a scene's feature maps (mnerf_cost_volume_operands)."""
    n = int(load().mnerf_cost_volume_operand_bytes(C.byref(scene)))
    if n < 0:
        raise MnerfError("mnerf_cost_volume_operand_bytes: unsupported scene")
    return n


def cost_volume_operands(scene, out=None, device=None, stream=None):
    """The split-fp16 operand image of ``scene.feat`` for the matrix form of the cost volume (ABI v9): returns the uint8
    tensor that holds it and points ``scene.feat_op`` at it.  The caller keeps the tensor alive as long as the scene is used."""
    import torch
    lib = load()
    n = cost_volume_operand_bytes(scene)
    if out is None or out.numel() < n:
        out = torch.empty(n, dtype=torch.uint8, device=torch.device(device) if device is not None else _current_device())
    with _on(out.device, stream) as st:
        check(lib.mnerf_cost_volume_operands(C.byref(scene), out.data_ptr(), st), "mnerf_cost_volume_operands")
    scene.feat_op = out.data_ptr()
    return out


def decoder_chunk(dec, view0, rays, cond, want_samples=False, stream=None):
    """K3+K4+K5 (cond_nerf.py:52-100, ray_transformer.py, nerf.py:101-124)."""
    import torch
    lib = load()
    r, s = rays.n_rays, rays.n_samples
    dev = cond.device
    rgb = torch.empty(r, 3, device=dev)
    depth = torch.empty(r, device=dev)
    opacity = torch.empty(r, device=dev)
    rgb_s = torch.empty(r, s, 3, device=dev) if want_samples else None
    sigma = torch.empty(r, s, device=dev) if want_samples else None
    with _on(dev, stream) as st:
        check(lib.mnerf_decoder_chunk(C.byref(dec), C.byref(view0), C.byref(rays), _ptr(cond), _ptr(rgb),
                                      _ptr(depth), _ptr(opacity), _ptr(rgb_s), _ptr(sigma), st),
              "mnerf_decoder_chunk")
    if want_samples:
        return rgb, depth, opacity, rgb_s, sigma
    return rgb, depth, opacity


def decoder_samples(dec, x_ndc, dirs, cond, legacy_coord=True, stream=None):
    """K3+K4 with caller-supplied inputs (the literal CondNeRF.forward, cond_nerf.py:52-100):
    x_ndc [R,S,3], dirs [R,S,3], cond [R*S, cond_stride] -> rgb_s [R,S,3], sigma [R,S]."""
    import torch
    lib = load()
    _f32c(x_ndc, "x_ndc"), _f32c(dirs, "dirs"), _f32c(cond, "cond")
    r, s, _ = x_ndc.shape
    if tuple(dirs.shape) != (r, s, 3) or cond.numel() != r * s * dec.cond_stride:
        raise MnerfError(f"decoder_samples: x_ndc {tuple(x_ndc.shape)}, dirs {tuple(dirs.shape)}, cond "
                         f"{tuple(cond.shape)} (stride {dec.cond_stride}) do not agree")
    dev = x_ndc.device
    rgb_s = torch.empty(r, s, 3, device=dev)
    sigma = torch.empty(r, s, device=dev)
    with _on(dev, stream) as st:
        check(lib.mnerf_decoder_samples(C.byref(dec), r, s, int(bool(legacy_coord)), _ptr(x_ndc), _ptr(dirs),
                                        _ptr(cond), _ptr(rgb_s), _ptr(sigma), st), "mnerf_decoder_samples")
    return rgb_s, sigma


class KernelTimer:
    """Per-kernel device timing with events recorded on the launch stream (bench.py roofline).
    ``spans[name]`` collects (start_event, end_event, n_rays) triples; ``summary`` syncs once."""

    def __init__(self):
        self.spans = {}

    def span(self, name, n_rays):
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.spans.setdefault(name, []).append((e0, e1, n_rays))
        return e0, e1

    def reset(self):
        self.spans = {}

    def summary(self):
        out = {}
        for spans in self.spans.values():
            for _, end, _ in spans:
                end.synchronize()
        for name, spans in self.spans.items():
            ms = [a.elapsed_time(b) for a, b, _ in spans]
            out[name] = dict(launches=len(ms), total_ms=sum(ms), avg_ms=sum(ms) / max(len(ms), 1),
                             rays=sum(n for _, _, n in spans))
        return out


def render_is_fused(scene, dec, rays):
    """True when the one-launch form of the ray chunk (mnerf_render_chunk_fused) exists for this configuration."""
    return bool(load().mnerf_render_chunk_is_fused(C.byref(scene), C.byref(dec), C.byref(rays)))


def render_takes_pose_table(scene, dec, n_samples, rays_per_pose):
    """True when a ray chunk of this scene / decoder may carry several target poses (``make_rays(pose_table_ptr=...)``)."""
    return bool(load().mnerf_render_takes_pose_table(C.byref(scene), C.byref(dec), int(n_samples), int(rays_per_pose)))


def render_chunk(scene, dec, rays, workspace, rgb, depth, opacity, stream=None, timer=None, fused=False):
    """a7 (matchnerf.py:88-143): writes rgb [R,3], depth [R], opacity [R] in place.
    ``fused=False`` (default, the faster form on MI355X): cost volume -> ``workspace`` -> decoder, two launches;
    with ``timer`` they go through their own entry points with spans "cost_volume" and "decoder".
    ``fused=True``: mnerf_render_chunk_fused, ONE launch, conditioning rows stay in LDS, ``workspace`` may be None
    (span "render_fused"); raises MnerfError where that form does not exist (``render_is_fused``)."""
    import torch
    lib = load()
    with _on(rgb.device, stream) as st:
        tst = stream if stream is not None else torch.cuda.current_stream(rgb.device)
        if fused:
            if timer is not None:
                e0, e1 = timer.span("render_fused", rays.n_rays)
                e0.record(tst)
            check(lib.mnerf_render_chunk_fused(C.byref(scene), C.byref(dec), C.byref(rays), _ptr(rgb), _ptr(depth),
                                               _ptr(opacity), st), "mnerf_render_chunk_fused")
            if timer is not None:
                e1.record(tst)
            return
        if timer is None:
            check(lib.mnerf_render_chunk(C.byref(scene), C.byref(dec), C.byref(rays), _ptr(workspace), _ptr(rgb),
                                         _ptr(depth), _ptr(opacity), st), "mnerf_render_chunk")
            return
        a0, a1 = timer.span("cost_volume", rays.n_rays)
        a0.record(tst)
        check(lib.mnerf_cost_volume(C.byref(scene), C.byref(rays), dec.cond_stride, _ptr(workspace), st),
              "mnerf_cost_volume")
        a1.record(tst)
        b0, b1 = timer.span("decoder", rays.n_rays)
        b0.record(tst)
        check(lib.mnerf_decoder_chunk(C.byref(dec), C.byref(scene.views[0]), C.byref(rays), _ptr(workspace),
                                      _ptr(rgb), _ptr(depth), _ptr(opacity), None, None, st), "mnerf_decoder_chunk")
        b1.record(tst)


def render_workspace_bytes(n_rays, n_samples, cond_stride):
    return int(load().mnerf_render_workspace_bytes(n_rays, n_samples, cond_stride))


def wa_math():
    """Matrix arithmetic of the window-attention kernel, MNERF_WA_MATH =
      'f16pre' (default)  split-fp16 products, K / V operand fragments + tile gains prepared once per call
      'f16x3'             the same arithmetic, every wave splits the K / V tiles itself (slower: 229 us per call)
      'bf16x6'            split-bf16, six products per MAC (204 us per call at 64x80 tokens)
      'f32'               exact-f32 MFMA"""
    m = os.environ.get("MNERF_WA_MATH", "f16pre")
    table = {"f16pre": WA_PRESPLIT_F16, "f16x3": WA_SPLIT_F16, "bf16x6": WA_SPLIT_BF16, "f32": WA_EXACT_F32}
    if m not in table:
        raise ValueError(f"MNERF_WA_MATH={m!r}: expected one of {sorted(table)}")
    return table[m]


def window_attention(q, k, v, h, w, num_splits, shifted, out=None, math=None, stream=None, row_stats=None):
    """K6 (gmflow/transformer.py:8-105). q,k,v [B,h*w,128] -> [B,h*w,128].  ``row_stats`` (training forward, default arithmetic
    only): a float32 [2, B*h*w] tensor that receives the softmax's row statistics for ``window_attention_backward``."""
    import torch
    lib = load()
    _f32c(q, "q"), _f32c(k, "k"), _f32c(v, "v")
    b, n, c = q.shape
    if c != 128 or n != h * w:
        raise MnerfError(f"window_attention: expected [B,{h * w},128], got {tuple(q.shape)}")
    if out is None:
        out = torch.empty_like(q)
    math = wa_math() if math is None else int(math)
    with _on(q.device, stream) as st:
        if math == WA_PRESPLIT_F16:
            nbytes = int(lib.mnerf_window_attention_workspace_bytes(b, h, w, int(num_splits)))
            ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=q.device)  # caching allocator: no hipMalloc
            if stream is not None:
                ws.record_stream(stream)
            if row_stats is not None:
                if tuple(row_stats.shape) != (2, b * n) or row_stats.dtype != torch.float32 or not row_stats.is_contiguous():
                    raise MnerfError(f"window_attention: row_stats must be a contiguous float32 [2, {b * n}] tensor")
                fn = lib.mnerf_window_attention_presplit_stats
                fn.restype = C.c_int
                fn.argtypes = [C.c_void_p] * 5 + [C.c_int32] * 5 + [C.c_void_p, C.c_size_t, C.c_void_p]
                check(fn(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), row_stats.data_ptr(), b, h, w, int(num_splits),
                         int(bool(shifted)), ws.data_ptr(), nbytes, st), "mnerf_window_attention_presplit_stats")
                return out
            check(lib.mnerf_window_attention_presplit(_ptr(q), _ptr(k), _ptr(v), _ptr(out), b, h, w, int(num_splits),
                                                      int(bool(shifted)), C.c_void_p(ws.data_ptr()), nbytes, st),
                  "mnerf_window_attention_presplit")
        elif row_stats is not None:
            raise MnerfError("window_attention: row_stats needs the default arithmetic (MNERF_WA_MATH=f16pre)")
        else:
            check(lib.mnerf_window_attention(_ptr(q), _ptr(k), _ptr(v), _ptr(out), b, h, w, int(num_splits),
                                             int(bool(shifted)), math, st), "mnerf_window_attention")
    return out


def window_attention_backward(q, k, v, out, g_out, h, w, num_splits, shifted, stream=None, row_stats=None):
    """K6 backward (mnerf_window_attention_backward): gradients of q, k, v [B,h*w,128] given the forward's ``out`` and its
    gradient ``g_out``; flash style (no score tensor), fp32-grade, deterministic.  ``row_stats``: what ``window_attention``
    filled in the forward (the statistics pass is skipped)."""
    import torch
    lib = load()
    for t, name in ((q, "q"), (k, "k"), (v, "v"), (out, "out"), (g_out, "g_out")):
        _f32c(t, name)
    b, n, c = q.shape
    if c != 128 or n != h * w or any(tuple(t.shape) != (b, n, c) for t in (k, v, out, g_out)):
        raise MnerfError(f"window_attention_backward: expected five [B,{h * w},128] tensors, got {tuple(q.shape)} ...")
    g_q, g_k, g_v = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    nbytes = int(lib.mnerf_window_attention_backward_workspace_bytes(b, h, w))
    ws = torch.empty(max(nbytes // 4, 4), device=q.device)
    if stream is not None:
        ws.record_stream(stream)
    with _on(q.device, stream) as st:
        if row_stats is not None:
            if tuple(row_stats.shape) != (2, b * n) or row_stats.dtype != torch.float32 or not row_stats.is_contiguous():
                raise MnerfError(f"window_attention_backward: row_stats must be a contiguous float32 [2, {b * n}] tensor")
            fn = lib.mnerf_window_attention_backward_stats
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p] * 9 + [C.c_int32] * 5 + [C.c_void_p, C.c_size_t, C.c_void_p]
            check(fn(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), g_out.data_ptr(), row_stats.data_ptr(), g_q.data_ptr(),
                     g_k.data_ptr(), g_v.data_ptr(), b, h, w, int(num_splits), int(bool(shifted)), ws.data_ptr(), ws.numel() * 4, st),
                  "mnerf_window_attention_backward_stats")
            return g_q, g_k, g_v
        check(lib.mnerf_window_attention_backward(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(g_out), _ptr(g_q), _ptr(g_k), _ptr(g_v),
                                                  b, h, w, int(num_splits), int(bool(shifted)), C.c_void_p(ws.data_ptr()),
                                                  ws.numel() * 4, st), "mnerf_window_attention_backward")
    return g_q, g_k, g_v


def qkv_projection(wstream, ews, x_q, x_kv=None, kv_swap=False, stream=None):
    """q, k, v = Wq x_q, Wk x_kv', Wv x_kv' of a GMFlow transformer layer in one launch (transformer.py:147-151);
    x_* [n_seq, seq_len, 128]; ``kv_swap``: x_kv' = x_kv with its batch halves exchanged.  wstream / ews from
    gmflow.pack_qkv."""
    import torch
    lib = load()
    x_kv = x_q if x_kv is None else x_kv
    _f32c(x_q, "x_q"), _f32c(x_kv, "x_kv"), _f32c(wstream, "wstream")
    if x_q.dim() != 3 or x_q.shape[2] != 128 or x_kv.shape != x_q.shape:
        raise MnerfError(f"qkv_projection: expected two [n_seq, seq_len, 128] tensors, got {tuple(x_q.shape)}, {tuple(x_kv.shape)}")
    if wstream.numel() != lib.mnerf_qkv_wstream_floats():
        raise MnerfError(f"qkv_projection: wstream has {wstream.numel()} floats, expected {lib.mnerf_qkv_wstream_floats()}")
    q, k, v = (torch.empty_like(x_q) for _ in range(3))
    ew = (C.c_int32 * 3)(*[int(e) for e in ews])
    with _on(x_q.device, stream) as st:
        check(lib.mnerf_qkv_projection(_ptr(wstream), ew, _ptr(x_q), _ptr(x_kv), int(bool(kv_swap)), _ptr(q), _ptr(k),
                                       _ptr(v), x_q.shape[0], x_q.shape[1], st), "mnerf_qkv_projection")
    return q, k, v


def qkv_window_images(wstream, ews, x_q, x_kv, kv_swap, h, w, num_splits, shifted, stream=None):
    """q|k|v projections with K and V written as the window attention's operand images (csrc/qkv.hip).
    -> (q [B,h*w,128], workspace) for ``window_attention_images`` with the same geometry."""
    import torch
    lib = load()
    x_kv = x_q if x_kv is None else x_kv
    _f32c(x_q, "x_q"), _f32c(x_kv, "x_kv"), _f32c(wstream, "wstream")
    b, n, c = x_q.shape
    if c != 128 or n != h * w or x_kv.shape != x_q.shape:
        raise MnerfError(f"qkv_window_images: expected two [B,{h * w},128] tensors, got {tuple(x_q.shape)}, {tuple(x_kv.shape)}")
    if wstream.numel() != lib.mnerf_qkv_wstream_floats():
        raise MnerfError(f"qkv_window_images: wstream has {wstream.numel()} floats, expected {lib.mnerf_qkv_wstream_floats()}")
    q = torch.empty_like(x_q)
    nbytes = int(lib.mnerf_window_attention_workspace_bytes(b, h, w, int(num_splits)))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x_q.device)
    if stream is not None:
        ws.record_stream(stream)
    ew = (C.c_int32 * 3)(*[int(e) for e in ews])
    with _on(x_q.device, stream) as st:
        check(lib.mnerf_qkv_window_images(_ptr(wstream), ew, _ptr(x_q), _ptr(x_kv), int(bool(kv_swap)), _ptr(q),
                                          C.c_void_p(ws.data_ptr()), nbytes, b, h, w, int(num_splits), int(bool(shifted)), st),
              "mnerf_qkv_window_images")
    return q, ws


def window_attention_images(q, workspace, h, w, num_splits, shifted, out=None, stream=None):
    """K6 on K / V operand images prepared by ``qkv_window_images`` (same geometry arguments)."""
    import torch
    lib = load()
    _f32c(q, "q")
    b, n, c = q.shape
    if c != 128 or n != h * w:
        raise MnerfError(f"window_attention_images: expected [B,{h * w},128], got {tuple(q.shape)}")
    if out is None:
        out = torch.empty_like(q)
    with _on(q.device, stream) as st:
        check(lib.mnerf_window_attention_images(_ptr(q), _ptr(out), b, h, w, int(num_splits), int(bool(shifted)),
                                                C.c_void_p(workspace.data_ptr()), workspace.numel(), st),
              "mnerf_window_attention_images")
    return out


def encoder_layer_backward(layer, attn, source, g_out, grads, stream=None, saved=None):
    """Backward of everything after the attention in a transformer layer (mnerf_encoder_layer_backward).  ``layer``: the
    TransformerLayer module (merge / norm1 / mlp / norm2 parameters are read in torch's layouts); attn, source, g_out [N,128];
    ``grads``: dict parameter tensor -> gradient tensor to ACCUMULATE into (a missing entry skips that parameter).
    ``saved``: (m1 [N,128], z1 [N,1024], m2 [N,128]) as ``encoder_block(..., save=True)`` returned them (z1, m2 None without an FFN;
    mnerf_encoder_layer_backward_saved: the GEMMs that would re-evaluate them are skipped).  -> (g_attn, g_source) [N,128]."""
    import torch
    lib = load()
    for t, name in ((attn, "attn"), (source, "source"), (g_out, "g_out")):
        _f32c(t, name)
    n, c = source.shape
    if c != 128 or tuple(attn.shape) != (n, c) or tuple(g_out.shape) != (n, c):
        raise MnerfError(f"encoder_layer_backward: attn {tuple(attn.shape)}, source {tuple(source.shape)}, g_out {tuple(g_out.shape)}")
    L = EncoderLayerTrain()
    L.ffn = int(not layer.no_ffn)
    keep = []

    def put(field, p):
        w = p.detach()
        _f32c(w, field)
        keep.append(w)
        setattr(L, field, w.data_ptr())
        g = grads.get(p)
        if g is not None:
            _f32c(g, "grad " + field)
            if tuple(g.shape) != tuple(p.shape):
                raise MnerfError(f"encoder_layer_backward: gradient of {field} has shape {tuple(g.shape)}")
            setattr(L, "g_" + field, g.data_ptr())

    put("w_merge", layer.merge.weight), put("ln1_w", layer.norm1.weight), put("ln1_b", layer.norm1.bias)
    if not layer.no_ffn:
        put("w_mlp0", layer.mlp[0].weight), put("w_mlp2", layer.mlp[2].weight)
        put("ln2_w", layer.norm2.weight), put("ln2_b", layer.norm2.bias)
    g_attn, g_source = torch.empty_like(source), torch.empty_like(source)
    ws = _grow_only_workspace(source.device, int(lib.mnerf_encoder_layer_backward_workspace_bytes(n)) // 4, stream)
    with _on(source.device, stream) as st:
        if saved is not None:
            m1, z1, m2 = saved
            _f32c(m1, "m1")
            if tuple(m1.shape) != (n, 128):
                raise MnerfError(f"encoder_layer_backward: saved m1 {tuple(m1.shape)}")
            if not layer.no_ffn:
                _f32c(z1, "z1"), _f32c(m2, "m2")
                if tuple(z1.shape) != (n, 1024) or tuple(m2.shape) != (n, 128):
                    raise MnerfError(f"encoder_layer_backward: saved z1 {tuple(z1.shape)}, m2 {tuple(m2.shape)}")
            fn = lib.mnerf_encoder_layer_backward_saved
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p] * 9 + [C.c_int32, C.c_void_p, C.c_void_p]
            check(fn(C.addressof(L), attn.data_ptr(), source.data_ptr(), g_out.data_ptr(), m1.data_ptr(),
                     None if layer.no_ffn else z1.data_ptr(), None if layer.no_ffn else m2.data_ptr(),
                     g_attn.data_ptr(), g_source.data_ptr(), n, ws.data_ptr(), st), "mnerf_encoder_layer_backward_saved")
            return g_attn, g_source
        check(lib.mnerf_encoder_layer_backward(C.byref(L), _ptr(attn), _ptr(source), _ptr(g_out), _ptr(g_attn), _ptr(g_source), n,
                                               C.c_void_p(ws.data_ptr()), st), "mnerf_encoder_layer_backward")
    return g_attn, g_source


def qkv_backward(w_q, w_k, w_v, x_q, x_kv, g_q, g_k, g_v, gw_q=None, gw_k=None, gw_v=None, stream=None):
    """Backward of the three bias-free projections (mnerf_qkv_backward): x_* and g_* [N,128], w_* [128,128];
    gw_* (optional) are ACCUMULATED into.  -> (g_xq, g_xkv) [N,128]."""
    import torch
    lib = load()
    ts = [w_q.detach(), w_k.detach(), w_v.detach(), x_q, x_kv, g_q, g_k, g_v]
    for t in ts:
        _f32c(t, "qkv_backward operand")
    n = x_q.shape[0]
    if any(tuple(t.shape) != (n, 128) for t in ts[3:]) or any(tuple(t.shape) != (128, 128) for t in ts[:3]):
        raise MnerfError("qkv_backward: expected [N,128] activations / gradients and [128,128] weights")
    g_xq, g_xkv = torch.empty_like(x_q), torch.empty_like(x_q)
    with _on(x_q.device, stream) as st:
        check(lib.mnerf_qkv_backward(*[_ptr(t) for t in ts], _ptr(g_xq), _ptr(g_xkv), _ptr(gw_q), _ptr(gw_k), _ptr(gw_v), n, st),
              "mnerf_qkv_backward")
    return g_xq, g_xkv


def debug_gemm(a, b, bias=None, out=None, mode=0, math="bf16x6", stream=None):
    """Test hook (mnerf_debug_gemm): C (mode 0: =, 1: +=, 2: atomic +=) a @ b (+ bias) for 2-D fp32 tensors of ANY strides (views and
    transposes go through as they are); ``math``: "bf16x6" (the library's default for products with I, J >= 128), "f16x3" (split-fp16 with row gains, round 6: selectable with MNERF_GEMM_MATH) or "f32"."""
    import torch
    lib = load()
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[0] or a.dtype != torch.float32 or b.dtype != torch.float32:
        raise MnerfError(f"debug_gemm: {tuple(a.shape)} @ {tuple(b.shape)}")
    I, K = a.shape
    J = b.shape[1]
    if out is None:
        out = torch.zeros(I, J, device=a.device)
    if out.stride(1) != 1:
        raise MnerfError("debug_gemm: out must have unit column stride")
    fn = lib.mnerf_debug_gemm
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                   C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    with _on(a.device, stream) as st:
        check(fn(a.data_ptr(), a.stride(0), a.stride(1), b.data_ptr(), b.stride(0), b.stride(1), out.data_ptr(), out.stride(0),
                 _ptr(bias), I, J, K, int(mode), {"f32": 0, "bf16x6": 1, "f16x3": 2}[math], st), "mnerf_debug_gemm")
    return out


def instance_norm(x, residual=None, relu_inner=False, relu_outer=False, eps=1e-5, out=None, out_absmax=None, stream=None):
    """F.instance_norm(x) of an NCHW tensor fused with the ReLU / residual add / ReLU that follow it in the GMFlow
    backbone (backbone.py:27-35): out = [relu](residual + [relu](IN(x))).  ``out_absmax``: absmax region that max|out|
    is merged into (the operand scale of the convolution that reads ``out``)."""
    import torch
    lib = load()
    _f32c(x, "x")
    if x.dim() != 4:
        raise MnerfError(f"instance_norm: expected [N,C,H,W], got {tuple(x.shape)}")
    if residual is not None:
        _f32c(residual, "residual")
        if residual.shape != x.shape:
            raise MnerfError(f"instance_norm: residual {tuple(residual.shape)} vs x {tuple(x.shape)}")
    if out is None:
        out = torch.empty_like(x)
    n, c, h, w = x.shape
    with _on(x.device, stream) as st:
        check(lib.mnerf_instance_norm(_ptr(x), _ptr(residual), _ptr(out), n * c, h * w, float(eps), int(bool(relu_inner)),
                                      int(bool(relu_outer)), _ptr(out_absmax), st), "mnerf_instance_norm")
    return out


def absmax_regions(n, device):
    """n zeroed absmax regions [n, ABSMAX_FLOATS] (include/mnerf.h): the hand-over of a tensor's largest magnitude from
    the kernel that writes it to the split-fp16 convolution that reads it."""
    import torch
    return torch.zeros(n, ABSMAX_FLOATS, device=device, dtype=torch.float32)


def absmax_value(region):
    """the maximum an absmax region holds (device tensor, no sync)"""
    return region.max()


def absmax(x, out, stream=None):
    """max|x| merged into the absmax region ``out`` (atomic maxima: zero it first)."""
    lib = load()
    _f32c(x, "x")
    with _on(x.device, stream) as st:
        check(lib.mnerf_absmax(_ptr(x), x.numel(), _ptr(out), st), "mnerf_absmax")
    return out


def upsample_bilinear2x(x, add=None, stream=None):
    """F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False) (+ add): [N,C,H,W] -> [N,C,2H,2W]."""
    import torch
    lib = load()
    _f32c(x, "x")
    n, c, h, w = x.shape
    out = torch.empty(n, c, 2 * h, 2 * w, device=x.device, dtype=torch.float32)
    if add is not None:
        _f32c(add, "add")
    with _on(x.device, stream) as st:
        check(lib.mnerf_upsample_bilinear2x(_ptr(x), _ptr(add), _ptr(out), n * c, h, w, st), "mnerf_upsample_bilinear2x")
    return out


def upsample_bilinear2x_backward(dout, stream=None):
    """the adjoint: [N,C,2H,2W] -> [N,C,H,W]"""
    import torch
    lib = load()
    _f32c(dout, "dout")
    n, c, h2, w2 = dout.shape
    din = torch.empty(n, c, h2 // 2, w2 // 2, device=dout.device, dtype=torch.float32)
    with _on(dout.device, stream) as st:
        check(lib.mnerf_upsample_bilinear2x_backward(_ptr(dout), _ptr(din), n * c, h2 // 2, w2 // 2, st), "mnerf_upsample_bilinear2x_backward")
    return din


def instance_norm_backward(x, dy, relu, eps=1e-5, stream=None):
    """dx of out = [relu](InstanceNorm2d(x)) (no affine): x, dy [N,C,H,W] -> [N,C,H,W] (csrc/instance_norm.hip)."""
    import torch
    lib = load()
    _f32c(x, "x"), _f32c(dy, "dy")
    n, c, h, w = x.shape
    dx = torch.empty_like(x)
    with _on(x.device, stream) as st:
        check(lib.mnerf_instance_norm_backward(_ptr(x), _ptr(dy), _ptr(dx), n * c, h * w, float(eps), int(bool(relu)), st),
              "mnerf_instance_norm_backward")
    return dx


def conv2d_backward_data(dy, weight, h_in, w_in, stride, stream=None):
    """dX of Conv2d(c_in, c_out, k, stride, padding=k//2) (csrc/conv_backward.hip).  dy [N,c_out,Ho,Wo], weight [c_out,c_in,k,k]
    -> [N,c_in,h_in,w_in]."""
    import torch
    lib = load()
    _f32c(dy, "dy")
    c_out, c_in, k, _ = weight.shape
    n = dy.shape[0]
    wt = weight.detach().permute(2, 3, 0, 1).contiguous()
    dx = torch.empty(n, c_in, h_in, w_in, device=dy.device, dtype=torch.float32)
    with _on(dy.device, stream) as st:
        check(lib.mnerf_conv2d_backward_data(_ptr(dy), _ptr(wt), _ptr(dx), n, c_in, c_out, int(h_in), int(w_in), int(k), int(stride), st),
              "mnerf_conv2d_backward_data")
    return dx


def conv2d_forward_f32(x, weight, bias, stride, stream=None):
    """Conv2d(c_in, c_out, k, stride, padding=k//2) in exact fp32 (the training forward; csrc/conv_backward.hip).
    x [N,c_in,H,W], weight [c_out,c_in,k,k] -> [N,c_out,Ho,Wo]."""
    import torch
    lib = load()
    _f32c(x, "x")
    c_out, c_in, k, _ = weight.shape
    n, _, h, w = x.shape
    wt = weight.detach().permute(2, 3, 1, 0).contiguous()
    ho, wo = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
    y = torch.empty(n, c_out, ho, wo, device=x.device, dtype=torch.float32)
    b = bias.detach().contiguous() if bias is not None else None
    with _on(x.device, stream) as st:
        check(lib.mnerf_conv2d_forward_f32(_ptr(x), _ptr(wt), _ptr(b), _ptr(y), n, c_in, c_out, h, w, int(k), int(stride), st),
              "mnerf_conv2d_forward_f32")
    return y


def conv_stem_backward_weight(x, dy, stream=None):
    """dW [64,3,7,7] of the stem Conv2d(3, 64, 7, 2, 3): x [N,3,H,W], dy [N,64,(H-1)//2+1,(W-1)//2+1]."""
    import torch
    lib = load()
    _f32c(x, "x"), _f32c(dy, "dy")
    n, _, h, w = x.shape
    nbytes = int(lib.mnerf_conv_stem_backward_weight_workspace_bytes(n, h, w))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    if stream is not None:
        ws.record_stream(stream)
    dw = torch.empty(64, 3, 7, 7, device=x.device, dtype=torch.float32)
    with _on(x.device, stream) as st:
        check(lib.mnerf_conv_stem_backward_weight(_ptr(x), _ptr(dy), _ptr(dw), ws.data_ptr(), nbytes, n, h, w, st),
              "mnerf_conv_stem_backward_weight")
    return dw


def conv2d_backward_weight(x, dy, ksize, stride, x_absmax=None, dy_absmax=None, stream=None):
    """dW of the same convolution: x [N,c_in,H,W], dy [N,c_out,Ho,Wo] -> [c_out,c_in,k,k].  With the two absmax regions (max|x|,
    max|dy|): three split-fp16 products per MAC on the 16-bit matrix pipe; without: exact-f32 matrix products."""
    import torch
    lib = load()
    _f32c(x, "x"), _f32c(dy, "dy")
    n, c_in, h, w = x.shape
    c_out = dy.shape[1]
    nbytes = int(lib.mnerf_conv2d_backward_weight_workspace_bytes(n, c_in, c_out, h, w, int(ksize), int(stride)))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    if stream is not None:
        ws.record_stream(stream)
    dw = torch.empty(c_out, c_in, ksize, ksize, device=x.device, dtype=torch.float32)
    with _on(x.device, stream) as st:
        if x_absmax is not None and dy_absmax is not None:
            check(lib.mnerf_conv2d_backward_weight_f16x3(_ptr(x), _ptr(dy), _ptr(x_absmax), _ptr(dy_absmax), _ptr(dw), ws.data_ptr(), nbytes, n,
                                                         c_in, c_out, h, w, int(ksize), int(stride), st), "mnerf_conv2d_backward_weight_f16x3")
        else:
            check(lib.mnerf_conv2d_backward_weight(_ptr(x), _ptr(dy), _ptr(dw), ws.data_ptr(), nbytes, n, c_in, c_out, h, w, int(ksize),
                                                   int(stride), st), "mnerf_conv2d_backward_weight")
    return dw


def conv_stem(x, wstream, ew, in_absmax, out=None, stream=None):
    """The backbone's 7x7 stride-2 stem (3 -> 64 channels, backbone.py:45) on the split-fp16 matrix path.
    x [N,3,H,W] -> [N,64,(H-1)//2+1,(W-1)//2+1]; wstream / ew from gmflow.pack_conv_stem."""
    import torch
    lib = load()
    _f32c(x, "x"), _f32c(wstream, "wstream")
    if x.dim() != 4 or x.shape[1] != 3:
        raise MnerfError(f"conv_stem: expected [N,3,H,W], got {tuple(x.shape)}")
    if wstream.numel() != lib.mnerf_conv_stem_wstream_floats():
        raise MnerfError(f"conv_stem: wstream has {wstream.numel()} floats, expected {lib.mnerf_conv_stem_wstream_floats()}")
    n, _, h, w = x.shape
    if out is None:
        out = torch.empty(n, 64, (h - 1) // 2 + 1, (w - 1) // 2 + 1, device=x.device, dtype=torch.float32)
    with _on(x.device, stream) as st:
        check(lib.mnerf_conv_stem(_ptr(wstream), int(ew), _ptr(x), _ptr(in_absmax), _ptr(out), n, h, w, st), "mnerf_conv_stem")
    return out


def conv2d(x, wstream, bias, c_in, c_out, ksize, stride, ew, in_absmax, leaky=1.0, channels_last=False, upsample2x=False,
           add_bilinear2x=None, out_layout=CONV_OUT_NCHW, add_channel_last=None, out_absmax=None, out=None, stream=None):
    """Split-fp16 implicit-GEMM convolution (csrc/conv.hip; gmflow/backbone.py, superres.py).  x [N,c_in,H,W], or
    [N,H,W,c_in] with ``channels_last``; ``wstream`` from gmflow.pack_conv; ``in_absmax``: absmax region filled by the
    producer of x; ``add_bilinear2x`` [N,c_out,H_out/2,W_out/2]: its bilinear 2x up-sampling is added to the result.
    ``out_layout``: CONV_OUT_NCHW -> [N,c_out,H_out,W_out]; CONV_OUT_CHANNEL_LAST -> tokens [N,H_out,W_out,c_out]
    (+ ``add_channel_last`` [H_out*W_out, c_out] added to every image); CONV_OUT_PAIR_MAJOR -> the cost volume's layout
    [N/2,2,H_out,W_out,c_out]."""
    import torch
    lib = load()
    _f32c(x, "x"), _f32c(wstream, "wstream")
    if x.dim() != 4:
        raise MnerfError(f"conv2d: expected a 4-D tensor, got {tuple(x.shape)}")
    n, h, w = (x.shape[0], x.shape[1], x.shape[2]) if channels_last else (x.shape[0], x.shape[2], x.shape[3])
    if (x.shape[3] if channels_last else x.shape[1]) != c_in:
        raise MnerfError(f"conv2d: input {tuple(x.shape)} does not have {c_in} channels")
    up = 1 if upsample2x else 0
    pad = ksize // 2
    h_out = ((h << up) + 2 * pad - ksize) // stride + 1
    w_out = ((w << up) + 2 * pad - ksize) // stride + 1
    if add_bilinear2x is not None:
        _f32c(add_bilinear2x, "add_bilinear2x")
        if tuple(add_bilinear2x.shape) != (n, c_out, h_out // 2, w_out // 2):
            raise MnerfError(f"conv2d: add_bilinear2x {tuple(add_bilinear2x.shape)} vs output {(n, c_out, h_out, w_out)}")
    if out is None:
        shape = {CONV_OUT_NCHW: (n, c_out, h_out, w_out), CONV_OUT_CHANNEL_LAST: (n, h_out, w_out, c_out),
                 CONV_OUT_PAIR_MAJOR: (n // 2, 2, h_out, w_out, c_out)}[out_layout]
        out = torch.empty(shape, device=x.device, dtype=torch.float32)
    if add_channel_last is not None:
        _f32c(add_channel_last, "add_channel_last")
        if add_channel_last.numel() != h_out * w_out * c_out:
            raise MnerfError(f"conv2d: add_channel_last {tuple(add_channel_last.shape)} vs {(h_out * w_out, c_out)}")
    cv = ConvLayer()
    cv.wstream, cv.wstream_floats = wstream.data_ptr(), wstream.numel()
    cv.bias = bias.data_ptr() if bias is not None else None
    cv.c_in, cv.c_out, cv.ksize, cv.stride, cv.ew, cv.leaky_slope = c_in, c_out, ksize, stride, int(ew), float(leaky)
    with _on(x.device, stream) as st:
        check(lib.mnerf_conv2d(C.byref(cv), _ptr(x), int(bool(channels_last)), up, _ptr(in_absmax), _ptr(add_bilinear2x),
                               _ptr(add_channel_last), _ptr(out), int(out_layout), _ptr(out_absmax), n, h, w, st),
              "mnerf_conv2d")
    return out


def encoder_block(attn, source, wstream, ln, ffn, ews, out=None, stream=None, save=False):
    """K7 (gmflow/transformer.py:176-185): out = source + norm2(mlp(cat[source, norm1(merge(attn))])) (or without the
    FFN).  attn, source [N,128]; wstream / ews from gmflow.pack_encoder_block; ln [4,128].  ``save`` (training): -> (out, m1 [N,128], z1 [N,1024],
    m2 [N,128]) - merge's output before norm1 and, with an FFN (None otherwise), mlp.0's output before the GELU and mlp.2's output
    before norm2 - for ``encoder_layer_backward(..., saved=(m1, z1, m2))`` (mnerf_encoder_block_save)."""
    import torch
    lib = load()
    _f32c(attn, "attn"), _f32c(source, "source"), _f32c(wstream, "wstream"), _f32c(ln, "ln")
    n, c = source.shape
    if c != 128 or tuple(attn.shape) != (n, c) or tuple(ln.shape) != (4, 128):
        raise MnerfError(f"encoder_block: attn {tuple(attn.shape)}, source {tuple(source.shape)}, ln {tuple(ln.shape)}")
    if out is None:
        out = torch.empty_like(source)
    blk = EncoderLayer()
    blk.wstream, blk.wstream_floats, blk.ln = wstream.data_ptr(), wstream.numel(), ln.data_ptr()
    blk.ffn, blk.ew_merge, blk.ew_w1, blk.ew_w2 = int(bool(ffn)), int(ews[0]), int(ews[1]), int(ews[2])
    with _on(source.device, stream) as st:
        if save:
            m1 = torch.empty(n, 128, device=source.device)
            z1 = torch.empty(n, 1024, device=source.device) if ffn else None
            m2 = torch.empty(n, 128, device=source.device) if ffn else None
            fn = lib.mnerf_encoder_block_save
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p] * 7 + [C.c_int32, C.c_void_p]
            check(fn(C.addressof(blk), attn.data_ptr(), source.data_ptr(), out.data_ptr(), m1.data_ptr(),
                     z1.data_ptr() if ffn else None, m2.data_ptr() if ffn else None, n, st), "mnerf_encoder_block_save")
            return out, m1, z1, m2
        check(lib.mnerf_encoder_block(C.byref(blk), _ptr(attn), _ptr(source), _ptr(out), n, st), "mnerf_encoder_block")
    return out
