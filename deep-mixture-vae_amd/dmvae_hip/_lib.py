"""ctypes binding of libdmvae_hip.so (C ABI: include/dmvae_hip.h).

The library is the product path: if it is missing this module raises at import
time -- there is no CPU fallback of any kind.
"""
import ctypes as C
import os

# torch must load its HIP runtime FIRST: libdmvae_hip.so then binds to that same
# libamdhip64 instance (streams and device pointers are shared with torch).  Loading
# the library before torch pulls in a second runtime and every launch fails with
# "no ROCm-capable device is detected".
import torch  # noqa: F401  (import order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
# DMVAE_HIP_LIB: another build of the SAME library (e.g. a measurement variant from tools/ablate.sh)
LIB_PATH = os.environ.get("DMVAE_HIP_LIB") or os.path.join(_HERE, "libdmvae_hip.so")

ABI_VERSION = 5          # DMVAE_ABI_VERSION of include/dmvae_hip.h
F32, BF16 = 0, 1
EUNSUPPORTED = -2        # DMVAE_EUNSUPPORTED
ADAM_ZERO_GRAD, ADAM_IEEE, ADAM_SHADOW = 1, 2, 4
GEMM_FWD, GEMM_DX, GEMM_DW = 0, 1, 2
(EPI_BIAS_RELU, EPI_BIAS_F32, EPI_BIAS_RECON, EPI_RELU_MASK, EPI_LATENT,
 EPI_STORE_F32, EPI_ATOMIC_F32, EPI_BIAS_SIGMOID, EPI_ADAM) = range(9)
MAX_LAYERS = 8
TRUNK_MLP, TRUNK_CNN = 0, 1
MODEL_DMVAE, MODEL_VADE = 0, 1

EXPORTS = [
    "dmvae_gemm", "dmvae_gemm_partials", "dmvae_gemm_grouped_dw", "dmvae_gemm_grouped", "dmvae_gemm_grouped_dw_adam", "dmvae_plan_train_step", "dmvae_latent_ws_bytes", "dmvae_latent_nblocks_vade",
    "dmvae_plan_forward_backward_stage", "dmvae_plan_grad_buckets", "dmvae_plan_set_stage_groups", "dmvae_plan_update_range",
"dmvae_latent_nblocks", "dmvae_latent_fwd", "dmvae_heads_latent_fwd", "dmvae_heads_latent_ok", "dmvae_heads_latent_kslice_floats",
    "dmvae_recon_fwd_bwd", "dmvae_recon_nblocks", "dmvae_colsum", "dmvae_loss_finalize",
    "dmvae_adam_tf", "dmvae_adam_finish", "dmvae_gather_rows", "dmvae_philox_normal",
    "dmvae_philox_gumbel", "dmvae_cast_f32_to_bf16", "dmvae_cast_bf16_to_f32",
    "dmvae_plan_create", "dmvae_plan_destroy", "dmvae_plan_sizes", "dmvae_plan_tensor",
    "dmvae_plan_bind", "dmvae_plan_load_batch", "dmvae_plan_load_batch_step", "dmvae_plan_prefetch_batch", "dmvae_plan_swap_batch", "dmvae_plan_forward_backward",
    "dmvae_plan_update", "dmvae_plan_encode", "dmvae_plan_decode", "dmvae_plan_view",
    "dmvae_prof_enable", "dmvae_prof_collect", "dmvae_debug_spin", "dmvae_debug_strip_fwd2", "dmvae_debug_stamps", "dmvae_debug_anatomy", "dmvae_debug_anatomy256", "dmvae_debug_set_tile", "dmvae_debug_set_knob", "dmvae_abi_version", "dmvae_last_error",
]


class Epilogue(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("m_valid", C.c_int32), ("n_valid", C.c_int32), ("d_off", C.c_int32),
        ("recon_kind", C.c_int32), ("scale", C.c_float),
        ("out", C.c_void_p), ("ldo", C.c_int64),
        ("out2", C.c_void_p), ("ldo2", C.c_int64),
        ("bias", C.c_void_p),
        ("aux0", C.c_void_p), ("ld0", C.c_int64),
        ("aux1", C.c_void_p), ("ld1", C.c_int64),
        ("aux2", C.c_void_p), ("ld2", C.c_int64),
        ("partials", C.c_void_p),
    ]


class GemmProblem(C.Structure):
    _fields_ = [("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("reserved", C.c_int32),
                ("A", C.c_void_p), ("lda", C.c_int64), ("B", C.c_void_p), ("ldb", C.c_int64),
                ("epi", Epilogue)]


class AdamCtx(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p),
                ("param_bf16", C.c_void_p), ("state", C.c_void_p),
                ("beta1", C.c_float), ("beta2", C.c_float), ("epsilon", C.c_float), ("grad_scale", C.c_float),
                ("store_grad", C.c_int32), ("ieee", C.c_int32), ("seg_off", C.c_int64), ("seg_n", C.c_int64)]


class LatentArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("B_pad", C.c_int32), ("D", C.c_int32), ("K", C.c_int32),
        ("mode", C.c_int32), ("act_dtype", C.c_int32),
        ("kl_ratio", C.c_float), ("temperature", C.c_float), ("inv_B", C.c_float),
        ("seed", C.c_uint64), ("noise_step", C.c_uint64),
        ("mean", C.c_void_p), ("ld_mean", C.c_int64),
        ("log_var", C.c_void_p), ("ld_log_var", C.c_int64),
        ("logits", C.c_void_p), ("ld_logits", C.c_int64),
        ("eps", C.c_void_p), ("ld_eps", C.c_int64),
        ("gumbel", C.c_void_p), ("ld_gumbel", C.c_int64),
        ("prior_means", C.c_void_p), ("prior_log_vars", C.c_void_p),
        ("Z_act", C.c_void_p), ("ld_Z", C.c_int64),
        ("Z_f32", C.c_void_p), ("ld_Zf", C.c_int64),
        ("weights", C.c_void_p), ("ld_w", C.c_int64),
        ("gmu", C.c_void_p), ("glv", C.c_void_p), ("clv", C.c_void_p), ("ld_g", C.c_int64),
        ("dlogits_act", C.c_void_p), ("ld_dl", C.c_int64),
        ("dprior_partials", C.c_void_p), ("loss_partials", C.c_void_p),
        ("state", C.c_void_p),
        ("mfma_ws", C.c_void_p), ("mfma_ws_bytes", C.c_int64),
    ]


class HeadsArgs(C.Structure):          # dmvae_heads_args
    _fields_ = [("hz", C.c_void_p), ("lda", C.c_int64),
                ("Hp", C.c_int32), ("Dp", C.c_int32), ("Kp", C.c_int32), ("reserved", C.c_int32),
                ("W_mv", C.c_void_p), ("ld_mv", C.c_int64), ("W_lg", C.c_void_p), ("ld_lg", C.c_int64),
                ("b_mv", C.c_void_p), ("b_lg", C.c_void_p),
                ("kslices", C.c_int32), ("reserved2", C.c_int32), ("kslice_ws", C.c_void_p), ("kslice_ws_floats", C.c_int64), ("kslice_tick", C.c_void_p)]


class State(C.Structure):
    _fields_ = [
        ("adam_t", C.c_uint64), ("noise_step", C.c_uint64),
        ("batch_cursor", C.c_uint32), ("batches_per_epoch", C.c_uint32),
        ("kl_ratio", C.c_float), ("lr", C.c_float), ("epoch_weight", C.c_float), ("lr_t", C.c_float),
        ("epoch_loss", C.c_float), ("epoch_recon", C.c_float), ("epoch_klz", C.c_float), ("epoch_klc", C.c_float),
        ("last_loss", C.c_float), ("last_recon", C.c_float), ("last_klz", C.c_float), ("last_klc", C.c_float),
    ]


class Config(C.Structure):
    _fields_ = [
        ("input_dim", C.c_int32), ("latent_dim", C.c_int32), ("n_classes", C.c_int32),
        ("n_enc", C.c_int32), ("enc", C.c_int32 * MAX_LAYERS),
        ("head_dim", C.c_int32),
        ("n_dec", C.c_int32), ("dec", C.c_int32 * MAX_LAYERS),
        ("input_type", C.c_int32), ("dtype", C.c_int32), ("max_batch", C.c_int32), ("mode", C.c_int32),
        ("temperature", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("adam_eps", C.c_float),
        ("seed", C.c_uint64), ("deterministic", C.c_int32), ("trunk", C.c_int32),
        ("model", C.c_int32), ("adam_ieee", C.c_int32),
    ]


class TensorInfo(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("offset", C.c_int64), ("rows", C.c_int32), ("cols", C.c_int32),
                ("ld", C.c_int64)]


class Sizes(C.Structure):
    _fields_ = [("param_elems", C.c_int64), ("work_bytes", C.c_int64), ("batch_pad", C.c_int32),
                ("input_pad", C.c_int32), ("n_tensors", C.c_int32), ("reserved", C.c_int32)]


class Buffers(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p),
                ("param_bf16", C.c_void_p), ("work", C.c_void_p), ("state", C.c_void_p), ("arena_elems", C.c_int64)]


class ProfRow(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("total_ms", C.c_double),
                ("flops", C.c_double), ("bytes", C.c_double), ("kernel_ms", C.c_double), ("kernel_launches", C.c_int64)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libdmvae_hip.so not found at %s: build it with `python deep-mixture-vae_amd/build.py` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u32, u64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_uint64, C.c_float
    P = C.POINTER
    sig = {
        "dmvae_gemm": [vp, i32, i32, i32, i32, i32, vp, i64, vp, i64, P(Epilogue), i32],
        "dmvae_gemm_partials": [i32, i32, i32],
        "dmvae_gemm_grouped_dw": [vp, i32, P(GemmProblem), i32],
        "dmvae_gemm_grouped": [vp, i32, i32, P(GemmProblem), i32],
        "dmvae_gemm_grouped_dw_adam": [vp, P(GemmProblem), i32, P(AdamCtx)],
        "dmvae_plan_train_step": [vp, vp, i32, vp, i64, vp, i64, f32],
        "dmvae_plan_forward_backward_stage": [vp, vp, i32, i32, vp, i64, vp, i64, f32],
        "dmvae_plan_grad_buckets": [vp, P(i64)],
        "dmvae_plan_set_stage_groups": [vp, i32],
        "dmvae_plan_update_range": [vp, vp, f32, i64, i64],
        "dmvae_latent_nblocks": [i32, i32, i32],
        "dmvae_latent_nblocks_vade": [i32],
        "dmvae_latent_ws_bytes": [i32, i32, i32, i32],
        "dmvae_latent_fwd": [vp, P(LatentArgs)],
        "dmvae_heads_latent_fwd": [vp, P(HeadsArgs), P(LatentArgs)],
        "dmvae_heads_latent_ok": [i32, i32, i32, i32, i32, i32, i32],
        "dmvae_heads_latent_kslice_floats": [i32, i32, i32],
        "dmvae_recon_fwd_bwd": [vp, i32, i32, i32, i32, i32, i32, vp, i64, vp, i64, f32, vp, i64, vp],
        "dmvae_recon_nblocks": [i32, i32],
        "dmvae_colsum": [vp, i32, vp, i64, i32, i32, vp],
        "dmvae_loss_finalize": [vp, vp, i32, vp, i32, f32, vp],
        "dmvae_adam_tf": [vp, i64, vp, vp, vp, vp, vp, f32, f32, f32, f32, f32, i32, u64, vp],
        "dmvae_adam_finish": [vp, vp],
        "dmvae_gather_rows": [vp, i32, vp, i64, i32, vp, i64, i32, i32, i32, vp, i64, vp, i64, vp],
        "dmvae_philox_normal": [vp, vp, i64, u64, u64, u32],
        "dmvae_philox_gumbel": [vp, vp, i64, u64, u64, u32],
        "dmvae_cast_f32_to_bf16": [vp, vp, vp, i64],
        "dmvae_cast_bf16_to_f32": [vp, vp, vp, i64],
        "dmvae_plan_create": [P(Config), P(vp)],
        "dmvae_plan_destroy": [vp],
        "dmvae_plan_sizes": [vp, P(Sizes)],
        "dmvae_plan_tensor": [vp, i32, P(TensorInfo)],
        "dmvae_plan_bind": [vp, P(Buffers)],
        "dmvae_plan_load_batch": [vp, vp, vp, i64, vp, i64, i32, i32],
        "dmvae_plan_load_batch_step": [vp, vp, vp, i64, vp, i64, i32, i32],
        "dmvae_plan_prefetch_batch": [vp, vp, i64, vp, i64, i32, i32],
        "dmvae_plan_swap_batch": [vp],
        "dmvae_plan_forward_backward": [vp, vp, i32, vp, i64, vp, i64, f32],
        "dmvae_plan_update": [vp, vp, f32],
        "dmvae_plan_encode": [vp, vp, i32],
        "dmvae_plan_decode": [vp, vp, vp, i64, i32],
        "dmvae_plan_view": [vp, C.c_char_p, P(vp), P(i64), P(C.c_int32)],
        "dmvae_prof_enable": [i32],
        "dmvae_debug_spin": [vp, i32],
        "dmvae_debug_strip_fwd2": [vp, i32, i32, vp, i64, vp, i64, vp, vp, i64, vp, vp, i64, vp, i64],
        "dmvae_debug_stamps": [P(vp)],
        "dmvae_debug_anatomy": [P(vp)],
        "dmvae_debug_anatomy256": [P(vp)],
        "dmvae_prof_collect": [P(ProfRow), i32],
        "dmvae_debug_set_tile": [i32, i32],
        "dmvae_debug_set_knob": [i32, i32],
        "dmvae_abi_version": [],
        "dmvae_last_error": [],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)          # AttributeError here = symbol missing from the .so
        fn.argtypes = args
        fn.restype = C.c_int
    lib.dmvae_last_error.restype = C.c_char_p
    lib.dmvae_latent_ws_bytes.restype = C.c_int64
    lib.dmvae_heads_latent_kslice_floats.restype = C.c_int64
    lib.dmvae_plan_destroy.restype = None
    got = lib.dmvae_abi_version()
    if got != ABI_VERSION:       # the public structs grew between versions: a mismatched pair would read past them
        raise ImportError("%s reports ABI version %d, this binding is written against %d (include/dmvae_hip.h): rebuild it with "
                          "`python deep-mixture-vae_amd/build.py`" % (LIB_PATH, got, ABI_VERSION))
    return lib


lib = _load()


class DmvaeError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        msg = lib.dmvae_last_error().decode("utf-8", "replace")
        raise DmvaeError("%s failed (code %d): %s" % (what or "dmvae call", rc, msg))


def ptr(t):
    """device pointer of a torch tensor (or None)"""
    return None if t is None else C.c_void_p(t.data_ptr())
