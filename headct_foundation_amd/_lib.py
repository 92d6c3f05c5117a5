"""ctypes binding of libheadct_hip.so (include/headct_hip.h).

The product path has NO CPU fallback: if the library is missing or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libheadct_hip.so")

HCT_F32, HCT_BF16, HCT_F16 = 0, 1, 2
HCT_ACT_NONE, HCT_ACT_GELU, HCT_ACT_DGELU, HCT_ACT_TANH, HCT_ACT_GELU_D, HCT_ACT_MULAUX = 0, 1, 2, 3, 4, 5
ACT_NONE, ACT_GELU, ACT_DGELU = 0, 1, 2

c_void_p, c_int, c_float, c_size_t, c_int64 = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_int64


class GemmArgs(C.Structure):
    _fields_ = [
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("A", c_void_p), ("a_dtype", c_int), ("lda", c_int64), ("transA", c_int),
        ("B", c_void_p), ("b_dtype", c_int), ("ldb", c_int64), ("transB", c_int),
        ("C", c_void_p), ("c_dtype", c_int), ("ldc", c_int64),
        ("bias", c_void_p), ("residual", c_void_p), ("ldr", c_int64),
        ("act", c_int),
        ("aux", c_void_p), ("aux_dtype", c_int), ("ldaux", c_int64),
        ("C2", c_void_p), ("c2_dtype", c_int), ("ldc2", c_int64),
        ("alpha", c_float), ("force_generic", c_int), ("colsum_out", c_void_p), ("workspace_armed", c_int),
    ]


class ProfShape(C.Structure):
    """include/headct_hip.h hct_prof_shape: launches of one GEMM shape recorded by the in-library HIP-event profile"""
    _fields_ = [("M", c_int), ("N", c_int), ("K", c_int), ("mode", c_int), ("tiles", c_int), ("sk_tiles", c_int),
                ("launches", c_int64), ("total_ms", C.c_double), ("work", C.c_double), ("bytes", C.c_double)]


def prof_shapes(lib, kernel_class: int, cap: int = 256):
    """List of ProfShape entries of the recorded launches of a kernel class (0 = NT GEMM, 1 = wgrad GEMM)."""
    buf = (ProfShape * cap)()
    n = lib.hct_prof_shapes(kernel_class, C.cast(buf, c_void_p), cap)
    if n < 0:
        raise HctError("hct_prof_shapes failed")
    return [buf[i] for i in range(min(n, cap))]


class MaeConfig(C.Structure):
    _fields_ = [
        ("input_size", c_int), ("patch_size", c_int), ("in_chans", c_int), ("mask_ratio", C.c_double),
        ("pos_embed", c_int),
        ("encoder_depth", c_int), ("encoder_embed_dim", c_int), ("encoder_mlp_dim", c_int), ("encoder_num_heads", c_int),
        ("decoder_depth", c_int), ("decoder_embed_dim", c_int), ("decoder_mlp_dim", c_int), ("decoder_num_heads", c_int),
        ("norm_pix_loss", c_int), ("use_bias", c_int),
        ("encoder_only", c_int), ("num_register_tokens", c_int), ("final_norm_eps", c_float),
    ]


class ParamInfo(C.Structure):
    _fields_ = [
        ("name", C.c_char * 96), ("ndim", c_int), ("shape", c_int64 * 5), ("offset", c_int64), ("numel", c_int64),
        ("requires_grad", c_int), ("is_matrix", c_int), ("bf16_t_offset", c_int64),
    ]


_PROTOS = {
    # name: (restype, argtypes)
    "hct_last_error_string": (C.c_char_p, []),
    "hct_version": (c_int, []),
    "hct_has_mfma_kernels": (c_int, []),
    "hct_gemm_workspace_bytes": (c_size_t, [C.POINTER(GemmArgs)]),
    "hct_gemm_nt_flags_offset": (c_size_t, [c_size_t]),
    "hct_set_cu_reserve": (None, [c_int]),
    "hct_gemm": (c_int, [C.POINTER(GemmArgs), c_void_p, c_size_t, c_void_p]),
    "hct_mask_rank": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "hct_patch_gather": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "hct_encoder_assemble_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "hct_encoder_assemble_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "hct_assemble_bwd_workspace_bytes": (c_size_t, [c_int]),
    "hct_layernorm_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "hct_layernorm_bwd_workspace_bytes": (c_size_t, [c_int, c_int]),
    "hct_layernorm_bwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                  c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "hct_layernorm_bwd_mapped": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                         c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "hct_tail_rows": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "hct_gather_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "hct_attention_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "hct_attention_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "hct_debug_force_simple_attention": (None, [c_int]),
    "hct_decoder_assemble_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "hct_decoder_assemble_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                         c_void_p, c_size_t, c_void_p]),
    "hct_masked_mse": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "hct_unpatchify": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "hct_vit_assemble_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "hct_channel_norm": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int64, c_int, c_void_p]),
    "hct_query_attention": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "hct_head_linear": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int,
                                c_int, c_void_p]),
    "hct_batchnorm_stats": (c_int, [c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "hct_softmax_xent": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "hct_head_linear_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "hct_dino_loss_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "hct_dino_loss": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_size_t, c_void_p]),
    "hct_dino_center_update": (c_int, [c_void_p, c_void_p, c_int, C.c_double, C.c_double, c_void_p]),
    "hct_ema_update": (c_int, [c_void_p, c_void_p, c_int64, C.c_double, c_void_p]),
    "hct_l2norm_rows_fwd": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "hct_l2norm_rows_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "hct_weight_norm_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "hct_weight_norm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "hct_bn_gelu_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "hct_bn_gelu_bwd_sums": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "hct_bn_gelu_bwd_apply": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, C.c_double, c_int, c_int, c_void_p, c_int,
                                      c_void_p]),
    "hct_hu_window": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "hct_augment_volume": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "hct_gaussian_smooth3d": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "hct_pos_embed_interp3d": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "hct_colsum_workspace_bytes": (c_size_t, [c_int, c_int]),
    "hct_colsum": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "hct_cast": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_void_p]),
    "hct_transpose_cast": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "hct_grad_norms_workspace_bytes": (c_size_t, [c_int64]),
    "hct_grad_norms": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_float, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "hct_adamw_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_float, c_float,
                               c_float, c_float, c_float, c_int, c_void_p, c_void_p]),
    "hct_debug_set_gemm_variant": (None, [c_int]),
    "hct_debug_set_gemm_stagger": (None, [c_int]),
    "hct_gemm_nt_stream_k_bytes": (c_size_t, []),
    "hct_gemm_tn_group_workspace_bytes": (c_size_t, [c_int]),
    "hct_gemm_tn_group_prepare": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "hct_gemm_tn_group_run": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "hct_prof_enable": (None, [c_int]),
    "hct_prof_reset": (None, []),
    "hct_prof_read": (c_int, [c_int, C.POINTER(C.c_double), C.POINTER(c_int64), C.POINTER(C.c_double)]),
    "hct_prof_read_bytes": (c_int, [c_int, C.POINTER(C.c_double)]),
    "hct_prof_shapes": (c_int, [c_int, c_void_p, c_int]),
    "hct_mae_plan_create": (c_void_p, [C.POINTER(MaeConfig), c_int, c_int]),
    "hct_mae_plan_destroy": (None, [c_void_p]),
    "hct_mae_plan_num_params": (c_int, [c_void_p]),
    "hct_mae_plan_param_info": (c_int, [c_void_p, c_int, C.POINTER(ParamInfo)]),
    "hct_mae_plan_param_elems": (c_int64, [c_void_p]),
    "hct_mae_plan_bf16_t_elems": (c_int64, [c_void_p]),
    "hct_mae_plan_workspace_bytes": (c_size_t, [c_void_p]),
    "hct_mae_plan_len_keep": (c_int, [c_void_p]),
    "hct_mae_plan_set_tail": (c_int, [c_void_p, c_int]),
    "hct_mae_plan_set_dec0": (c_int, [c_void_p, c_int]),
    "hct_mae_backward_final_offset": (c_int64, [c_void_p]),
    "hct_mae_plan_set_wgrad_defer": (c_int, [c_void_p, c_int, c_int]),
    "hct_mae_plan_bind": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "hct_mae_refresh_weights": (c_int, [c_void_p, c_int, c_void_p]),
    "hct_mae_forward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float, c_void_p]),
    "hct_mae_set_loss_grad": (c_int, [c_void_p, c_void_p]),
    "hct_mae_num_backward_stages": (c_int, [c_void_p]),
    "hct_mae_backward_stage_range": (c_int, [c_void_p, c_int, C.POINTER(c_int64), C.POINTER(c_int64)]),
    "hct_mae_backward_stage": (c_int, [c_void_p, c_int, c_void_p]),
    "hct_vit_forward": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "hct_vit_forward_parts": (c_int, [c_void_p, C.POINTER(c_void_p), c_int, c_int, c_void_p]),
    "hct_vit_backward_stage": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "hct_vit_assemble_bwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "hct_mae_plan_activation": (c_void_p, [c_void_p, C.c_char_p, C.POINTER(c_int64), C.POINTER(c_int64), C.POINTER(c_int)]),
}

_lib: Optional[C.CDLL] = None


class HctError(RuntimeError):
    pass


def exported_symbols():
    """Names every build of the library must export (checked by the CPU test-suite)."""
    return sorted(_PROTOS)


def load() -> C.CDLL:
    """Load libheadct_hip.so; raise loudly if it is absent (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HctError(
            f"{LIB_PATH} not found: build it with `python -m headct_foundation_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the MAE hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().hct_last_error_string()
        raise HctError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def ptr(t) -> Optional[int]:
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def stream_ptr() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
