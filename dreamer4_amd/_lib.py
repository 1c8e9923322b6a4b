"""ctypes binding of libd4hip.so (C-ABI declared in include/d4hip.h).

The shared object is built in-tree by `dreamer4_amd.build.build()` (plain hipcc, no torch
headers).  Loading fails loudly when it is missing: there is no CPU / eager fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libd4hip.so')

D4_MAX_ACTION_TYPES = 8

GEMM_RMS_ROWSCALE, GEMM_SILU, GEMM_SWIGLU, GEMM_TRANS_A, GEMM_TRANS_B, GEMM_ACCUMULATE = 1, 2, 4, 8, 16, 32


class Config(C.Structure):
    _fields_ = [
        ('dim', C.c_int32), ('dim_latent', C.c_int32), ('num_latent_tokens', C.c_int32), ('depth', C.c_int32),
        ('time_block_every', C.c_int32), ('attn_heads', C.c_int32), ('attn_dim_head', C.c_int32),
        ('attn_softclamp_value', C.c_float),
        ('num_spatial_tokens', C.c_int32), ('num_register_tokens', C.c_int32), ('max_steps', C.c_int32),
        ('num_tasks', C.c_int32), ('num_discrete_action_types', C.c_int32),
        ('num_discrete_actions', C.c_int32 * D4_MAX_ACTION_TYPES), ('num_continuous_actions', C.c_int32),
        ('multi_token_pred_len', C.c_int32),
        ('policy_head_mlp_depth', C.c_int32), ('value_head_mlp_depth', C.c_int32),
        ('terminal_mlp_depth', C.c_int32), ('predict_terminals', C.c_int32),
        ('reward_num_bins', C.c_int32), ('value_num_bins', C.c_int32), ('reward_encoder_type', C.c_int32), ('matmul_bf16', C.c_int32), ('head_mlp_recipe', C.c_int32), ('continuous_beta_param', C.c_int32),
        ('pool_heads', C.c_int32), ('pool_dim_head', C.c_int32),
        ('gae_discount_factor', C.c_float), ('gae_lambda', C.c_float), ('ppo_eps_clip', C.c_float),
        ('policy_entropy_weight', C.c_float), ('use_delight_gating', C.c_int32),
        ('delight_temperature', C.c_float), ('pmpo_pos_to_neg_weight', C.c_float),
        ('pmpo_kl_div_loss_weight', C.c_float), ('pmpo_reverse_kl', C.c_int32),
        ('hl_gauss_sigma_to_bin_ratio', C.c_float), ('hl_gauss_eps', C.c_float),
        ('value_min', C.c_float), ('value_max', C.c_float),
        ('mode', C.c_int32), ('patch_size', C.c_int32), ('channels', C.c_int32), ('image_height', C.c_int32), ('image_width', C.c_int32),
        ('decoder_flow_steps', C.c_int32), ('decoder_pos_mlp_depth', C.c_int32),
        ('max_batch', C.c_int32), ('max_frames', C.c_int32), ('max_parallel_frames', C.c_int32),
        ('max_learn_rows', C.c_int32),
    ]


class RolloutIO(C.Structure):
    _fields_ = [
        ('batch', C.c_int32), ('time_steps', C.c_int32), ('prompt_frames', C.c_int32), ('num_steps', C.c_int32),
        ('use_time_cache', C.c_int32), ('sample_terminals', C.c_int32), ('sample_actions', C.c_int32),
        ('context_signal_noise', C.c_float), ('discrete_temperature', C.c_float), ('continuous_temperature', C.c_float),
        ('noise_latent', C.c_void_p), ('noise_context', C.c_void_p), ('gumbel_u', C.c_void_p), ('bern_u', C.c_void_p),
        ('beta_noise', C.c_void_p), ('tasks', C.c_void_p),
        ('latents', C.c_void_p), ('actions', C.c_void_p), ('actions_cont', C.c_void_p), ('rewards', C.c_void_p), ('ctx_hist', C.c_void_p),
        ('agent_embed', C.c_void_p), ('log_probs', C.c_void_p), ('log_probs_cont', C.c_void_p), ('cont_params', C.c_void_p),
        ('values', C.c_void_p), ('action_logits', C.c_void_p),
        ('lens', C.c_void_p), ('terminals', C.c_void_p),
    ]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p)


class LearnIO(C.Structure):
    _fields_ = [
        ('batch', C.c_int32), ('time', C.c_int32), ('objective', C.c_int32), ('normalize_advantages', C.c_int32),
        ('eps', C.c_float), ('use_delight_gating', C.c_int32), ('delight_temperature', C.c_float),
        ('agent_embed', C.c_void_p), ('actions', C.c_void_p), ('old_log_probs', C.c_void_p),
        ('actions_cont', C.c_void_p), ('old_log_probs_cont', C.c_void_p), ('old_cont_params', C.c_void_p), ('old_values', C.c_void_p),
        ('rewards', C.c_void_p), ('old_action_logits', C.c_void_p), ('lens', C.c_void_p), ('is_truncated', C.c_void_p),
        ('terminals', C.c_void_p),
        ('allreduce_sum', ALLREDUCE_FN), ('allreduce_user', C.c_void_p),
        ('losses', C.c_void_p), ('returns', C.c_void_p),
        ('d_agent_embed_policy', C.c_void_p), ('d_agent_embed_value', C.c_void_p),
    ]


# every symbol include/d4hip.h declares: (restype, argtypes)
_P, _I, _F, _L = C.c_void_p, C.c_int, C.c_float, C.c_int64
SYMBOLS = {
    'd4_last_error': (C.c_char_p, []),
    'd4_version': (_I, []),
    'd4_engine_create': (_I, [C.POINTER(Config), C.POINTER(_P)]),
    'd4_engine_destroy': (None, [_P]),
    'd4_engine_workspace_bytes': (C.c_size_t, [_P]),
    'd4_engine_set_workspace': (_I, [_P, _P, C.c_size_t]),
    'd4_engine_bind': (_I, [_P, C.c_char_p, _P, _P, _L]),
    'd4_engine_prepare': (_I, [_P, _P]),
    'd4_engine_cache_frames': (_I, [_P]),
    'd4_engine_cache_reset': (_I, [_P, _I]),
    'd4_engine_cache_export': (_I, [_P, _P, _I, _P]),
    'd4_engine_cache_import': (_I, [_P, _P, _I, _I, _P]),
    'd4_wm_forward': (_I, [_P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    'd4_decoder_forward': (_I, [_P, _P, _P, _I, _I, _I, _P, _P]),
    'd4_ff_workspace_bytes': (C.c_size_t, [_I, _I, _I]),
    'd4_ff_forward': (_I, [_P] * 6 + [_I, _I, _I, _P, _P, C.c_size_t, _P]),
    'd4_ff_backward': (_I, [_P] * 6 + [_I, _I, _I] + [_P] * 6 + [_P, C.c_size_t, _P]),
    'd4_ff_backward_saved': (_I, [_P] * 6 + [_I, _I, _I] + [_P] * 6 + [_P, C.c_size_t, _P]),
    'd4_attn_workspace_bytes': (C.c_size_t, [_I] * 5),
    'd4_space_attn_forward': (_I, [_P] * 11 + [_I] * 5 + [_F, _I, _I, _P, _P, C.c_size_t, _P]),
    'd4_space_attn_backward': (_I, [_P] * 12 + [_I] * 5 + [_F, _I, _I] + [_P] * 11 + [_P, C.c_size_t, _P]),
    'd4_space_attn_backward_saved': (_I, [_P] * 12 + [_I] * 5 + [_F, _I, _I] + [_P] * 11 + [_P, C.c_size_t, _P]),
    'd4_time_attn_workspace_bytes': (C.c_size_t, [_I] * 6),
    'd4_time_attn_forward': (_I, [_P] * 12 + [_I] * 6 + [_F, _I, _P, _P, C.c_size_t, _P]),
    'd4_time_attn_backward': (_I, [_P] * 13 + [_I] * 6 + [_F, _I] + [_P] * 11 + [_P, C.c_size_t, _P]),
    'd4_time_attn_backward_saved': (_I, [_P] * 13 + [_I] * 6 + [_F, _I] + [_P] * 11 + [_P, C.c_size_t, _P]),
    'd4_cross_attn_workspace_bytes': (C.c_size_t, [_I] * 7),
    'd4_cross_attn_forward': (_I, [_P] * 10 + [_I] * 8 + [_F, _P, _P, C.c_size_t, _P]),
    'd4_cross_attn_backward': (_I, [_P] * 11 + [_I] * 8 + [_F] + [_P] * 10 + [_P, C.c_size_t, _P]),
    'd4_cross_attn_backward_saved': (_I, [_P] * 11 + [_I] * 8 + [_F] + [_P] * 10 + [_P, C.c_size_t, _P]),
    'd4_encoder_forward': (_I, [_P, _P, _I, _I, _P, _P]),
    'd4_euler_step': (_I, [_P, _P, _L, _F, _F, _P]),
    'd4_rollout': (_I, [_P, C.POINTER(RolloutIO), _P]),
    'd4_learn': (_I, [_P, C.POINTER(LearnIO), _P]),
    'd4_adamw_clip': (_I, [_P, _P, _P, _P, _L, _I, _F, _F, _F, _F, _F, _F, _F, _P, _P]),
    'd4_profile_bf16_enable': (_I, [_I]),
    'd4_profile_bf16_read': (_I, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    'd4_profile_enable': (_I, [_I]),
    'd4_profile_read': (_I, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64), _I]),
    'd4_profile_classes': (_I, []),
    'd4_gemm_force_config': (_I, [_I]),
    'd4_debug_switch': (_I, [C.c_char_p, _I]),
    'd4_frame_fused_set': (_I, [_I]),
    'd4_profile_class_name': (C.c_char_p, [_I]),
    'd4_profile_glue_enable': (_I, [_I]),
    'd4_profile_glue_read': (_I, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64), _I]),
    'd4_profile_glue_classes': (_I, []),
    'd4_profile_glue_read_flops': (_I, [C.POINTER(C.c_double), _I]),
    'd4_measure_peaks': (_I, [_P, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), _P]),
    'd4_profile_glue_class_name': (C.c_char_p, [_I]),
    'd4_debug_buffer': (_I, [_P, C.c_char_p, C.POINTER(_P)]),
    'd4_gemm': (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    'd4_gemm_tn': (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, C.c_int64, _I, _I, _P]),
    'd4_gemm_pair': (_I, [_P, _I, _P, _P, _I, _I, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    'd4_gemm_batched': (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _F, _I, _L, _L, _L, _P]),
    'd4_gemm_bf16': (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    'd4_gemm_bf16a': (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P]),
    'd4_cvt_bf16': (_I, [_P, _P, _L, _P]),
    'd4_gemm_bf16a_compact': (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'd4_gemm_bf16a_batched': (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _L, _L, _L, _I, _P]),
    'd4_cvt_rows_bf16': (_I, [_P, _L, _P, _L, _I, _I, _P]),
    'd4_split_bf16x3': (_I, [_P, _P, _L, _L, _P]),
    'd4_gemm_split': (_I, [_P, _I, _P, _L, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P]),
    'd4_split_f16x2': (_I, [_P, _P, _I, _I, _I, _L, _P, _P]),
    'd4_row_scale_exp': (_I, [_P, _L, _I, _I, _P, _P]),
    'd4_gemm_split2': (_I, [_P, _I, _P, _L, _I, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P, _P]),
    'd4_rmsnorm': (_I, [_P, _I, _P, _P, _I, _I, _I, _F, _P]),
    'd4_rmsnorm_backward': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _F, _P]),
    'd4_hl_gauss_scalar': (_I, [_P, _I, _P, _P, _I, _I, _P]),
    'd4_ppo_policy_loss': (_I, [_P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P, _P, _P, _P]),
    'd4_gae': (_I, [_P, _P, _P, _P, _P, _F, _F, _I, _I, _P, _P]),
    'd4_categorical_sample_logp': (_I, [_P, _I, _P, _I, _P, _I, _I, _F, _P, _P, _P]),
    'd4_hl_gauss_ce': (_I, [_P, _I, _P, _P, _P, _I, _I, _F, _F, _F, _F, _I, _P, _P, _P, _P]),
}

_lib = None


class D4Error(RuntimeError):
    pass


def load():
    """Load libd4hip.so; raises when it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise D4Error(f'{LIB_PATH} is missing: run `python -m dreamer4_amd.build` (hipcc, gfx950). '
                      'There is no CPU fallback for the imagination path.')
    # tile choices of the benchmark configurations measured on MI355X, shipped with the package: preloaded by the GEMM dispatcher so
    # that those shapes never time anything at first use (reproducible performance, no first-call synchronisation)
    default_table = os.path.join(_HERE, 'gemm_tune_default.txt')
    if os.path.isfile(default_table):
        os.environ.setdefault('D4_GEMM_TUNE_DEFAULT', default_table)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().d4_last_error()
        raise D4Error(f'd4hip error {rc}: {msg.decode() if msg else "?"}')


def ptr(t):
    """Device (or host) pointer of a torch tensor, None -> NULL."""
    if t is None:
        return None
    assert t.is_contiguous(), 'd4hip expects contiguous tensors'
    return C.c_void_p(t.data_ptr())
