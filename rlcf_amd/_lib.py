"""ctypes binding of librlcf_hip.so (include/rlcf_hip.h).

The product path has no CPU fallback: if the HIP library is missing or fails to
load, importing this module raises.  `build()` compiles it in-tree with hipcc.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RLCF_LIB_PATH") or os.path.join(HERE, "librlcf_hip.so")     # (override: kernel experiments)
CSRC = os.path.join(HERE, "csrc")

PREC_F32, PREC_F16, PREC_F16X3 = 0, 1, 2       # F16: single-pass performance mode (not parity-grade); F16X3: the default
EPI_NONE, EPI_QUICKGELU, EPI_QUICKGELU_BWD = 0, 1, 2
TEXT_DENSE, TEXT_PACKED, TEXT_SHARED = 0, 1, 2
STUDENT, REWARD = 0, 1
MAX_REWARDS = 4
F_REWARD_PROCESS, F_AMPLIFY, F_PROCESS_BATCH, F_MIN_ENTROPY = 1, 2, 4, 8


class ClipCfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("embed_dim", "image_resolution", "vision_layers", "vision_width",
                                       "vision_patch_size", "context_length", "vocab_size", "text_width",
                                       "text_heads", "text_layers")] + [("vision_stages", C.c_int * 4)]


class AugmixOp(C.Structure):
    _fields_ = [("op", C.c_int), ("ip", C.c_int), ("c", C.c_double * 6)]


class HardAug(C.Structure):    # rlcf_hard_aug: the draws of the hard_aug recipe for one view (include/rlcf_hip.h)
    _fields_ = [("order", C.c_int * 4), ("b", C.c_float), ("c", C.c_float), ("s", C.c_float), ("hue", C.c_int), ("gray", C.c_int),
                ("blur", C.c_int), ("k", C.c_float * 9)]


class Crop(C.Structure):       # rlcf_crop: RandomResizedCrop box + RandomHorizontalFlip
    _fields_ = [(n, C.c_int) for n in ("top", "left", "h", "w", "flip")]


class Seq(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("q_start", "q_len", "pre_start", "pre_len")]


class TTAArgs(C.Structure):
    _fields_ = [("selection_p", C.c_float), ("tta_steps", C.c_int), ("sample_k", C.c_int),
                ("lr", C.c_float), ("weight_decay", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("eps", C.c_float), ("flags", C.c_int), ("clipscore_weight", C.c_float),
                ("min_entropy_w", C.c_float), ("sparse_backward", C.c_int), ("skip_final", C.c_int),
                ("ctx_in", C.c_void_p), ("n_sel", C.c_int)]


TTA_OUT_FIELDS = ("logits", "entropy", "selected_idx", "topk_idx", "clip_score", "rewards", "loss", "dlogits",
                  "ctx_grad", "ctx_after", "reward_image_features", "final_logits", "top5", "ln_grad", "ln_after", "vis_grad", "vis_after", "step_skipped")


class TTAOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in TTA_OUT_FIELDS]


P, I, F, I64, D = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_double

# name -> (restype, argtypes): every symbol include/rlcf_hip.h declares
SIGNATURES = {
    "rlcf_last_error": (C.c_char_p, []),
    "rlcf_version": (I, []),
    "rlcf_gemm_nt": (I, [P, I, P, I, P, P, I, P, I, P, I, I, I, I, F, I, I, P]),
    "rlcf_split_f16x2": (I, [P, P, P, I64, P]),
    "rlcf_conv3x3_nhwc_f16x3": (I, [P, P, P, P, P, I, I, I, I, I, I, P]),
    "rlcf_gemm_f16": (I, [P, I, P, I, P, P, I, P, I, P, I, I, I, I, F, I, P]),
    "rlcf_gemm_f16_ln": (I, [P, I, P, I, P, P, I, I, I, I, F, I, I, P, P, P, P]),
    "rlcf_ln_stats_final": (I, [P, I, I, I, P, P]),
    "rlcf_resid16_init": (I, [P, P, P, I, I, P]),
    "rlcf_gemm_f16x3": (I, [P, P, I, P, P, I, P, P, I, P, I, P, I, P, P, I, I, I, I, F, I, P]),
    "rlcf_layernorm_fwd": (I, [P, P, P, P, I, I, P]),
    "rlcf_layernorm_bwd": (I, [P, P, P, P, P, P, I, I, P]),
    "rlcf_attention_bwd_flash": (I, [P, P, P, P, P, I, I, I, I, P, P]),
    "rlcf_attention_bwd_flash_prec": (I, [P, P, P, P, P, I, I, I, I, P, I, P]),
    "rlcf_attention_fwd": (I, [P, P, I, I, I, I, P, P, I, P]),
    "rlcf_attention_fwd_pairs": (I, [P, P, I, I, I, P, P, P, I, P]),
    "rlcf_split_pairs": (I, [P, P, C.c_int64, I, P]),
    "rlcf_gemm_skinny": (I, [P, I, P, P, P, I, P, I, P, I, I, I, I, F, I, P, I, P]),
    "rlcf_attention_debug": (I, [I, I]),
    "rlcf_attention_bwd": (I, [P, P, P, I, I, I, I, P, P]),
    "rlcf_entropy_select": (I, [P, I, I, I, P, P, P]),
    "rlcf_reward_loss": (I, [P, I, P, I, I, I, P, P, I, F, I, F, P, P, P, P, P, P]),
    "rlcf_adamw_step": (I, [P, P, P, P, I64, I, F, F, F, F, F, P]),
    "rlcf_engine_create": (P, [C.POINTER(ClipCfg), C.POINTER(ClipCfg), I, I, I]),
    "rlcf_engine_destroy": (None, [P]),
    "rlcf_engine_load_weight": (I, [P, I, C.c_char_p, P, I64]),
    "rlcf_engine_finalize": (I, [P, P]),
    "rlcf_engine_set_class_bank": (I, [P, P, I, I, P, I, P]),
    "rlcf_engine_set_class_bank_ex": (I, [P, P, I, I, P, I, P, P, P]),
    "rlcf_encode_image": (I, [P, I, P, I, P, P]),
    "rlcf_encode_image_resized": (I, [P, I, P, I, I, P, P]),
    "rlcf_text_features": (I, [P, P, P, P]),
    "rlcf_reward_class_features": (I, [P, I, P, P]),
    "rlcf_make_views_scratch_bytes": (C.c_size_t, [I, I, I]),
    "rlcf_make_views": (I, [P, I, I, P, I, I, P, P, P, P, C.c_size_t, P]),
    "rlcf_make_views_augmix_scratch_bytes": (C.c_size_t, [I, I, I]),
    "rlcf_make_views_augmix": (I, [P, I, I, P, I, I, P, P, P, P, P, P, P, C.c_size_t, P]),
    "rlcf_make_views_hard_scratch_bytes": (C.c_size_t, [I, I, I]),
    "rlcf_make_views_hard": (I, [P, I, I, P, I, I, P, P, P, P, P, P, P, P, C.c_size_t, P]),
    "rlcf_tta_batch_ln": (I, [P, P, I, I, C.POINTER(TTAArgs), P, P, P]),
    "rlcf_engine_momentum_update": (I, [P, P, D, D, I, P]),
    "rlcf_engine_reset_visual_state": (I, [P, P]),
    "rlcf_engine_create_ensemble": (P, [C.POINTER(ClipCfg), C.POINTER(ClipCfg), I, I, I, I]),
    "rlcf_engine_set_reward_mix": (I, [P, P, I, I]),
    "rlcf_reward_loss_ensemble": (I, [P, I, P, I, I, I, I, P, P, P, P, I, F, I, F, P, P, P, P, P, P]),
    "rlcf_logits": (I, [P, P, I, P, I, P, P]),
    "rlcf_text_backward_dense": (I, [P, P, P, I, P, P, P]),
    "rlcf_tta_sample": (I, [P, P, I, C.POINTER(TTAArgs), C.POINTER(TTAOut), P]),
    "rlcf_tta_batch": (I, [P, P, I, I, C.POINTER(TTAArgs), P, P, P]),
    "rlcf_tta_sample_ln": (I, [P, P, I, C.POINTER(TTAArgs), C.POINTER(TTAOut), P]),
    "rlcf_tta_retrieval_image": (I, [P, P, I, C.POINTER(TTAArgs), C.POINTER(TTAOut), P]),
    "rlcf_tta_sample_visual": (I, [P, P, I, C.POINTER(TTAArgs), C.POINTER(TTAOut), P]),
    "rlcf_engine_set_image_bank": (I, [P, P, P, I, P]),
    "rlcf_tta_retrieval_text": (I, [P, P, P, P, P]),
    "rlcf_engine_text_param_count": (I64, [P, P, P]),
    "rlcf_engine_text_param_layout": (I, [P, P, P, I, P]),
    "rlcf_engine_get_text_params": (I, [P, P, P, I, P]),
    "rlcf_engine_momentum_update_text": (I, [P, P, P, D, D, I, P]),
    "rlcf_engine_visual_param_count": (I64, [P, P]),
    "rlcf_engine_visual_param_layout": (I, [P, P, P, I, P]),
    "rlcf_engine_get_visual_params": (I, [P, P, I, P]),
    "rlcf_engine_set_visual_params": (I, [P, P, P]),
    "rlcf_engine_momentum_update_visual": (I, [P, P, D, D, I, P]),
    "rlcf_engine_ln_param_count": (I, [P]),
    "rlcf_engine_f16_grid_weights": (I, [P, I, P]),
    "rlcf_avg_entropy": (I, [P, I, I, P, P]),
    "rlcf_accuracy": (I, [P, P, I, I, P, P, P]),
    "rlcf_engine_set_bn_prior_strength": (I, [P, I]),
    "rlcf_engine_set_side_stream": (I, [P, I]),
    "rlcf_engine_set_f16_lnfold": (I, [P, I]),
    "rlcf_lanes_create": (P, [P, I]),
    "rlcf_lanes_create_on": (P, [P, I, P]),
    "rlcf_lanes_destroy": (None, [P]),
    "rlcf_lanes_count": (I, [P]),
    "rlcf_lanes_stream": (P, [P, I]),
    "rlcf_lanes_submit": (I, [P, P, I, I, C.POINTER(TTAArgs), P, P, I, P]),
    "rlcf_lanes_join": (I, [P, P]),
    "rlcf_tta_lanes": (I, [P, I, P, I, I, C.POINTER(TTAArgs), P, P, I, P]),
    "rlcf_top5_hits": (I, [P, P, I, P, P]),
    "rlcf_engine_bn_stats_count": (I, [P]),
    "rlcf_engine_encode_image_bn": (I, [P, P, I, P, P]),
    "rlcf_engine_encode_image_bn_form": (I, [P, P, I, I, P, P]),
    "rlcf_engine_get_bn_stats": (I, [P, P, I, P]),
    "rlcf_engine_get_ln_params": (I, [P, P, I, P]),
    "rlcf_engine_set_ln_params": (I, [P, P, P]),
    "rlcf_engine_last_flops": (D, [P]),
    "rlcf_engine_text_rows": (I, [P]),
    "rlcf_profile_gemm": (I, [I]),
    "rlcf_profile_read": (I, [I, C.POINTER(I), C.POINTER(D), C.POINTER(D)]),
    "rlcf_profile_count": (I, []),
    "rlcf_profile_entry": (I, [I, C.POINTER(I), C.POINTER(D), C.POINTER(D), C.POINTER(I)]),
}


def build(verbose: bool = False) -> str:
    """Compile librlcf_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", CSRC, "-j", str(os.cpu_count() or 4)], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:], r.stderr[-4000:])
    if r.returncode != 0 or not os.path.exists(LIB_PATH):
        raise RuntimeError("building librlcf_hip.so failed:\n" + r.stderr[-4000:])
    return LIB_PATH


class RlcfError(RuntimeError):
    pass


_lib = None


def lib() -> C.CDLL:
    """The loaded library; raises (loudly) if it is absent — there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RlcfError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(the RLCF HIP path has no CPU fallback)")
        # PyTorch-ROCm ships its own libamdhip64; the library must bind to THAT runtime (the one that owns the device context and
        # the tensors whose pointers it receives), so torch and its HIP runtime are brought up before the dlopen.  Loaded the other
        # way round, the system runtime from /opt/rocm is pulled in first and sees no device next to torch's.
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        _lib = h
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise RlcfError(f"{what} failed ({rc}): {lib().rlcf_last_error().decode(errors='replace')}")
