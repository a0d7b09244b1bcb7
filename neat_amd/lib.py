"""ctypes binding of libneat_hip.so (the C ABI declared in include/neat_hip.h).

There is no fallback: if the shared library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NEAT_HIP_LIB: an alternative build of the same library (kernel A/B experiments from tools/); default: the in-tree build
LIB_PATH = os.environ.get("NEAT_HIP_LIB") or os.path.join(_HERE, "libneat_hip.so")

EPI_BF16, EPI_BIAS_BF16, EPI_BIAS_RES_F32, EPI_BIAS_GELU, EPI_GELU_BWD, EPI_F32 = range(6)

_P, _I, _L, _F, _D = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_double

# name -> (restype, argtypes); must list every symbol of include/neat_hip.h (tests/test_abi.py checks)
SIGNATURES = {
    "neat_version": (C.c_char_p, []),
    "neat_last_error": (C.c_char_p, []),
    "neat_ln_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _P]),
    "neat_ln_bwd_blocks": (_I, [_I]),
    "neat_ln_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "neat_ln_bwd_finalize": (_I, [_P, _I, _I, _P, _P, _I, _P]),
    "neat_ln_bwd_finalize_batched": (_I, [_P, _I, _I, _I, _P, _I, _P]),
    "neat_pack_weight_bf16": (_I, [_P, _P, _I, _I, _I, _P]),
    "neat_gemm_bf16": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "neat_pack_weight_bf16_t256": (_I, [_P, _P, _I, _I, _I, _P]),
    "neat_gemm_bf16_t256": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "neat_gemm_prefers_t256": (_I, [_I, _I, _I, _I]),
    "neat_gemm_bf16_res_ln": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "neat_gemm_bf16_t256_gelu_rows": (_I, [_P, _P, _P, _P, _I, _P, _I, _I, _I, _P]),
    "neat_image_preprocess_ws_bytes": (_L, [_I, _I, _I]),
    "neat_image_preprocess": (_I, [_P, _I, _I, _P, _I, _F, _F, _F, _F, _F, _F, _P, _P]),
    "neat_image_preprocess_batch": (_I, [_P, _P, _I, _I, _I, _P, _I, _F, _F, _F, _F, _F, _F, _P, _P]),
    "neat_frames_preprocess_ws_bytes": (_L, [_I, _I, _I, _I]),
    "neat_frames_preprocess": (_I, [_P, _I, _I, _I, _P, _I, _F, _F, _F, _F, _F, _F, _P, _P]),
    "neat_attn_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "neat_attn_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "neat_pack_plan": (_I, [_P, _P, _P, _I, _I, _P]),
    "neat_embed_packed_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "neat_embed_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "neat_eot_index": (_I, [_P, _P, _I, _I, _P]),
    "neat_pool_proj_ws_floats": (_L, [_I, _I, _I]),
    "neat_pool_proj_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "neat_pool_proj_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "neat_patchify": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "neat_vision_embed": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "neat_frame_mean_norm": (_I, [_P, _P, _I, _I, _I, _P]),
    "neat_coarse_ws_floats": (_L, [_I, _I]),
    "neat_coarse_align": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "neat_coarse_align_i2t": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "neat_coarse_static_max_slots": (_L, []),
    "neat_coarse_static_init": (_I, []),
    "neat_coarse_align_static": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "neat_loss_ws_floats": (_L, [_I, _I]),
    "neat_tta_loss_fwd_bwd": (_I, [_P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "neat_adamw_flat": (_I, [_P, _P, _P, _P, _L, _D, _D, _D, _D, _D, _I, _P]),
    "neat_topk_ws_bytes": (_L, [_I, _I, _I]),
    "neat_topk_rows": (_I, [_P, _L, _L, _I, _I, _I, _P, _P, _P, _P]),
    "neat_recall_from_topk": (_I, [_P, _P, _L, _L, _I, _I, _I, _P, _P]),
    "neat_recall_owner_topk": (_I, [_P, _I, _I, _P, _P, _P, _P, _I, _P, _P]),
    "neat_best_positive_rank": (_I, [_P, _L, _L, _P, _L, _L, _I, _I, _P, _P]),
    "neat_best_positive_rank_owner": (_I, [_P, _L, _L, _P, _P, _I, _I, _P, _P, _P]),
    "neat_mcq_scores": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "neat_gemm_f32_nt": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "neat_sim_topk_ws_bytes": (_L, [_I, _I, _I]),
    "neat_sim_topk": (_I, [_P, _P, _I, _P, _P, _I, _I, _I, _P, _P]),
    "neat_comm_unique_id": (_I, [_P]),
    "neat_comm_init": (_I, [_P, _I, _I]),
    "neat_comm_world": (_I, []),
    "neat_comm_rank": (_I, []),
    "neat_comm_allgather": (_I, [_P, _P, _L, _P]),
    "neat_comm_allreduce_f32": (_I, [_P, _P, _L, _P]),
    "neat_comm_destroy": (_I, []),
    "neat_cast_bf16": (_I, [_P, _P, _L, _P]),
    "neat_transpose_cast_bf16": (_I, [_P, _P, _I, _I, _P]),
}

_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(or `make -C neat_amd/csrc`).  There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def last_error() -> str:
    return load().neat_last_error().decode()


def check(rc: int, name: str = ""):
    if rc != 0:
        raise RuntimeError(f"libneat_hip {name} failed ({rc}): {last_error()}")


def call(name: str, *args):
    """Call an int-returning entry point and raise on a negative status."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise RuntimeError(f"libneat_hip {name} failed ({rc}): {last_error()}")
