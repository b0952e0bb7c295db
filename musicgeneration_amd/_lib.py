"""ctypes binding of libmgx.so (include/mgx.h).  No fallback: if the library is
missing the import fails loudly — the product path never runs without the HIP kernels."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MGX_LIB") or os.path.join(_HERE, "libmgx.so")    # MGX_LIB: A/B builds during tuning

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build the HIP kernels first (python -m musicgeneration_amd.build). "
        "musicgeneration_amd has no CPU or eager fallback.")

lib = C.CDLL(LIB_PATH)

F32, BF16 = 0, 1
GENERIC_KERNELS = 0x100        # OR-ed into an attention call's dtype: generic kernels (tests)
ACT_GELU, ACT_RELU = 0, 1

_vp, _i, _l, _f, _u64, _ll = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_ulonglong, C.c_longlong

PROTOTYPES = {
    "mgx_version": (C.c_int, []),
    "mgx_last_error_string": (C.c_char_p, []),
    "mgx_alibi_attn_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "mgx_alibi_attn_bwd_workspace_floats": (_l, [_i, _i, _i]),
    "mgx_alibi_attn_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "mgx_rel_attn_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "mgx_rel_attn_bwd_workspace_floats": (_l, [_i, _i, _i, _i, _i, _i]),
    "mgx_rel_attn_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "mgx_ln_residual_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _f, _f, _u64, _i, _vp]),
    "mgx_ln_bwd_partial_rows": (_i, []),
    "mgx_ln_residual_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _f, _u64, _i, _vp]),
    "mgx_bias_act_fwd": (_i, [_vp, _vp, _vp, _vp, _l, _i, _i, _f, _u64, _i, _vp]),
    "mgx_bias_act_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _l, _i, _i, _f, _u64, _i, _vp]),
    "mgx_xent_sums_len": (_i, []),
    "mgx_softmax_xent_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _ll, _i, _vp]),
    "mgx_softmax_xent_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _ll, _i, _vp]),
    "mgx_decode_attn": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "mgx_kv_append": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "mgx_decode_attn_at": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "mgx_kv_append_at": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "mgx_decode_step_attn": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "mgx_adamw": (_i, [_vp, _vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _f, _i, _f, _i, _vp]),
    "mgx_skinny_gemm_workspace_floats": (C.c_long, [_i, _i, _i]),
    "mgx_skinny_gemm_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mgx_rope_qk": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "mgx_unified_decode_events": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "mgx_separated_decode_events": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "mgx_unified_encode_tokens": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "mgx_adamw_multi": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f, _i, _f, _vp]),
    "mgx_decode_gather_embed": (_i, [_vp, _i, _l, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "mgx_decode_head_argmax": (_i, [_vp, _i, _i, _i, _vp, _vp, _f, _vp, _vp, _i, _vp, _vp, _vp]),
    "mgx_decode_commit": (_i, [_vp, _i, _l, _i, _vp, _vp, _vp]),
    "mgx_sample_gather_embed": (_i, [_vp, _i, _l, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "mgx_sample_head": (_i, [_vp, _i, _i, _i, _vp, _vp, _f, _vp, _vp, _i, _vp, _i, _u64, _vp, _vp, _vp, _vp]),
    "mgx_sample_commit": (_i, [_vp, _i, _l, _i, _i, _vp, _vp, _vp]),
    "mgx_embed_rows_bf16": (_i, [_vp, _vp, _i, _i, _l, _vp, _vp]),
    "mgx_embed_cat_bf16": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "mgx_colsum_workspace_floats": (_l, [_i]),
    "mgx_colsum_bf16": (_i, [_vp, _vp, _vp, _l, _i, _vp]),
    "mgx_colsum_static_max_n": (_i, []),
    "mgx_colsum_bf16_static": (_i, [_vp, _vp, _l, _i, _vp]),
    "mgx_cast_shadow_bf16": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mgx_gemm_nt_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _u64, _vp]),
    "mgx_gemm_nt_dfac_colpart_floats": (_l, [_i, _i]),
    "mgx_gemm_nt_dfac_colpart_rows": (_i, [_i, _i]),
    "mgx_gemm_nt_dfac_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "mgx_gemm_generic": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "mgx_gemm_tn_workspace_floats": (_l, [_i, _i, _i, _vp]),
    "mgx_gemm_tn_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "mgx_gemm_tn_pair_workspace_floats": (_l, [_i, _i, _i, _i, _i]),
    "mgx_gemm_tn_pair_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp]),
}

for _name, (_res, _args) in PROTOTYPES.items():
    _fn = getattr(lib, _name)       # AttributeError here = header and library out of sync
    _fn.restype = _res
    _fn.argtypes = _args


class MgxError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        raise MgxError(f"{what} failed ({rc}): {lib.mgx_last_error_string().decode()}")
