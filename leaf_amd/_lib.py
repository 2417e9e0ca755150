"""ctypes binding of libleaf_hip.so (the C ABI in include/leaf_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C leaf_amd/csrc``.  There is no
fallback: if the shared object is missing, every product entry point raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LEAF_HIP_LIB") or os.path.join(_HERE, "libleaf_hip.so")   # override: diagnostic builds


class TextCfgC(C.Structure):
    _fields_ = [("layers", C.c_int32), ("width", C.c_int32), ("heads", C.c_int32), ("embed_dim", C.c_int32),
                ("context_length", C.c_int32), ("vocab_size", C.c_int32), ("activation", C.c_int32),
                ("ln_eps", C.c_float)]


_SIGS = {
    "leaf_last_error": (C.c_char_p, []),
    "leaf_version": (C.c_int, []),
    "leaf_text_create": (C.c_int, [C.POINTER(TextCfgC), C.c_int, C.POINTER(C.c_void_p)]),
    "leaf_text_destroy": (None, [C.c_void_p]),
    "leaf_text_set_chunk": (C.c_int, [C.c_void_p, C.c_int]),
    "leaf_text_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "leaf_text_get_option": (C.c_int, [C.c_void_p, C.c_char_p]),
    "leaf_text_set_grad_scaler": (C.c_int, [C.c_void_p, C.c_void_p]),
    "leaf_text_param_count": (C.c_size_t, [C.c_void_p]),
    "leaf_text_decay_count": (C.c_size_t, [C.c_void_p]),
    "leaf_text_num_tensors": (C.c_int, [C.c_void_p]),
    "leaf_text_param_info": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t),
                                       C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "leaf_text_w16_bytes": (C.c_size_t, [C.c_void_p]),
    "leaf_text_pack_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "leaf_text_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "leaf_text_stash_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "leaf_text_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "leaf_text_precise_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "leaf_text_forward_precise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                            C.c_void_p, C.c_size_t, C.c_void_p]),
    "leaf_score_candidates": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_size_t, C.c_void_p]),
    "leaf_text_kv_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "leaf_text_forward_kv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "leaf_score_candidates_prefix": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p,
                                               C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_size_t, C.c_void_p]),
    "leaf_score_candidates_prefix_fused": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                                     C.c_void_p]),
    "leaf_text_forward_train_delta": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                                C.c_size_t, C.c_void_p]),
    "leaf_textfare_input_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_size_t, C.c_void_p]),
    "leaf_pgd_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                               C.c_int, C.c_void_p]),
    "leaf_text_forward_train": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                          C.c_void_p]),
    "leaf_textfare_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "leaf_textfare_backward_events": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "leaf_clip_grads_inplace": (C.c_int, [C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "leaf_adamw_step_clip": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_float,
                                      C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float,
                                      C.c_void_p, C.c_void_p]),
    "leaf_adamw_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_float,
                                  C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_void_p]),
    "leaf_text_split_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "leaf_text_split_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "leaf_text_split_pack_masks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "leaf_tok_create": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "leaf_tok_destroy": (None, [C.c_void_p]),
    "leaf_tok_encode_batch": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_int]),
    "leaf_tok_mutate_encode": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "leaf_dict_create": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "leaf_dict_destroy": (None, [C.c_void_p]),
    "leaf_dict_size": (C.c_int64, [C.c_void_p]),
    "leaf_tok_constrain_ranges": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "leaf_punkt_create": (C.c_int, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                   C.POINTER(C.c_void_p)]),
    "leaf_punkt_destroy": (None, [C.c_void_p]),
    "leaf_punkt_set_strict": (C.c_int, [C.c_void_p, C.c_int]),
    "leaf_punkt_spans": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int32)]),
    "leaf_tok_constrain_punkt": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_char_p), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "leaf_tok_count_words": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int32)]),
    "leaf_tok_duplicate_map": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "leaf_tok_word_tokens": (C.c_int, [C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int)]),
    "leaf_tok_constrain": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_int]),
    "leaf_prof_begin": (C.c_int, []),
    "leaf_prof_pause": (C.c_int, [C.c_int]),
    "leaf_prof_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64),
                               C.c_int]),
    "leaf_prof_end_shapes": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                      C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int)]),
    "leaf_op_gemm": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                               C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "leaf_op_gemm_ld": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "leaf_op_gemm_awrap": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "leaf_op_gemm_resid_ln": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_int, C.c_int, C.c_void_p]),
    "leaf_op_gemm_resid_ln8": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_int, C.c_int, C.c_void_p]),
    "leaf_op_resid_pack": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "leaf_op_resid_unpack": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "leaf_op_ln_finalize": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "leaf_op_gemm_lnfold": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "leaf_op_qkv_attn": (C.c_int, [C.c_int] + [C.c_void_p] * 13 + [C.c_int] * 6 + [C.c_void_p]),
    "leaf_op_attention_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "leaf_op_layernorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int,
                                    C.c_int, C.c_void_p]),
    "leaf_op_attention_bwd_t": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_void_p]),
    "leaf_op_wgrad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_void_p, C.c_void_p]),
    "leaf_op_layernorm_bwd_ws_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "leaf_op_layernorm_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "leaf_op_project_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "leaf_op_attention_bwd": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p]),
}

# gradient-scaler state layout (include/leaf_hip.h LEAF_SC_*)
SC_COEF, SC_NORM, SC_SKIPPED, SC_BACKOFF, SC_GOOD, SC_SAT_FLAG, SC_SAT_STEPS, SC_INTERVAL, SC_APPLIED = range(9)
SC_WORDS = 16

EXPORTS = tuple(_SIGS)
# include/leaf_hip_diag.h: exported only by the diagnostic builds under tools/diag/ (make -C leaf_amd/csrc variants, stamps, ...);
# bound when the loaded library (LEAF_HIP_LIB) has them, absent from libleaf_hip.so
DIAG_SIGS = {
    "leaf_debug_qkv_attn_plan": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "leaf_debug_gemm_stamps": (C.c_int, [C.c_void_p]),
    "leaf_debug_gemm_min_tiles": (C.c_int, [C.c_int]),
}
VARIANTS_LIB = os.path.join(os.path.dirname(_HERE), "tools", "diag", "libleaf_hip_variants.so")
_lib = None


class LeafHipError(RuntimeError):
    pass


def lib():
    """Load libleaf_hip.so once; raise loudly when it is missing (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LeafHipError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                               f"g.build()'` or `make -C leaf_amd/csrc` (the HIP extension is mandatory)")
        l = C.CDLL(LIB_PATH)
        _bind(l, _SIGS)
        _bind(l, DIAG_SIGS, optional=True)
        _lib = l
    return _lib


def _bind(l, sigs, optional=False):
    for name, (res, args) in sigs.items():
        fn = getattr(l, name, None)
        if fn is None:
            if optional:
                continue
            raise LeafHipError(f"{name} missing from {LIB_PATH}")
        fn.restype, fn.argtypes = res, args


def diag_lib():
    """The diagnostic build with the kept experiments and the leaf_debug_* exports (make -C leaf_amd/csrc variants), loaded beside
    the product library: host-side hooks only (the tile-plan test); GPU tools select it for the whole process with LEAF_HIP_LIB."""
    if not os.path.exists(VARIANTS_LIB):
        raise LeafHipError(f"{VARIANTS_LIB} not found: make -C leaf_amd/csrc variants (or __graft_entry__.build())")
    l = C.CDLL(VARIANTS_LIB)
    _bind(l, DIAG_SIGS)
    return l


def check(rc: int, what: str):
    if rc != 0:
        raise LeafHipError(f"{what} failed: {lib().leaf_last_error().decode()}")
