"""ctypes binding of ``libdia_hip.so`` (``include/dia_hip.h``).

The library is the product path.  There is no CPU fallback: if the shared object is missing or
does not export the expected ABI this module raises, and every call that returns a non-zero
status raises :class:`DiaHipError` carrying ``dia_last_error()``.
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DIA_HIP_LIB") or os.path.join(_HERE, "libdia_hip.so")   # override: experiments only

ABI_VERSION = 6
KV_F32, KV_BF16, KV_BF16X2 = 0, 1, 2
EPI_SCALE_STORE, EPI_RESID_EMIT, EPI_SWIGLU_EMIT, EPI_CROSSKV = 0, 1, 2, 3
ATTN_SELF, ATTN_CROSS, ATTN_ENC = 0, 1, 2

EXPORTS = (
    "dia_last_error", "dia_abi_version", "dia_device_count", "dia_set_tuning", "dia_get_tuning", "dia_has_experiments", "dia_gemm", "dia_gemm_timed", "dia_mlp_fused", "dia_mlp_fused_timed", "dia_engine_mlp_fused", "dia_attn", "dia_attn_scratch_floats", "dia_enc_kv_prep", "dia_enc_attn", "dia_dec_prefill_embed", "dia_dec_prefill_kv", "dia_dec_prefill_attn",
    "dia_embed_text", "dia_embed_tokens", "dia_sample", "dia_prefetch", "dia_engine_create", "dia_engine_destroy",
    "dia_engine_decode", "dia_engine_set_prefetch", "dia_engine_step_logits_only", "dia_engine_profile_step", "dia_engine_time_step", "dia_timed_kernel_name", "dia_engine_launches_per_step",
    "dia_seg_mlp", "dia_seg_workspace_bytes", "dia_seg_workspace_control_bytes", "dia_seg_slots", "dia_seg_supported", "dia_seg_error",
)


class DiaHipError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("a_plane_stride", C.c_int64), ("a_ktiles", C.c_int32), ("M", C.c_int32),
        ("W", C.c_void_p), ("KT", C.c_int32), ("nstrips", C.c_int32), ("epi", C.c_int32), ("nw", C.c_int32),
        ("ssq_in", C.c_void_p), ("ssq_in_n", C.c_int32), ("ssq_ld", C.c_int32), ("inv_d", C.c_float), ("eps", C.c_float),
        ("out", C.c_void_p), ("ldo", C.c_int32), ("spw", C.c_int32),
        ("gnext", C.c_void_p), ("P", C.c_void_p), ("p_plane_stride", C.c_int64), ("p_ktiles", C.c_int32), ("_pad1", C.c_int32),
        ("ssq_out", C.c_void_p),
        ("kc", C.c_void_p), ("vc", C.c_void_p), ("kv_dtype", C.c_int32), ("kv_heads", C.c_int32), ("kv_cap", C.c_int32),
        ("kv_batch_index", C.c_int32), ("cos_t", C.c_void_p), ("sin_t", C.c_void_p),
        ("cmap", C.c_void_p), ("strip_map", C.c_void_p),
        ("sk_scratch", C.c_void_p), ("sk_tickets", C.c_void_p), ("sk", C.c_int32), ("kv_vblocked", C.c_int32),
        ("row_b", C.c_void_p), ("seg_off", C.c_void_p), ("sk_scratch_floats", C.c_int64),
        ("sp_blocks", C.c_void_p), ("sp_toff", C.c_void_p),
        ("act_f32", C.c_int32), ("w_planes", C.c_int32), ("kv_plane_stride", C.c_int64),
        ("w_layout", C.c_int32), ("kv_layer_strips", C.c_int32), ("kv_layer_stride", C.c_int64),
    ]


class AttnArgs(C.Structure):
    _fields_ = [
        ("mode", C.c_int32), ("kv_dtype", C.c_int32), ("n_kv_heads", C.c_int32), ("group", C.c_int32),
        ("n_rows", C.c_int32), ("kv_cap", C.c_int32),
        ("q", C.c_void_p), ("ldq", C.c_int32), ("q_off", C.c_int32), ("k_off", C.c_int32), ("v_off", C.c_int32),
        ("kc", C.c_void_p), ("vc", C.c_void_p), ("cur", C.c_void_p), ("len", C.c_void_p),
        ("enc_len", C.c_int32), ("rope_rows", C.c_int32), ("cos_t", C.c_void_p), ("sin_t", C.c_void_p),
        ("P", C.c_void_p), ("p_plane_stride", C.c_int64), ("p_ktiles", C.c_int32), ("_pad1", C.c_int32),
        ("scratch", C.c_void_p), ("tickets", C.c_void_p), ("head_map", C.c_void_p),
        ("v_blocked", C.c_int32), ("act_f32", C.c_int32), ("kv_plane_stride", C.c_int64),
    ]


class EncAttnArgs(C.Structure):
    _fields_ = [
        ("qkv", C.c_void_p), ("ldq", C.c_int32), ("q_off", C.c_int32), ("k_off", C.c_int32), ("v_off", C.c_int32),
        ("heads", C.c_int32), ("rows", C.c_int32),
        ("row_b", C.c_void_p), ("seg_off", C.c_void_p), ("seg_len", C.c_void_p), ("cos_t", C.c_void_p), ("sin_t", C.c_void_p),
        ("kp", C.c_void_p), ("vp", C.c_void_p), ("P", C.c_void_p), ("p_plane_stride", C.c_int64), ("p_ktiles", C.c_int32),
        ("_pad0", C.c_int32),
    ]


class DecPrefillArgs(C.Structure):
    _fields_ = [
        ("row_seg", C.c_void_p), ("seg_off", C.c_void_p), ("seg_len", C.c_void_p), ("seg_row", C.c_void_p),
        ("rows", C.c_int32), ("_pad0", C.c_int32),
        ("tokens", C.c_void_p), ("T", C.c_int32), ("C", C.c_int32), ("V", C.c_int32), ("D", C.c_int32),
        ("emb", C.c_void_p), ("g", C.c_void_p), ("x", C.c_void_p),
        ("P", C.c_void_p), ("p_plane_stride", C.c_int64), ("p_ktiles", C.c_int32), ("ssq_ld", C.c_int32),
        ("ssq", C.c_void_p),
        ("q", C.c_void_p), ("ldq", C.c_int32), ("q_off", C.c_int32), ("k_off", C.c_int32), ("v_off", C.c_int32),
        ("q_heads", C.c_int32), ("kv_heads", C.c_int32), ("kv_cap", C.c_int32), ("causal", C.c_int32),
        ("kc", C.c_void_p), ("vc", C.c_void_p), ("cos_t", C.c_void_p), ("sin_t", C.c_void_p), ("text_len", C.c_void_p),
    ]


class EmbedArgs(C.Structure):
    _fields_ = [
        ("tokens", C.c_void_p), ("cur", C.c_void_p),
        ("B", C.c_int32), ("T", C.c_int32), ("C", C.c_int32), ("V", C.c_int32), ("D", C.c_int32), ("act_f32", C.c_int32),
        ("emb", C.c_void_p), ("g", C.c_void_p), ("x", C.c_void_p), ("P", C.c_void_p),
        ("p_plane_stride", C.c_int64), ("p_ktiles", C.c_int32), ("ssq_ld", C.c_int32), ("ssq", C.c_void_p),
        ("cmap", C.c_void_p),
    ]


class SampleArgs(C.Structure):
    _fields_ = [
        ("logits", C.c_void_p), ("ld_logits", C.c_int32), ("B", C.c_int32), ("T", C.c_int32), ("C", C.c_int32),
        ("V", C.c_int32), ("max_tokens", C.c_int32),
        ("cfg_scale", C.c_float), ("temperature", C.c_float), ("top_p", C.c_float), ("top_k", C.c_int32),
        ("eos", C.c_int32), ("pad", C.c_int32), ("bos", C.c_int32), ("max_delay", C.c_int32),
        ("ignore_eos", C.c_int32), ("teacher", C.c_int32),
        ("delay", C.c_void_p), ("noise", C.c_void_p), ("noise_steps", C.c_int32), ("_pad0", C.c_int32),
        ("tokens", C.c_void_p), ("pred", C.c_void_p), ("cur", C.c_void_p), ("fsm", C.c_void_p),
        ("first_step", C.c_void_p), ("embed", EmbedArgs),
    ]


class DecLayer(C.Structure):
    _fields_ = [
        ("w_qkv", C.c_void_p), ("w_o", C.c_void_p), ("w_cq", C.c_void_p), ("w_co", C.c_void_p),
        ("w_wi", C.c_void_p), ("w_wo", C.c_void_p), ("w_wo_diag", C.c_void_p),
        ("g_sa", C.c_void_p), ("g_ca", C.c_void_p), ("g_mlp", C.c_void_p),
        ("k_self", C.c_void_p), ("v_self", C.c_void_p), ("k_cross", C.c_void_p), ("v_cross", C.c_void_p),
        ("kt_qkv", C.c_int32), ("ns_qkv", C.c_int32), ("kt_o", C.c_int32), ("ns_o", C.c_int32),
        ("kt_cq", C.c_int32), ("ns_cq", C.c_int32), ("kt_co", C.c_int32), ("ns_co", C.c_int32),
        ("kt_wi", C.c_int32), ("ns_wi", C.c_int32), ("kt_wo", C.c_int32), ("ns_wo", C.c_int32),
        ("cmap_ca", C.c_void_p), ("cmap_mlp", C.c_void_p), ("cmap_next", C.c_void_p),
        ("smap_qkv", C.c_void_p), ("smap_cq", C.c_void_p), ("hmap_self", C.c_void_p), ("hmap_cross", C.c_void_p),
    ]


class EngineDesc(C.Structure):
    _fields_ = [
        ("n_layer", C.c_int32), ("D", C.c_int32), ("F", C.c_int32), ("q_heads", C.c_int32), ("kv_heads", C.c_int32),
        ("cq_heads", C.c_int32), ("C", C.c_int32), ("V", C.c_int32),
        ("B", C.c_int32), ("T", C.c_int32), ("S", C.c_int32), ("kv_dtype", C.c_int32),
        ("rows_pad", C.c_int32), ("ld_logits", C.c_int32), ("eps", C.c_float), ("v_blocked", C.c_int32),
        ("layers", C.POINTER(DecLayer)), ("w_logits", C.c_void_p), ("kt_logits", C.c_int32), ("ns_logits", C.c_int32),
        ("g_final", C.c_void_p),
        ("x", C.c_void_p), ("planes_x", C.c_void_p), ("planes_a", C.c_void_p), ("planes_h", C.c_void_p),
        ("ssq", C.c_void_p), ("qkv", C.c_void_p), ("qc", C.c_void_p), ("logits", C.c_void_p),
        ("cos_t", C.c_void_p), ("sin_t", C.c_void_p), ("text_len", C.c_void_p),
        ("sk_scratch", C.c_void_p), ("sk_tickets", C.c_void_p),
        ("attn_scratch", C.c_void_p), ("attn_tickets", C.c_void_p), ("sk_scratch_floats", C.c_int64), ("mlp_barrier", C.c_void_p),
        ("act_f32", C.c_int32), ("w_planes", C.c_int32),
        ("sample", SampleArgs),
        ("seg_w", C.POINTER(C.c_void_p)), ("seg_ws", C.c_void_p),
        ("kv_plane_self", C.c_int64), ("kv_plane_cross", C.c_int64),
    ]


class SegArgs(C.Structure):
    _fields_ = [
        ("a_in", C.c_void_p), ("a_ktiles", C.c_int32), ("M", C.c_int32),
        ("W", C.c_void_p), ("nslots", C.c_int32), ("has_qkv", C.c_int32), ("D", C.c_int32), ("F", C.c_int32),
        ("x", C.c_void_p), ("ldx", C.c_int32), ("_pad0", C.c_int32),
        ("g_mlp", C.c_void_p), ("g_next", C.c_void_p), ("qkv_out", C.c_void_p), ("ldq", C.c_int32), ("_pad1", C.c_int32),
        ("planes_x", C.c_void_p), ("xkt", C.c_int32), ("_pad2", C.c_int32), ("ssq", C.c_void_p), ("ssq_ld", C.c_int32), ("eps", C.c_float),
        ("ws", C.c_void_p), ("timeout_us", C.c_int32), ("_pad3", C.c_int32), ("stamps", C.c_void_p),
    ]


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load the HIP library once.  Raises (never falls back) when it is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise DiaHipError(
            f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` or "
            f"`make -C dia-tts-prune_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    # torch ships its own libamdhip64 / libhsa-runtime64; load it FIRST so that this library binds to
    # the same HIP runtime instance as the tensors whose pointers it receives (two runtimes in one
    # process do not share a device context)
    import torch  # noqa: F401

    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64.so not found
        raise DiaHipError(f"cannot load {LIB_PATH}: {e}") from e
    for name in EXPORTS:
        if not hasattr(L, name):
            raise DiaHipError(f"{LIB_PATH} does not export {name}; rebuild it")
    L.dia_last_error.restype = C.c_char_p
    L.dia_abi_version.restype = C.c_int
    if L.dia_abi_version() != ABI_VERSION:
        raise DiaHipError(f"ABI mismatch: library {L.dia_abi_version()} vs binding {ABI_VERSION}")
    L.dia_set_tuning.argtypes = [C.c_char_p, C.c_int]
    L.dia_get_tuning.argtypes = [C.c_char_p]
    L.dia_has_experiments.restype = C.c_int
    L.dia_gemm.argtypes = [C.POINTER(GemmArgs), C.c_void_p]
    L.dia_gemm_timed.argtypes = [C.POINTER(GemmArgs), C.c_void_p, C.POINTER(C.c_float)]
    L.dia_mlp_fused.argtypes = [C.POINTER(GemmArgs), C.POINTER(GemmArgs), C.c_void_p, C.c_void_p]
    L.dia_mlp_fused_timed.argtypes = [C.POINTER(GemmArgs), C.POINTER(GemmArgs), C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
    L.dia_engine_mlp_fused.argtypes = [C.c_void_p]
    L.dia_attn.argtypes = [C.POINTER(AttnArgs), C.c_void_p]
    L.dia_attn_scratch_floats.argtypes = [C.c_int, C.c_int, C.c_int]
    L.dia_enc_attn.argtypes = [C.POINTER(EncAttnArgs), C.c_void_p]
    for fn in (L.dia_dec_prefill_embed, L.dia_dec_prefill_kv, L.dia_dec_prefill_attn):
        fn.argtypes = [C.POINTER(DecPrefillArgs), C.c_void_p]
    L.dia_enc_kv_prep.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.dia_embed_text.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.dia_embed_tokens.argtypes = [C.POINTER(EmbedArgs), C.c_void_p]
    L.dia_sample.argtypes = [C.POINTER(SampleArgs), C.c_void_p]
    L.dia_prefetch.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    L.dia_engine_create.argtypes = [C.POINTER(EngineDesc), C.c_void_p, C.POINTER(C.c_void_p)]
    L.dia_engine_destroy.argtypes = [C.c_void_p]
    L.dia_engine_decode.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.dia_engine_step_logits_only.argtypes = [C.c_void_p]
    L.dia_engine_set_prefetch.argtypes = [C.c_void_p, C.c_int]
    L.dia_engine_launches_per_step.argtypes = [C.c_void_p]
    L.dia_engine_profile_step.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
    L.dia_engine_time_step.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int]
    L.dia_timed_kernel_name.argtypes = [C.c_int]
    L.dia_timed_kernel_name.restype = C.c_char_p
    L.dia_seg_mlp.argtypes = [C.POINTER(SegArgs), C.c_void_p]
    L.dia_seg_workspace_bytes.restype = C.c_int64
    L.dia_seg_workspace_control_bytes.restype = C.c_int64
    L.dia_seg_slots.argtypes = [C.c_int]
    L.dia_seg_supported.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    L.dia_seg_error.argtypes = [C.c_void_p, C.c_void_p]
    _lib = L
    return L


def set_tuning(name: str, value: int) -> None:
    """launch-heuristic override (csrc/tuning.hpp); value < 0 clears it"""
    check(lib().dia_set_tuning(name.encode(), int(value)), f"dia_set_tuning({name})")


def get_tuning(name: str) -> int:
    """current value of a launch-heuristic override, -1 = unset (DIA_TUNE is read on first use)"""
    return int(lib().dia_get_tuning(name.encode()))


def has_experiments() -> bool:
    return bool(lib().dia_has_experiments())


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().dia_last_error()
        raise DiaHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def ptr(t) -> int:
    """Raw device pointer of a torch tensor (or None -> NULL)."""
    return 0 if t is None else t.data_ptr()
