"""ctypes binding of libtrace_hip.so (C ABI: include/trace_hip.h).  No CPU fallback: importing the
symbols fails loudly if the library has not been built (`python -m trace_amd.build`)."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libtrace_hip.so")
LIB_PATHS = {"bf16": LIB_PATH, "f16": os.path.join(HERE, "libtrace_hip_f16.so")}      # element type of the build -> library (trace_element_type)


class TraceConfigC(C.Structure):
    _fields_ = [
        ("vocab_size", C.c_int32), ("hidden_size", C.c_int32), ("intermediate_size", C.c_int32),
        ("num_layers", C.c_int32), ("num_heads", C.c_int32), ("num_kv_heads", C.c_int32),
        ("time_vocab", C.c_int32), ("score_vocab", C.c_int32),
        ("rms_eps", C.c_float), ("rope_theta", C.c_float),
        ("v_hidden", C.c_int32), ("v_inter", C.c_int32), ("v_layers_used", C.c_int32), ("v_heads", C.c_int32),
        ("v_image", C.c_int32), ("v_patch", C.c_int32),
        ("v_eps", C.c_float),
        ("num_slots", C.c_int32),
        ("slot_eps", C.c_float), ("slot_rope_base", C.c_float),
        ("max_frames", C.c_int32), ("max_ctx", C.c_int32), ("max_batch", C.c_int32), ("max_new_tokens", C.c_int32),
        ("projector_type", C.c_int32),
        ("vit_batch_frames", C.c_int32),
        ("llm_weights_fp8", C.c_int32),
    ]


# name -> (restype, argtypes); every symbol include/trace_hip.h declares
P, I, F = C.c_void_p, C.c_int, C.c_float
SIGNATURES = {
    "trace_last_error": (C.c_char_p, []),
    "trace_abi_version": (I, []),
    "trace_element_type": (I, []),
    "trace_ctx_create": (I, [C.POINTER(TraceConfigC), I, C.POINTER(P)]),
    "trace_ctx_destroy": (I, [P]),
    "trace_ctx_load_tensor": (I, [P, C.c_char_p, P, I, C.POINTER(C.c_int64), I]),
    "trace_ctx_finalize": (I, [P]),
    "trace_ctx_device_bytes": (C.c_int64, [P]),
    "trace_preprocess_frames": (I, [P, P, I, I, I, I, P, P, P, I, P]),
    "trace_vit_forward": (I, [P, P, I, I, P, P]),
    "trace_slot_pool": (I, [P, P, I, P, P]),
    "trace_stc_connector": (I, [P, P, I, P, C.POINTER(I), P]),
    "trace_encode_video": (I, [P, P, I, I, P, P, P]),
    "trace_encode_features": (I, [P, P, I, P, P, P]),
    "trace_splice_embeds": (I, [P, P, I, P, I, P, I, C.POINTER(I), P, P]),
    "trace_llm_prefill": (I, [P, I, P, I, P, P]),
    "trace_llm_prefill_pair": (I, [P, I, P, P, I, P]),
    "trace_llm_prefill_multi": (I, [P, I, C.POINTER(P), I, I, P]),
    "trace_llm_head_logits": (I, [P, P, I, I, P, P]),
    "trace_decode_begin": (I, [P, P, I, P, I, I, P, P, P]),
    "trace_decode_steps": (I, [P, I, I, P, P]),
    "trace_decode_read": (I, [P, P, P, P, P]),
    "trace_decode_host_mode": (I, [P, I]),
    "trace_decode_feed": (I, [P, P, I, P]),
    "trace_stream_create": (I, [P, I, I, C.POINTER(P)]),
    "trace_stream_destroy": (I, [P, P]),
    "trace_set_gemm_cus": (I, [P, I]),
    "trace_set_profile": (I, [P, I]),
    "trace_debug_buffers": (I, [P, C.POINTER(P), C.POINTER(P), C.POINTER(P), C.POINTER(C.c_int64)]),
    "trace_get_profile": (I, [P, P, I]),
    "trace_set_profile_brackets": (I, [P, I]),
    "trace_op_gemm": (I, [P, I, P, I, P, I, P, P, I, I, I, I, I, P]),
    "trace_op_set_gemm_variant": (I, [I]),
    "trace_op_layernorm": (I, [P, P, P, P, I, I, F, P]),
    "trace_op_rmsnorm": (I, [P, P, P, I, I, F, P]),
    "trace_op_attention": (I, [P, P, P, P, P, I, I, I, I, I, I, I, F, P]),
    "trace_op_skinny_gemm": (I, [P, P, P, P, I, I, I, I, I, P]),
    "trace_op_set_gemm_trace": (I, [P]),
    "trace_op_skinny_ks": (I, [I, I, I, I]),
    "trace_op_sk_rows": (I, []),
    "trace_op_skinny_fused_norm": (I, [P, I, P, P, P, F, P, P, I, I, I, P]),
    "trace_op_gemm_partial_ks": (I, [I, I]),
    "trace_op_gemm_partial": (I, [P, P, P, I, I, I, I, P]),
    "trace_op_gemm_swiglu_tiled": (I, [P, P, P, I, I, I, I, P]),
    "trace_op_tile_pack": (I, [P, P, I, I, P]),
    "trace_op_quant_rows_fp8": (I, [P, P, P, I, I, P]),
    "trace_op_gemm_fp8": (I, [P, P, P, P, P, P, I, I, I, I, P]),
    "trace_op_skinny_fp8": (I, [P, P, P, P, P, I, I, I, P]),
    "trace_op_skinny_w8": (I, [P, P, P, P, I, I, I, P]),
    "trace_op_swiglu_combine": (I, [P, I, I, P, I, P]),
    "trace_op_add_rmsnorm": (I, [P, I, P, P, P, P, I, I, F, P]),
    "trace_op_attn_decode": (I, [P, P, P, P, P, P, I, I, I, I, I, F, P]),
}

NOT_A_STATUS = {"trace_abi_version", "trace_element_type", "trace_op_skinny_ks", "trace_op_sk_rows", "trace_op_gemm_partial_ks"}   # ints that are values
_libs = {}


class TraceHipError(RuntimeError):
    pass


def load(element: str = "bf16"):
    """dlopen the library of one element type ("bf16": libtrace_hip.so, "f16": libtrace_hip_f16.so — the same sources compiled with -DTRACE_F16) and
    check every symbol include/trace_hip.h declares.  The two may be loaded side by side (RTLD_LOCAL)."""
    if element in _libs:
        return _libs[element]
    if element not in LIB_PATHS:
        raise ValueError(f"element type must be one of {sorted(LIB_PATHS)}, got {element!r}")
    # PyTorch-ROCm bundles its own libamdhip64; load it first so this library binds to the same HIP runtime
    # instance (two runtimes in one process cannot both own the device: "no ROCm-capable device is detected").
    import torch  # noqa: F401
    path = LIB_PATHS[element]
    if not os.path.exists(path):
        raise TraceHipError(
            f"{path} is missing: build it with `python -m trace_amd.build` (hipcc, gfx950). "
            "trace_amd has no CPU or PyTorch fallback for the hot path.")
    lib = C.CDLL(path)

    def errcheck(rc, fn, args):        # a negative status becomes an exception carrying THIS library's message (each .so has its own thread-local text)
        if rc < 0:
            msg = lib.trace_last_error()
            raise TraceHipError(f"{os.path.basename(path)} error {rc} in {fn.__name__}: {msg.decode() if msg else '?'}")
        return rc

    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
        if res is I and name not in NOT_A_STATUS:
            fn.errcheck = errcheck
    if lib.trace_abi_version() != 4:
        raise TraceHipError(f"{os.path.basename(path)} ABI version mismatch")
    if lib.trace_element_type() != {"bf16": 0, "f16": 1}[element]:
        raise TraceHipError(f"{os.path.basename(path)} was not built for {element} elements")
    _libs[element] = lib
    return lib


def element_of(dtype) -> str:
    """torch dtype of a 16-bit tensor -> the element type of the library that computes in it"""
    import torch
    if dtype == torch.bfloat16:
        return "bf16"
    if dtype == torch.float16:
        return "f16"
    raise ValueError(f"the engine computes in torch.bfloat16 or torch.float16, not {dtype}")


def check(rc: int) -> int:
    """Status of a C-ABI call.  Every status-returning entry point already raises through its library's own errcheck (load()), with the message of
    the .so that returned it; this stays as the call-site idiom and as a backstop for a status obtained some other way — it never dlopens anything."""
    if rc < 0:
        msgs = [m.decode() for m in (lib.trace_last_error() for lib in _libs.values()) if m]
        raise TraceHipError(f"libtrace_hip error {rc}: {' | '.join(msgs) if msgs else '?'}")
    return rc
