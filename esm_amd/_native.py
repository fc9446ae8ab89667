"""ctypes binding of libesmk.so (C ABI declared in include/esmk.h).

PyTorch is used only as the owner of device memory and streams: every call below takes raw
device pointers (``tensor.data_ptr()``) and the current HIP stream.  There is no CPU fallback:
if the shared library is missing or cannot be loaded the import of this module raises.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libesmk.so")

F32, F16, BF16 = 0, 1, 2
OUT_LOGITS, OUT_ATTN, OUT_CONTACTS, OUT_COL_ATTN, OUT_REPR_LOWP, OUT_ATTN_LOWP = 1, 2, 4, 8, 16, 32
EPI_STORE_T, EPI_STORE_F32, EPI_GELU_T, EPI_GELU_F32, EPI_RESID_F32 = 0, 1, 2, 3, 4


class EsmkConfig(ctypes.Structure):
    _fields_ = [
        ("num_layers", c_int32),
        ("embed_dim", c_int32),
        ("num_heads", c_int32),
        ("ffn_dim", c_int32),
        ("vocab", c_int32),
        ("pad_idx", c_int32),
        ("mask_idx", c_int32),
        ("cls_idx", c_int32),
        ("eos_idx", c_int32),
        ("token_dropout", c_int32),
        ("prepend_bos", c_int32),
        ("append_eos", c_int32),
        ("operand_dtype", c_int32),
        ("no_rope", c_int32),
        ("num_positions", c_int32),
        ("ln_before", c_int32),
        ("weight_split", c_int32),
        ("ln_fold", c_int32),
    ]


class EsmkMsaConfig(ctypes.Structure):
    _fields_ = [(n, c_int32) for n in (
        "num_layers", "embed_dim", "num_heads", "ffn_dim", "vocab", "pad_idx", "mask_idx", "cls_idx", "eos_idx",
        "prepend_bos", "append_eos", "num_positions", "has_msa_position_embedding", "operand_dtype", "weight_split")]


class EsmkProfileEntry(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 32), ("launches", c_int32), ("ms", ctypes.c_double),
                ("flops", ctypes.c_double), ("bytes", ctypes.c_double)]


# name -> (restype, argtypes); must list every symbol include/esmk.h declares
SIGNATURES = {
    "esmk_last_error": (c_char_p, []),
    "esmk_version": (c_char_p, []),
    "esmk_create": (c_int, [POINTER(EsmkConfig), POINTER(c_void_p)]),
    "esmk_destroy": (None, [c_void_p]),
    "esmk_set_rope_inv_freq": (c_int, [c_void_p, POINTER(c_float), c_int]),
    "esmk_packed_bytes": (c_int, [c_void_p, POINTER(c_size_t)]),
    "esmk_pack_weight": (
        c_int,
        [c_void_p, c_void_p, c_size_t, c_char_p, c_void_p, c_int, POINTER(c_int64), c_int, c_void_p],
    ),
    "esmk_workspace_bytes": (c_int, [c_void_p, c_int, c_int, c_uint32, POINTER(c_size_t)]),
    "esmk_forward": (
        c_int,
        [
            c_void_p, c_void_p, c_void_p, c_int, c_int,
            POINTER(c_int32), c_int, POINTER(c_void_p),
            c_uint32, c_void_p, c_void_p, c_void_p,
            c_void_p, c_size_t, c_void_p,
        ],
    ),
    "esmk_packed_workspace_bytes": (c_int, [c_void_p, c_int, c_int, c_uint32, POINTER(c_size_t)]),
    "esmk_forward_packed": (
        c_int,
        [
            c_void_p, c_void_p, c_void_p, POINTER(c_int32), c_int, c_int,
            POINTER(c_int32), c_int, POINTER(c_void_p),
            c_uint32, c_void_p, c_void_p, c_size_t, c_void_p,
        ],
    ),
    "esmk_msa_create": (c_int, [POINTER(EsmkMsaConfig), POINTER(c_void_p)]),
    "esmk_msa_workspace_bytes": (c_int, [c_void_p, c_int, c_int, c_int, c_uint32, POINTER(c_size_t)]),
    "esmk_msa_forward": (
        c_int,
        [
            c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
            POINTER(c_int32), c_int, POINTER(c_void_p),
            c_uint32, c_void_p, c_void_p, c_void_p, c_void_p,
            c_void_p, c_size_t, c_void_p,
        ],
    ),
    "esmk_profile_begin": (c_int, [c_void_p]),
    "esmk_profile_end": (c_int, [c_void_p, POINTER(EsmkProfileEntry), c_int, POINTER(c_int)]),
    "esmk_ln_fold_enabled": (c_int, [c_void_p]),
    "esmk_op_layernorm": (
        c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "esmk_op_masked_row_mean": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "esmk_op_linear": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    ),
    "esmk_debug_gemm_timing": (c_int, [c_void_p]),
    "esmk_debug_gemm_impl": (c_int, [c_int, c_int]),
    "esmk_debug_set": (c_int, [c_char_p, c_double]),
    "esmk_op_rowstats": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "esmk_op_ln_finalize": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "esmk_op_fold_weight": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "esmk_op_linear_ln": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                  c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "esmk_op_qkv_rope_ln": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_int, c_int, c_int, c_void_p]),
    "esmk_debug_mma_selftest": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "esmk_op_split_weight": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "esmk_op_linear_split": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "esmk_debug_linear_splitk": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "esmk_op_qkv_rope": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    ),
    "esmk_op_qkv_rope2": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    ),
    "esmk_op_attention": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
         c_void_p],
    ),
    "esmk_op_attention_probs": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
         c_void_p],
    ),
    "esmk_op_contacts": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
         c_int, c_int, c_void_p],
    ),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -m esm_amd.build` "
            "(needs hipcc; there is no CPU fallback for the engine)"
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class EsmkError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise EsmkError(lib.esmk_last_error().decode())


def dtype_code(torch_dtype):
    import torch

    return {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}[torch_dtype]


def torch_dtype(code):
    import torch

    return {F32: torch.float32, F16: torch.float16, BF16: torch.bfloat16}[code]


def cur_stream():
    import torch

    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)
