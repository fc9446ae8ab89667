"""Argument validation of the C ABI (include/esmk.h) without a GPU: every check below fails BEFORE the library touches
the HIP runtime (dimension checks, missing output buffers for the requested flags, workspace size, the row limit, the
segment table of esmk_forward_packed), so the error codes and messages are testable in the build container.  Buffers
are fake non-null addresses: a call that got past validation would dereference nothing on the host either — it would
fail in hipMalloc / the first launch — but none of these calls gets that far."""
import ctypes

import pytest
import torch

from esm_amd import _native as N

FAKE = ctypes.c_void_p(0x1000)


def make(L=2, E=128, H=2, **kw):
    cfg = N.EsmkConfig(L, E, H, 4 * E, 33, 1, 32, 0, 2, 1, 1, 1, N.dtype_code(torch.float16), 0, 0, 0)
    for k, v in kw.items():
        setattr(cfg, k, v)
    h = ctypes.c_void_p()
    rc = N.lib.esmk_create(ctypes.byref(cfg), ctypes.byref(h))
    return rc, h


def err():
    return N.lib.esmk_last_error().decode()


def test_create_rejects_bad_dimensions():
    for kw, msg in ((dict(num_layers=0), "non-positive"), (dict(num_heads=3), "divisible"),
                    (dict(embed_dim=144, num_heads=1), "head_dim"),          # head_dim 144: not 128, not <= 64
                    (dict(operand_dtype=0), "operand_dtype"), (dict(ffn_dim=100), "multiple")):
        rc, _ = make(**kw)
        assert rc != 0 and msg in err(), (kw, err())
    rc, h = make()
    assert rc == 0
    N.lib.esmk_destroy(h)
    N.lib.esmk_destroy(None)  # like free(NULL)


def test_workspace_queries_and_row_limit():
    rc, h = make()
    n = ctypes.c_size_t()
    assert N.lib.esmk_workspace_bytes(h, 2, 64, N.OUT_LOGITS, ctypes.byref(n)) == 0
    base = n.value
    assert N.lib.esmk_workspace_bytes(h, 2, 64, N.OUT_LOGITS | N.OUT_ATTN | N.OUT_CONTACTS, ctypes.byref(n)) == 0
    materialised = n.value
    assert N.lib.esmk_workspace_bytes(h, 2, 64, N.OUT_CONTACTS, ctypes.byref(n)) == 0
    fused = n.value  # [T,T] accumulators + per-channel sums instead of the [L,H,T,T] scratch of the caller's tensor
    assert base < materialised and base < fused
    assert N.lib.esmk_workspace_bytes(h, 0, 64, N.OUT_LOGITS, ctypes.byref(n)) != 0 and "positive" in err()
    assert N.lib.esmk_workspace_bytes(h, 1 << 14, 1 << 11, N.OUT_LOGITS, ctypes.byref(n)) != 0 and "2^24" in err()
    assert N.lib.esmk_packed_workspace_bytes(h, 2, 100, N.OUT_LOGITS, ctypes.byref(n)) != 0  # rows % 64
    assert N.lib.esmk_packed_workspace_bytes(h, 2, 128, N.OUT_ATTN, ctypes.byref(n)) != 0
    assert N.lib.esmk_packed_workspace_bytes(h, 2, 128, N.OUT_LOGITS | N.OUT_REPR_LOWP, ctypes.byref(n)) == 0
    N.lib.esmk_destroy(h)


def test_forward_argument_checks():
    rc, h = make()
    layers = (ctypes.c_int32 * 1)(2)
    outs = (ctypes.c_void_p * 1)(0x2000)

    def call(flags, logits=FAKE, attn=None, contacts=None, ws_bytes=1 << 40, B=2, T=16, n_repr=1, lay=layers, out=outs,
             tokens=FAKE, packed=FAKE, ws=FAKE):
        return N.lib.esmk_forward(h, packed, tokens, B, T, lay, n_repr, out, flags, logits, attn, contacts, ws,
                                  ctypes.c_size_t(ws_bytes), None)

    assert call(N.OUT_LOGITS, tokens=None) != 0 and "null" in err()
    assert call(N.OUT_LOGITS, B=0) != 0 and "positive" in err()
    assert call(N.OUT_LOGITS, logits=None) != 0 and "logits buffer" in err()
    assert call(N.OUT_LOGITS | N.OUT_ATTN) != 0 and "attention buffer" in err()
    assert call(N.OUT_LOGITS | N.OUT_CONTACTS) != 0 and "contacts buffer" in err()
    assert call(N.OUT_LOGITS | N.OUT_ATTN | N.OUT_CONTACTS | N.OUT_ATTN_LOWP, attn=FAKE, contacts=FAKE) != 0 \
        and "ESMK_OUT_ATTN_LOWP" in err()
    bad_layer = (ctypes.c_int32 * 1)(3)  # L = 2
    assert call(N.OUT_LOGITS, lay=bad_layer) != 0 and "repr layer" in err()
    assert call(N.OUT_LOGITS, ws_bytes=16) != 0 and "workspace too small" in err()
    N.lib.esmk_destroy(h)


def test_packed_segment_table_checks():
    rc, h = make()
    layers = (ctypes.c_int32 * 1)(2)
    outs = (ctypes.c_void_p * 1)(0x2000)

    def call(seg, rows=128, flags=N.OUT_LOGITS):
        arr = (ctypes.c_int32 * len(seg))(*seg)
        return N.lib.esmk_forward_packed(h, FAKE, FAKE, arr, len(seg) // 2, rows, layers, 1, outs, flags, FAKE, FAKE,
                                         ctypes.c_size_t(16), None)

    assert call([0, 20, 32, 5], rows=100) != 0 and "multiple of 64" in err()
    assert call([0, 0]) != 0 and "empty segment" in err()
    assert call([0, 20, 24, 5]) != 0 and "multiples of 16" in err()
    assert call([16, 20]) != 0 and "start at row 0" in err()
    assert call([0, 40, 32, 5]) != 0 and "disjoint" in err()
    assert call([0, 20, 112, 30]) != 0 and "past the last row" in err()
    assert call([0, 20, 32, 5], flags=N.OUT_LOGITS | N.OUT_CONTACTS) != 0 and "padded batches" in err()
    assert call([0, 20, 32, 5]) != 0 and "workspace too small" in err()  # a valid table reaches the size check
    N.lib.esmk_destroy(h)


def test_msa_flag_check():
    cfg = N.EsmkMsaConfig(2, 128, 2, 256, 33, 1, 32, 0, 2, 1, 0, 1026, 1, N.dtype_code(torch.float16))
    h = ctypes.c_void_p()
    assert N.lib.esmk_msa_create(ctypes.byref(cfg), ctypes.byref(h)) == 0
    n = ctypes.c_size_t()
    assert N.lib.esmk_msa_workspace_bytes(h, 1, 4, 16, N.OUT_LOGITS, ctypes.byref(n)) == 0 and n.value > 0
    N.lib.esmk_destroy(h)


def test_handle_kind_mix_up_is_rejected():
    """An MSA handle passed to the ESM-2 entry points (and the other way round) is refused before anything is
    indexed: esmk_forward on an MSA handle used to walk the empty ESM-2 layer table."""
    cfg = N.EsmkMsaConfig(2, 128, 2, 256, 33, 1, 32, 0, 2, 1, 0, 1026, 1, N.dtype_code(torch.float16))
    hm = ctypes.c_void_p()
    assert N.lib.esmk_msa_create(ctypes.byref(cfg), ctypes.byref(hm)) == 0
    n = ctypes.c_size_t()
    assert N.lib.esmk_workspace_bytes(hm, 2, 64, N.OUT_LOGITS, ctypes.byref(n)) != 0 and "MSA handle" in err()
    assert N.lib.esmk_packed_workspace_bytes(hm, 2, 128, N.OUT_LOGITS, ctypes.byref(n)) != 0 and "ESM-2 handle" in err()
    layers = (ctypes.c_int32 * 1)(2)
    outs = (ctypes.c_void_p * 1)(0x2000)
    assert N.lib.esmk_forward(hm, FAKE, FAKE, 2, 16, layers, 1, outs, N.OUT_LOGITS, FAKE, None, None, FAKE,
                              ctypes.c_size_t(1 << 40), None) != 0 and "MSA handle" in err()
    N.lib.esmk_destroy(hm)
    rc, h = make()
    assert N.lib.esmk_msa_workspace_bytes(h, 1, 4, 16, N.OUT_LOGITS, ctypes.byref(n)) != 0 and "MSA model handle" in err()
    N.lib.esmk_destroy(h)


def test_documented_config_struct_matches_header_and_binding():
    """A stale document must fail CI (Weak-9): the `_Cfg` field list of INTEGRATION.md's ctypes snippet, the
    `esmk_config` members of include/esmk.h, the shipped reference-side stub and esm_amd._native.EsmkConfig are ONE list —
    a maintainer who copies the documented snippet must pass a struct of the size esmk_create reads."""
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    native = [n for n, _ in N.EsmkConfig._fields_]
    # include/esmk.h: the int32_t members of `typedef struct { ... } esmk_config;`
    hdr = open(os.path.join(root, "include", "esmk.h")).read()
    body = re.search(r"typedef struct[^{]*\{(.*?)\}\s*esmk_config\s*;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    header = []
    for decl in re.findall(r"int32_t\s+([^;]+);", body):
        header += [v.strip() for v in decl.split(",")]
    assert header == native, (header, native)
    # the names inside the `_Cfg` class of the python snippet
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    snippet = re.search(r"class _Cfg\(ctypes\.Structure\):.*?_fields_\s*=\s*\[.*?for n in \((.*?)\)\]", doc, re.S).group(1)
    documented = re.findall(r'"(\w+)"', snippet)
    assert documented == native, (documented, native)
    # ... and the constructor call of the snippet passes one value per field
    call = re.search(r"cfg = _Cfg\((.*?)\)\s*#", doc, re.S).group(1)
    depth, nargs = 0, 1
    for ch in call:
        depth += ch in "([" 
        depth -= ch in ")]"
        nargs += ch == "," and depth == 0
    assert nargs == len(native), (nargs, len(native))
    # the shipped reference-side stub
    stub = open(os.path.join(root, "examples", "reference_binding", "_esmk.py")).read()
    m = re.search(r"_fields_\s*=\s*\[(.*?)\]\s*\n", stub, re.S)
    assert re.findall(r'"(\w+)"', m.group(1)) == native
    assert ctypes.sizeof(N.EsmkConfig) == 4 * len(native) == 72


def test_ln_fold_validation_without_gpu():
    """esmk_config.ln_fold: refused together with weight_split and for head_dim 128; with the fold a q/k/v or fc1 weight
    packed before the layer's LayerNorm parameters fails before the library touches the device."""
    h = ctypes.c_void_p()
    f16 = N.dtype_code(torch.float16)
    cfg = N.EsmkConfig(1, 128, 2, 512, 33, 1, 32, 0, 2, 1, 1, 1, f16, 0, 0, 0, 1, 1)
    assert N.lib.esmk_create(ctypes.byref(cfg), ctypes.byref(h)) != 0 and "ln_fold" in err()
    cfg = N.EsmkConfig(1, 256, 2, 1024, 33, 1, 32, 0, 2, 1, 1, 1, f16, 0, 0, 0, 0, 1)  # head_dim 128
    assert N.lib.esmk_create(ctypes.byref(cfg), ctypes.byref(h)) != 0 and "ln_fold" in err()
    cfg = N.EsmkConfig(1, 128, 2, 512, 33, 1, 32, 0, 2, 1, 1, 1, f16, 0, 0, 0, 0, 1)
    assert N.lib.esmk_create(ctypes.byref(cfg), ctypes.byref(h)) == 0
    try:
        nb = ctypes.c_size_t()
        assert N.lib.esmk_packed_bytes(h, ctypes.byref(nb)) == 0
        shape = (ctypes.c_int64 * 2)(128, 128)
        for key in (b"layers.0.self_attn.q_proj.weight", b"layers.0.self_attn.v_proj.weight", b"layers.0.fc1.weight"):
            rc = N.lib.esmk_pack_weight(h, FAKE, nb.value, key, FAKE, 0, shape, 2, None)
            assert rc != 0 and "LayerNorm" in err(), key
    finally:
        N.lib.esmk_destroy(h)
    # the fold is off by request: the same keys are plain conversions (which would touch the device: not called here)
    cfg = N.EsmkConfig(1, 128, 2, 512, 33, 1, 32, 0, 2, 1, 1, 1, f16, 0, 0, 0, 0, -1)
    assert N.lib.esmk_create(ctypes.byref(cfg), ctypes.byref(h)) == 0
    N.lib.esmk_destroy(h)
