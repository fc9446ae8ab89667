"""q, k and v projections as ONE launch (gemm9 EPI_QKV_ALL, kernels.h) against the two launches (EPI_QKV_ROPE + EPI_V_T):
the same tiles computed by the same instruction sequences — every output must be bit-equal, for aligned and ragged
sequence lengths, both tile heights, padded and token-packed batches.  Reference: the three projections + rotary + head
split of esm/multihead_attention.py:256-284,354-355."""
import ctypes
import math
import os

import pytest
import torch

import esm
from esm_amd import _native as N
from esm_amd import ops
from esm_amd.synth import skip_param_init, synth_esm2_state_dict

pytestmark = pytest.mark.gpu


def knob(v):
    N.check(N.lib.esmk_debug_set(b"qkv_one_launch", ctypes.c_double(v)))


@pytest.fixture(autouse=True)
def _restore_knob():
    yield
    knob(-1)


@pytest.mark.parametrize("B,T,E,H", [(4, 1022, 1280, 20), (1, 1022, 1280, 20), (3, 160, 1280, 20), (2, 763, 1280, 20),
                                     (8, 256, 1280, 20), (16, 1022, 1280, 20), (5, 333, 640, 10), (2, 64, 2560, 40)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_one_launch_equals_two_launches_bit_for_bit(B, T, E, H, dtype):
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + T)
    a = (torch.randn(B * T, E, device="cuda", generator=g)).to(dtype)
    w = (torch.randn(3 * E, E, device="cuda", generator=g) / math.sqrt(E)).to(dtype)
    bias = 0.1 * torch.randn(3 * E, device="cuda", generator=g)
    qkv = ops.QkvHandle(E, H, operand_dtype=dtype)
    Tp = (T + 63) // 64 * 64
    outs = []
    for mode in (0, 1):
        knob(mode)
        q, k, vt = qkv(a, w, bias, B, T, log2_domain=True)
        outs.append((q.clone(), k.clone(), vt[..., :T].clone() if Tp != T else vt.clone()))
    for name, x, y in zip("q k vt".split(), outs[0], outs[1]):
        assert torch.isfinite(y.float()).all(), name
        assert torch.equal(x, y), (name, int((x != y).sum()))


def test_one_launch_is_taken_and_is_refused_where_it_does_not_apply():
    """The knob reaches the kernel choice: forced on, a shape the combined form does not support (E % 128 != 0) still
    runs (two launches); the C ABI rejects unknown knobs."""
    E, H, B, T = 320, 5, 2, 100
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn(B * T, E, device="cuda", generator=g).half()
    w = (torch.randn(3 * E, E, device="cuda", generator=g) / math.sqrt(E)).half()
    bias = 0.1 * torch.randn(3 * E, device="cuda", generator=g)
    qkv = ops.QkvHandle(E, H)
    knob(0)
    ref = [t.clone() for t in qkv(a, w, bias, B, T)]
    knob(1)
    got = qkv(a, w, bias, B, T)
    for x, y in zip(ref, got):
        assert torch.equal(x[..., :T] if x.shape[-1] != 64 else x, y[..., :T] if y.shape[-1] != 64 else y)
    assert N.lib.esmk_debug_set(b"no_such_knob", ctypes.c_double(1)) != 0


@pytest.mark.parametrize("B,T", [(4, 1022), (1, 1022), (2, 763), (8, 256)])
def test_layernorm_fold_form_one_launch_equals_two_launches(B, T):
    """The LayerNorm-fold consumer form (esmk_op_qkv_rope_ln: row scale + folded bias in the epilogues) through the one
    launch and through the two."""
    E, H = 1280, 20
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + T + 1)
    M = B * T
    x = torch.randn(M, E, device="cuda", generator=g) * 2 + 0.3
    gamma = 1 + 0.1 * torch.randn(E, device="cuda", generator=g)
    beta = 0.1 * torch.randn(E, device="cuda", generator=g)
    w = torch.randn(3 * E, E, device="cuda", generator=g) / math.sqrt(E)
    bias = 0.1 * torch.randn(3 * E, device="cuda", generator=g)
    wf = torch.zeros(3 * E, E, dtype=torch.float16, device="cuda")
    b2 = torch.empty(3 * E, device="cuda")
    N.check(N.lib.esmk_op_fold_weight(N.ptr(w), 0, N.ptr(gamma), N.ptr(beta), N.ptr(wf), 1, N.ptr(b2), 3 * E, E, E, N.cur_stream()))
    Mp = (M + 255) // 256 * 256
    y = torch.zeros(M, E, dtype=torch.float16, device="cuda")
    mean, rstd = torch.zeros(Mp, device="cuda"), torch.zeros(Mp, device="cuda")
    N.check(N.lib.esmk_op_rowstats(N.ptr(x), N.ptr(y), N.ptr(mean), N.ptr(rstd), M, E, E, 1, N.cur_stream()))
    qkv = ops.QkvHandle(E, H)
    Tp = (T + 63) // 64 * 64
    outs = []
    for mode in (0, 1):
        knob(mode)
        q = torch.empty((B, H, T, 64), dtype=torch.float16, device="cuda")
        k = torch.empty_like(q)
        vt = torch.zeros((B, H, 64, Tp), dtype=torch.float16, device="cuda")
        N.check(N.lib.esmk_op_qkv_rope_ln(qkv.h, N.ptr(y), N.ptr(wf), N.ptr(bias), N.ptr(b2), N.ptr(rstd), N.ptr(q), N.ptr(k),
                                          N.ptr(vt), B, T, 1, N.cur_stream()))
        outs.append((q, k, vt[..., :T].clone()))
    for name, a, b in zip("q k vt".split(), outs[0], outs[1]):
        assert torch.isfinite(b.float()).all(), name
        assert torch.equal(a, b), (name, int((a != b).sum()))


@pytest.mark.parametrize("fold", [False, True], ids=["plain", "ln_fold"])
@pytest.mark.parametrize("lens", [[1022, 1022, 1022, 1022], [150, 33, 97, 128, 64], [763, 336]])
def test_forward_bits_do_not_depend_on_the_launch_form(lens, fold, monkeypatch):
    """Whole forward (650M dims, 3 layers): logits, representations and contacts with the combined launch forced on ==
    forced off == the library's choice, padded and token-packed, plain and LayerNorm-fold engines."""
    monkeypatch.setenv("ESM_AMD_LN_FOLD", "1" if fold else "0")
    L, E, H = 3, 1280, 20
    sd = synth_esm2_state_dict(L, E, H, seed=11)
    with skip_param_init():
        model = esm.ESM2(L, E, H).eval()
    model.load_state_dict(sd)
    model = model.cuda()
    gen = torch.Generator().manual_seed(3)
    toks = torch.full((len(lens), max(lens)), 1, dtype=torch.int64)  # <pad>
    for b, n in enumerate(lens):  # <cls> residues <eos> <pad>... as BatchConverter yields it
        toks[b, 0], toks[b, n - 1] = 0, 2
        toks[b, 1:n - 1] = torch.randint(4, 24, (n - 2,), generator=gen)
    toks = toks.cuda()
    res = {}
    with torch.no_grad():
        for mode in (0, 1, -1):
            knob(mode)
            out = model(toks, repr_layers=[0, 1, L], return_contacts=True)
            pk = model.forward_varlen(toks, repr_layers=[L], min_saving=None)
            res[mode] = (out["logits"].clone(), out["representations"][L].clone(), out["representations"][1].clone(),
                         out["contacts"].clone(), pk["logits"].clone(), pk["representations"][L].clone())
    for mode in (1, -1):
        for i, (x, y) in enumerate(zip(res[0], res[mode])):
            assert torch.equal(x, y), (mode, i, float((x - y).abs().max()))
