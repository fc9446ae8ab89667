"""The attention kernel with 64 query rows per wave (csrc/attention_w64.hip) against the 32-row kernel (attention.hip):
a 32-row block of the new kernel performs exactly the operations of a wave of the old one, in the same order, so the
context rows and the saved log-sum-exp must be BIT-equal — plain and padded batches, ragged lengths, masked first tiles,
spiked keys (the exact path), both operand dtypes, and a whole model forward on a padded batch (the seq_info path with
skipped all-pad tiles).  Reference: esm/multihead_attention.py:357-394."""
import ctypes

import pytest
import torch

import esm
from esm_amd import _native as N
from esm_amd import ops
from esm_amd.synth import synth_esm2_state_dict, synth_tokens

pytestmark = pytest.mark.gpu


def knob(v):
    N.check(N.lib.esmk_debug_set(b"attn_w64", ctypes.c_double(v)))


@pytest.fixture(autouse=True)
def _restore_knob():
    yield
    knob(-1)  # the library's default


def _qkv(B, H, T, dt, seed, scale=0.6):
    g = torch.Generator(device="cuda").manual_seed(seed)
    qk, _ = ops.to_log2_domain(torch.randn(B, H, T, 64, device="cuda", generator=g) * scale, dt)
    k = (torch.randn(B, H, T, 64, device="cuda", generator=g) * 0.6).to(dt)
    v = torch.randn(B, H, T, 64, device="cuda", generator=g).to(dt)
    return qk, k, v


@pytest.mark.parametrize("B,H,T,pad", [(2, 3, 1024, 0), (2, 8, 1022, 0), (3, 2, 257, 60), (1, 1, 256, 0), (2, 2, 300, 5),
                                       (1, 9, 513, 200), (2, 2, 767, 0)])
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_w64_equals_the_32_row_kernel_bit_for_bit(B, H, T, pad, dt):
    qk, k, v = _qkv(B, H, T, dt, seed=T + B)
    key_bias = None
    if pad:
        key_bias = torch.zeros(B, T, device="cuda")
        key_bias[0, T - pad:] = float("-inf")
        if B > 1:
            key_bias[1, 3] = float("-inf")
    vt = ops.make_vt(v)
    knob(0)
    ctx0, lse0 = ops.attention(qk, k, vt, key_bias, want_lse=True)
    knob(1)
    ctx1, lse1 = ops.attention(qk, k, vt, key_bias, want_lse=True)
    assert torch.isfinite(ctx1.float()).all()
    assert torch.equal(ctx0, ctx1), int((ctx0 != ctx1).sum())
    assert torch.equal(lse0, lse1)


def test_w64_exact_path_cases():
    """The lazy offset's exact path inside ONE of a wave's two blocks: a spiked key late in the sweep (rows 17 and 300 sit
    in different blocks / waves), a first key tile that is entirely masked, scores of +-100."""
    g = torch.Generator(device="cuda").manual_seed(5)
    B, H, T = 2, 2, 512
    q = torch.randn(B, H, T, 64, device="cuda", generator=g) * 0.3
    k = torch.randn(B, H, T, 64, device="cuda", generator=g) * 0.3
    k[0, 0, 400] = q[0, 0, 17] * 40.0
    k[0, 0, 130] = q[0, 0, 300] * 25.0
    q[1] *= 12.0
    k[1, :, :128] *= 0.02
    (qk, _), k = ops.to_log2_domain(q, torch.float16), k.half()
    v = torch.randn(B, H, T, 64, device="cuda", generator=g).half()
    key_bias = torch.zeros(B, T, device="cuda")
    key_bias[1, :70] = float("-inf")
    vt = ops.make_vt(v)
    knob(0)
    ctx0, lse0 = ops.attention(qk, k, vt, key_bias, want_lse=True)
    knob(1)
    ctx1, lse1 = ops.attention(qk, k, vt, key_bias, want_lse=True)
    assert torch.isfinite(ctx1.float()).all()
    assert torch.equal(ctx0, ctx1) and torch.equal(lse0, lse1)


def test_w64_whole_forward_on_a_padded_batch():
    """esmk_forward with the knob on: representations, logits and contacts of a right-padded batch (seq_info path: masked
    tails, skipped all-pad key tiles, padding-only query blocks) equal the 32-row kernel's bit for bit."""
    L, E, H = 3, 256, 4
    sd = synth_esm2_state_dict(L, E, H, seed=3)
    model = esm.ESM2(L, E, H).eval()
    model.load_state_dict(sd)
    model = model.to("cuda")
    toks = synth_tokens(4, 640, seed=9)
    for i, n in enumerate((642, 300, 517, 131)):
        if n < toks.shape[1]:
            toks[i, n - 1] = 2
            toks[i, n:] = 1
    toks = toks.cuda()
    outs = []
    for m in (0, 1):
        knob(m)
        with torch.no_grad():
            outs.append(model(toks, repr_layers=[L], return_contacts=True))
    nonpad = toks.ne(1)
    assert torch.equal(outs[0]["representations"][L][nonpad], outs[1]["representations"][L][nonpad])
    assert torch.equal(outs[0]["logits"][nonpad], outs[1]["logits"][nonpad])
    assert torch.equal(outs[0]["contacts"], outs[1]["contacts"])
