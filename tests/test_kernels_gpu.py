"""Single-kernel parity tests on the MI355X: every HIP kernel of libesmk.so, called through the
C ABI (esm_amd.ops -> ctypes), against a plain PyTorch fp32 reference of the same op evaluated on
the SAME operand values (inputs are rounded to the operand dtype first, so the tolerance only has
to cover fp32 accumulation order and the final rounding of the output)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DT = [torch.float16, torch.bfloat16]


def _eps(dt):
    return 2.0 ** -11 if dt == torch.float16 else 2.0 ** -8


def _gelu(x):
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


@pytest.fixture(scope="module")
def ops():
    from esm_amd import ops as _ops

    return _ops


@pytest.mark.parametrize("E,rows", [(1280, 1001), (128, 7), (2560, 64), (320, 33), (5120, 5)])
@pytest.mark.parametrize("dt", DT)
def test_layernorm(ops, E, rows, dt):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(rows, E, device="cuda", generator=g) * 3 + 0.5
    gamma = 1 + 0.1 * torch.randn(E, device="cuda", generator=g)
    beta = 0.1 * torch.randn(E, device="cuda", generator=g)
    y, y32 = ops.layernorm(x, gamma, beta, dt, want_op=True, want_f32=True)
    ref = torch.nn.functional.layer_norm(x, (E,), gamma, beta, 1e-5)
    assert (y32 - ref).abs().max().item() < 2e-5
    assert (y.float() - ref).abs().max().item() <= _eps(dt) * ref.abs().max().item() * 1.01 + 1e-6


@pytest.mark.parametrize(
    "M,N,K,generic",
    [
        (300, 384, 256, False),   # M and N tails of the 256x256 tile
        (1024, 1280, 1280, False),
        (512, 1280, 5120, False),
        (64, 128, 64, False),
        (77, 33, 320, False),     # K % 64 != 0 and N % 4 != 0 -> generic kernel
        (130, 200, 96, True),
        (256, 256, 128, True),
    ],
)
@pytest.mark.parametrize("dt", DT)
def test_linear_epilogues(ops, M, N, K, generic, dt):
    from esm_amd import _native as nat

    g = torch.Generator(device="cuda").manual_seed(2)
    a = torch.randn(M, K, device="cuda", generator=g).to(dt)
    # asymmetric, non-uniform weights so that a transposed / permuted tile cannot pass
    w = (torch.randn(N, K, device="cuda", generator=g) * torch.linspace(0.5, 1.5, N, device="cuda")[:, None]).to(dt)
    bias = torch.randn(N, device="cuda", generator=g)
    ref = a.float() @ w.float().t() + bias
    tol = 3e-5 * math.sqrt(K) * 4 + 1e-5

    out = ops.linear(a, w, bias, nat.EPI_STORE_F32, force_generic=generic)
    assert (out - ref).abs().max().item() < tol
    out = ops.linear(a, w, None, nat.EPI_STORE_F32, force_generic=generic)
    assert (out - (ref - bias)).abs().max().item() < tol
    out = ops.linear(a, w, bias, nat.EPI_STORE_T, force_generic=generic)
    assert (out.float() - ref).abs().max().item() < tol + _eps(dt) * ref.abs().max().item()
    out = ops.linear(a, w, bias, nat.EPI_GELU_F32, force_generic=generic)
    assert (out - _gelu(ref)).abs().max().item() < tol
    out = ops.linear(a, w, bias, nat.EPI_GELU_T, force_generic=generic)
    assert (out.float() - _gelu(ref)).abs().max().item() < tol + _eps(dt) * ref.abs().max().item()
    resid = torch.randn(M, N, device="cuda", generator=g)
    acc = resid.clone()
    ops.linear(a, w, bias, nat.EPI_RESID_F32, out=acc, force_generic=generic)
    assert (acc - (resid + ref)).abs().max().item() < tol


@pytest.mark.parametrize(
    "M,N,K",
    [
        (8192, 2560, 1280),  # 320 tiles: workgroups walk 2 tiles, the K-tile stream crosses a tile seam
        (2304, 1280, 192),   # odd number of K tiles (LDS buffer parity flips between tiles)
        (512, 256, 64),      # a single K tile per tile
        (300, 384, 256),     # M and N tails
        (4224, 5120, 1280),  # panel order with 20 N tiles, M tail tile
        (1024, 1288, 128),   # N % 8 == 0 only
    ],
)
@pytest.mark.parametrize("dt", DT)
def test_persistent_gemm_bit_exact_vs_tile_kernel(ops, M, N, K, dt):
    """gemm8.hip (persistent, ping-pong) accumulates the K products in the same order as the
    one-tile-per-workgroup kernel of gemm.hip, so without a bias the two must agree BIT FOR BIT
    (with a bias gemm8 starts from acc = bias instead of adding it last: compared with a tolerance);
    repeated launches with different tile orders catch LDS-DMA races."""
    from esm_amd import _native as nat

    g = torch.Generator(device="cuda").manual_seed(7)
    a = torch.randn(M, K, device="cuda", generator=g).to(dt)
    w = (torch.randn(N, K, device="cuda", generator=g) * torch.linspace(0.5, 1.5, N, device="cuda")[:, None]).to(dt)
    bias = torch.randn(N, device="cuda", generator=g)
    ref = a.float() @ w.float().t() + bias
    tol = 3e-5 * math.sqrt(K) * 4 + 1e-5
    for epi in (nat.EPI_STORE_F32, nat.EPI_GELU_T, nat.EPI_STORE_T, nat.EPI_GELU_F32):
        old = ops.linear(a, w, None, epi, force_old=True)
        for rep in range(3):
            new = ops.linear(a, w, None, epi, panel_c=(0, 4, 1)[rep])
            assert torch.equal(new, old), (epi, rep, (new.float() - old.float()).abs().max().item())
    new = ops.linear(a, w, bias, nat.EPI_STORE_F32)
    assert (new - ref).abs().max().item() < tol
    new = ops.linear(a, w, bias, nat.EPI_GELU_T)
    assert (new.float() - _gelu(ref)).abs().max().item() < tol + _eps(dt) * ref.abs().max().item()
    resid = torch.randn(M, N, device="cuda", generator=g)
    acc_old, acc_new = resid.clone(), resid.clone()
    ops.linear(a, w, None, nat.EPI_RESID_F32, out=acc_old, force_old=True)
    ops.linear(a, w, None, nat.EPI_RESID_F32, out=acc_new)
    assert torch.equal(acc_new, acc_old)
    acc_new = resid.clone()
    ops.linear(a, w, bias, nat.EPI_RESID_F32, out=acc_new)
    assert (acc_new - (resid + ref)).abs().max().item() < tol


@pytest.mark.parametrize("M,N,K", [(4096, 1280, 1280), (4096, 1280, 5120), (1100, 2560, 1280), (4224, 5120, 320), (100, 264, 64)])
@pytest.mark.parametrize("dt", DT)
def test_half_height_tiles_bit_exact(ops, M, N, K, dt):
    """gemm8_kernel<..., HM>: 128 x 256 tiles for shapes that leave most CUs idle with 256-row tiles.  Every output
    element sees the same MFMA sequence over K as in the full-height kernel, so the results are bit-identical —
    whichever tile height the batch size selects (small-batch forwards must not differ from large-batch ones)."""
    from esm_amd import _native as nat

    g = torch.Generator(device="cuda").manual_seed(13)
    a = torch.randn(M, K, device="cuda", generator=g).to(dt)
    w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(dt)
    bias = torch.randn(N, device="cuda", generator=g)
    for epi in (nat.EPI_STORE_F32, nat.EPI_GELU_T, nat.EPI_STORE_T, nat.EPI_GELU_F32):
        full = ops.linear(a, w, bias, epi, half_m=-1)
        for rep in range(2):
            half = ops.linear(a, w, bias, epi, half_m=1, panel_c=(0, 2)[rep])
            assert torch.equal(half, full), (epi, rep, (half.float() - full.float()).abs().max().item())
    resid = torch.randn(M, N, device="cuda", generator=g)
    x_full, x_half = resid.clone(), resid.clone()
    ops.linear(a, w, bias, nat.EPI_RESID_F32, out=x_full, half_m=-1)
    ops.linear(a, w, bias, nat.EPI_RESID_F32, out=x_half, half_m=1)
    assert torch.equal(x_half, x_full)


def test_small_batch_forward_equals_large_batch_rows():
    """B = 2 (half-height GEMM tiles, chosen from the tile count) and B = 16 (full-height): the same sequence gives
    the same bits in both batches."""
    import esm
    from esm_amd.synth import synth_esm2_state_dict, synth_tokens

    L, E, H = 2, 1280, 20
    model = esm.ESM2(L, E, H).eval().requires_grad_(False)
    model.load_state_dict(synth_esm2_state_dict(L, E, H, seed=3))
    model = model.cuda()
    toks = synth_tokens(16, 1022, seed=5).cuda()
    big = model(toks, repr_layers=[L])
    small = model(toks[:2], repr_layers=[L])
    assert torch.equal(small["representations"][L], big["representations"][L][:2])
    assert torch.equal(small["logits"], big["logits"][:2])


def _rope_ref(x, inv_freq):
    # reference esm/rotary_embedding.py:11-20,47-61 restated for [B,H,T,d]
    T = x.shape[-2]
    t = torch.arange(T, device=x.device).float()
    freqs = torch.einsum("i,j->ij", t, inv_freq.to(x.device))
    emb = torch.cat((freqs, freqs), dim=-1)
    cos, sin = emb.cos(), emb.sin()
    x1, x2 = x.chunk(2, dim=-1)
    return x * cos + torch.cat((-x2, x1), dim=-1) * sin


@pytest.mark.parametrize("E,H,B,T", [(128, 2, 3, 100), (1280, 20, 2, 256), (256, 4, 1, 1024), (128, 2, 2, 5)])
@pytest.mark.parametrize("dt", DT)
def test_qkv_rope(ops, E, H, B, T, dt):
    g = torch.Generator(device="cuda").manual_seed(3)
    a = torch.randn(B * T, E, device="cuda", generator=g).to(dt)
    w = (torch.randn(3 * E, E, device="cuda", generator=g) / math.sqrt(E)).to(dt)
    bias = 0.1 * torch.randn(3 * E, device="cuda", generator=g)
    hnd = ops.QkvHandle(E, H, dt)
    q, k, vt = hnd(a, w, bias, B, T)
    y = a.float() @ w.float().t() + bias  # [B*T, 3E]
    yq, yk, yv = [t.reshape(B, T, H, 64).permute(0, 2, 1, 3) for t in y.split(E, dim=1)]
    yq = _rope_ref(yq * 64 ** -0.5, hnd.inv_freq)
    yk = _rope_ref(yk, hnd.inv_freq)
    tol = 1e-4 + _eps(dt) * max(yq.abs().max().item(), yk.abs().max().item(), yv.abs().max().item())
    assert (q.float() - yq).abs().max().item() < tol
    assert (k.float() - yk).abs().max().item() < tol
    ref_vt = ops.make_vt(yv.contiguous())
    assert vt.shape == ref_vt.shape
    assert (vt.float() - ref_vt).abs().max().item() < tol


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_inline_asm_mfma_matches_the_builtin(dt):
    """common.h's mma_keep_c hides a v_mfma inside inline asm with hand-counted wait states; this runs it
    next to the builtin MFMA (VALU writes of C right before, a dependent MFMA right after) on every toolchain the
    library is built with: bit-equal results, C intact."""
    from esm_amd import _native as N

    g = torch.Generator(device="cuda").manual_seed(5)
    a = torch.randn(64, 8, device="cuda", generator=g).to(dt)
    b = torch.randn(64, 8, device="cuda", generator=g).to(dt)
    c = torch.randn(64, 16, device="cuda", generator=g)
    out = torch.zeros(3, 64, 16, device="cuda")
    for _ in range(3):
        N.check(N.lib.esmk_debug_mma_selftest(N.ptr(a), N.ptr(b), N.ptr(c), N.ptr(out), N.dtype_code(dt), N.cur_stream()))
    torch.cuda.synchronize()
    assert torch.equal(out[0], out[1]), (out[0] - out[1]).abs().max().item()
    assert torch.equal(out[2], 2 * c)
    assert out[0].abs().max().item() > 0


def test_qkv_rope_composes_with_attention_through_the_c_abi(ops):
    """esmk_op_qkv_rope2(log2_domain=1) hands esmk_op_attention the q it expects (log2(e) folded into the q
    scale); chained, the two public ops reproduce softmax(q k^T / sqrt(d)) v of the reference
    (multihead_attention.py:256-261,354-394) — without the flag the softmax would run at the wrong temperature."""
    E, H, B, T, dt = 256, 4, 2, 192, torch.float16
    g = torch.Generator(device="cuda").manual_seed(11)
    a = torch.randn(B * T, E, device="cuda", generator=g).to(dt)
    w = (torch.randn(3 * E, E, device="cuda", generator=g) * (2.0 / math.sqrt(E))).to(dt)
    bias = 0.1 * torch.randn(3 * E, device="cuda", generator=g)
    hnd = ops.QkvHandle(E, H, dt)
    q, k, vt = hnd(a, w, bias, B, T, log2_domain=True)
    ctx = ops.attention(q, k, vt)
    y = a.float() @ w.float().t() + bias
    yq, yk, yv = [t.reshape(B, T, H, 64).permute(0, 2, 1, 3) for t in y.split(E, dim=1)]
    yq, yk = _rope_ref(yq * 64 ** -0.5, hnd.inv_freq), _rope_ref(yk, hnd.inv_freq)
    want = torch.softmax(yq @ yk.transpose(-1, -2), dim=-1) @ yv  # [B,H,T,64]
    want = want.permute(0, 2, 1, 3).reshape(B * T, E)
    err = (ctx.float() - want).abs().max().item()
    assert err < 4e-3 * want.abs().max().item(), err
    # and the natural-domain q of esmk_op_qkv_rope is that q divided by log2(e), to fp16 rounding
    q0, _, _ = hnd(a, w, bias, B, T)
    assert (q.float() / 1.4426950408889634 - q0.float()).abs().max().item() < 2e-3 * q0.float().abs().max().item()


def _attn_ref(q, k, v, key_bias):
    s = q.float() @ k.float().transpose(-1, -2)
    if key_bias is not None:
        s = s + key_bias[:, None, None, :]
    p = torch.softmax(s, dim=-1)
    return p, p @ v.float()


@pytest.mark.parametrize("B,H,T,pad", [(2, 3, 1024, 0), (2, 2, 100, 0), (3, 2, 257, 60), (1, 1, 64, 0), (2, 2, 12, 5)])
@pytest.mark.parametrize("dt", DT)
def test_attention(ops, B, H, T, pad, dt):
    g = torch.Generator(device="cuda").manual_seed(4)
    # scores with a usable dynamic range (row max of softmax well above uniform)
    # the kernels take q with log2(e) folded in; `q` below is the natural-domain q that operand represents exactly
    qk, q = ops.to_log2_domain(torch.randn(B, H, T, 64, device="cuda", generator=g) * 0.6, dt)
    k = (torch.randn(B, H, T, 64, device="cuda", generator=g) * 0.6).to(dt)
    v = torch.randn(B, H, T, 64, device="cuda", generator=g).to(dt)
    key_bias = None
    if pad:
        key_bias = torch.zeros(B, T, device="cuda")
        key_bias[0, T - pad:] = float("-inf")  # trailing pads on sequence 0
        if B > 1:
            key_bias[1, 3] = float("-inf")      # an interior pad on sequence 1
    vt = ops.make_vt(v)
    ctx, lse = ops.attention(qk, k, vt, key_bias, want_lse=True)
    p_ref, o_ref = _attn_ref(q, k, v, key_bias)
    o_ref = o_ref.permute(0, 2, 1, 3).reshape(B * T, H * 64)
    err = (ctx.float() - o_ref).abs().max().item()
    assert err < 4 * _eps(dt) * max(1.0, o_ref.abs().max().item()), err
    s = q.float() @ k.float().transpose(-1, -2)
    if key_bias is not None:
        s = s + key_bias[:, None, None, :]
    lse_ref = torch.logsumexp(s, dim=-1)
    assert (lse - lse_ref).abs().max().item() < 1e-3
    probs = ops.attention_probs(qk, k, lse, key_bias)
    if key_bias is not None:
        keep = (key_bias == 0).float()
        p_ref = p_ref * keep[:, None, :, None] * keep[:, None, None, :]
    assert (probs[:, 0] - p_ref).abs().max().item() < 2e-3 * p_ref.max().item() + 1e-6
    if key_bias is not None:
        assert (probs[:, 0][p_ref == 0] == 0).all()  # exact zeros on padded rows / cols


def test_attention_rescale_spike(ops):
    """A key that dominates one query late in the sweep forces the online-softmax rescale path."""
    g = torch.Generator(device="cuda").manual_seed(5)
    B, H, T = 1, 1, 512
    q = (torch.randn(B, H, T, 64, device="cuda", generator=g) * 0.3)
    k = (torch.randn(B, H, T, 64, device="cuda", generator=g) * 0.3)
    k[0, 0, 400] = q[0, 0, 17] * 40.0
    k[0, 0, 130] = q[0, 0, 300] * 25.0
    (qk, q), k = ops.to_log2_domain(q, torch.float16), k.half()
    v = torch.randn(B, H, T, 64, device="cuda", generator=g).half()
    ctx = ops.attention(qk, k, ops.make_vt(v))
    _, o_ref = _attn_ref(q, k, v, None)
    o_ref = o_ref.permute(0, 2, 1, 3).reshape(B * T, H * 64)
    assert (ctx.float() - o_ref).abs().max().item() < 4 * _eps(torch.float16) * max(1.0, o_ref.abs().max().item())


@pytest.mark.parametrize("scale", [0.6, 4.0])
def test_attention_lazy_offset_edge_cases(ops, scale):
    """The flash kernel moves its exponent offset only when it has to (attention.hip, LAZY): exercise the
    bookkeeping — a first key tile that is ENTIRELY masked (no finite score yet), rows whose early scores are far
    below later ones (offset must move up, by a lot at scale 4: raw scores of +-100), and rows whose later scores are
    far below the first tile's (offset stays; tiny probabilities)."""
    g = torch.Generator(device="cuda").manual_seed(11)
    B, H, T = 2, 2, 448
    q = (torch.randn(B, H, T, 64, device="cuda", generator=g) * scale)
    k = (torch.randn(B, H, T, 64, device="cuda", generator=g) * 0.5)
    k[0, :, 256:] *= 0.02          # sequence 0: late keys score ~0, early ones dominate or are hugely negative
    k[1, :, :128] *= 0.02          # sequence 1: the first two tiles score ~0, later keys dominate
    (qk, q), k = ops.to_log2_domain(q, torch.float16), k.half()
    v = torch.randn(B, H, T, 64, device="cuda", generator=g).half()
    key_bias = torch.zeros(B, T, device="cuda")
    key_bias[0, :70] = float("-inf")   # the whole first 64-key tile (and a bit) of sequence 0 is padding
    key_bias[1, 5] = float("-inf")
    ctx, lse = ops.attention(qk, k, ops.make_vt(v), key_bias, want_lse=True)
    p_ref, o_ref = _attn_ref(q, k, v, key_bias)
    o_ref = o_ref.permute(0, 2, 1, 3).reshape(B * T, H * 64)
    assert torch.isfinite(ctx).all()
    assert (ctx.float() - o_ref).abs().max().item() < 4 * _eps(torch.float16) * max(1.0, o_ref.abs().max().item())
    s = q.float() @ k.float().transpose(-1, -2) + key_bias[:, None, None, :]
    lse_ref = torch.logsumexp(s, dim=-1)
    assert (lse - lse_ref).abs().max().item() < 2e-3 * max(1.0, lse_ref.abs().max().item())


@pytest.mark.parametrize("B,L,H,T", [(2, 2, 3, 40), (1, 1, 2, 130)])
def test_contacts(ops, B, L, H, T):
    g = torch.Generator(device="cuda").manual_seed(6)
    attn = torch.rand(B, L, H, T, T, device="cuda", generator=g)
    attn = attn / attn.sum(-1, keepdim=True)
    tokens = torch.randint(4, 24, (B, T), device="cuda", generator=g)
    tokens[:, 0] = 0
    tokens[0, T - 1] = 2
    if B > 1:  # shorter second sequence: eos inside, pads after
        tokens[1, T - 10] = 2
        tokens[1, T - 9:] = 1
    w = torch.randn(1, L * H, device="cuda", generator=g)
    b = torch.randn(1, device="cuda", generator=g)
    out = ops.contacts(attn, tokens, w, b)
    # reference: esm/modules.py:338-357 + symmetrize/apc (modules.py:27-41), restated
    em = tokens.ne(2).float()
    a = attn * (em[:, :, None] * em[:, None, :])[:, None, None]
    a = a[..., :-1, :-1][..., 1:, 1:].reshape(B, L * H, T - 2, T - 2)
    s = a + a.transpose(-1, -2)
    a1, a2, a12 = s.sum(-1, keepdim=True), s.sum(-2, keepdim=True), s.sum((-1, -2), keepdim=True)
    n = s - a1 * a2 / a12
    ref = torch.sigmoid((n.permute(0, 2, 3, 1) @ w.t()).squeeze(-1) + b)
    assert (out - ref).abs().max().item() < 2e-5
