"""LayerNorm fold of the engine (esmk_config.ln_fold; reference esm/modules.py:120-140: the two
LayerNorm -> Linear pairs of a TransformerLayer) on the MI355X: every piece as a single op against plain torch fp32, the
chain against the unfolded engine and the oracle, and the properties the unfolded engine has (packed == padded == alone
bit for bit, small batch == rows of a large batch) with the fold on."""
import ctypes
import math
import os

import pytest
import torch

import esm
from esm_amd import _native as N
from esm_amd import ops
from esm_amd.synth import skip_param_init, synth_esm2_state_dict, synth_tokens
from oracle.esm2_oracle import esm2_forward

pytestmark = pytest.mark.gpu
DT = {torch.float16: 1, torch.bfloat16: 2}


def _gen(seed):
    return torch.Generator(device="cuda").manual_seed(seed)


def rowstats(x, ldy=None, dtype=torch.float16):
    rows, E = x.shape
    ldy = ldy or E
    y = torch.zeros(rows, ldy, dtype=dtype, device="cuda")
    mean = torch.empty(rows, device="cuda")
    rstd = torch.empty(rows, device="cuda")
    N.check(N.lib.esmk_op_rowstats(N.ptr(x), N.ptr(y), N.ptr(mean), N.ptr(rstd), rows, E, ldy, DT[dtype], N.cur_stream()))
    return y, mean, rstd


def fold_weight(w, gamma, beta, dtype=torch.float16, ld=None):
    Nn, K = w.shape
    ld = ld or K
    dst = torch.zeros(Nn, ld, dtype=dtype, device="cuda")
    b2 = torch.empty(Nn, device="cuda")
    N.check(N.lib.esmk_op_fold_weight(N.ptr(w), N.dtype_code(w.dtype), N.ptr(gamma), N.ptr(beta), N.ptr(dst), DT[dtype],
                                      N.ptr(b2), Nn, K, ld, N.cur_stream()))
    return dst, b2


def linear_ln(a, w, bias, bias2, out, epilogue, rstd=None, h16=None, part=None, parts=0, mean=None, half_m=0):
    M, K = a.shape
    Nn = w.shape[0]
    N.check(N.lib.esmk_op_linear_ln(N.ptr(a), N.ptr(w), N.ptr(bias), N.ptr(bias2), N.ptr(out), M, Nn, K, epilogue, DT[a.dtype],
                                    N.ptr(rstd), N.ptr(h16), h16.shape[1] if h16 is not None else 0, N.ptr(part), parts,
                                    N.ptr(mean), half_m, N.cur_stream()))
    return out


@pytest.mark.parametrize("rows,E,ldy,offset", [(1000, 1280, 1280, 0.0), (77, 480, 512, 3.0), (4096, 2560, 2560, -1.5)])
def test_rowstats_against_torch(rows, E, ldy, offset):
    x = torch.randn(rows, E, device="cuda", generator=_gen(1)) * 2.0 + offset
    y, mean, rstd = rowstats(x, ldy)
    m = x.double().mean(-1)
    v = x.double().var(-1, unbiased=False)
    assert (mean.double() - m).abs().max().item() < 1e-5
    assert ((rstd.double() - torch.rsqrt(v + 1e-5)).abs() / torch.rsqrt(v + 1e-5)).max().item() < 1e-5
    exp = (x - mean[:, None]).half()  # the kernel subtracts ITS mean in fp32, then rounds once
    assert torch.equal(y[:, :E], exp)
    assert ldy == E or not y[:, E:].any()


@pytest.mark.parametrize("Nn,K,ld,wdt", [(1280, 1280, 1280, torch.float32), (96, 480, 512, torch.float32), (512, 320, 320, torch.float16)])
def test_fold_weight_against_torch(Nn, K, ld, wdt):
    g = _gen(2)
    w = (torch.randn(Nn, K, device="cuda", generator=g) / math.sqrt(K)).to(wdt)
    gamma = 1 + 0.1 * torch.randn(K, device="cuda", generator=g)
    beta = 0.1 * torch.randn(K, device="cuda", generator=g)
    dst, b2 = fold_weight(w, gamma, beta, ld=ld)
    wg = w.double() * gamma.double()
    exp = wg - wg.mean(-1, keepdim=True)
    # one fp16 rounding of a value computed in fp32 with a different summation order for the row mean
    err = (dst[:, :K].double() - exp).abs().max().item()
    assert err <= 2.0 ** -11 * exp.abs().max().item() + 1e-7, err
    assert (dst[:, :K].double().sum(-1).abs().max().item()) < 2e-3  # rows are centred up to the rounding of the image
    assert (b2.double() - w.double() @ beta.double()).abs().max().item() < 1e-5
    assert ld == K or not dst[:, K:].any()


@pytest.mark.parametrize("M,Nn,K,half_m", [(1024, 512, 320, 0), (2048, 5120, 1280, 0), (1000, 520, 192, 0), (512, 1280, 1280, 1)])
def test_consumer_gelu_against_torch(M, Nn, K, half_m):
    g = _gen(3)
    a = torch.randn(M, K, device="cuda", generator=g).half()
    w = (torch.randn(Nn, K, device="cuda", generator=g) / math.sqrt(K)).half()
    bias = torch.randn(Nn, device="cuda", generator=g)
    bias2 = torch.randn(Nn, device="cuda", generator=g)
    rstd = torch.rand((M + 255) // 256 * 256, device="cuda", generator=g) + 0.5
    out = torch.empty(M, Nn, dtype=torch.float16, device="cuda")
    linear_ln(a, w, bias, bias2, out, N.EPI_GELU_T, rstd=rstd, half_m=half_m)
    pre = rstd[:M, None].double() * (a.double() @ w.double().t()) + (bias + bias2).double()
    ref = torch.nn.functional.gelu(pre)
    err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
    print(f"consumer M={M} N={Nn} K={K}: rel err {err:.2e}")
    assert err < 1e-3


# half_m: 0 = the library's choice (half-height tiles at these sizes), 1 / -1 = force half- / full-height tiles (the two kernels)
@pytest.mark.parametrize("M,Nn,K,half_m", [(1024, 1280, 1280, 0), (2048, 1280, 5120, 0), (1000, 480, 320, 0), (777, 264, 192, 0), (512, 1280, 1280, 1),
                                           (1024, 1280, 1280, -1), (2048, 1280, 5120, -1), (1000, 480, 320, -1)])
def test_producer_against_plain_residual_gemm(M, Nn, K, half_m):
    """out must be bit-identical to the plain residual epilogue; h16 = fp16(out - mean_prev); partial sums vs torch."""
    g = _gen(4)
    a = torch.randn(M, K, device="cuda", generator=g).half()
    w = (torch.randn(Nn, K, device="cuda", generator=g) / math.sqrt(K)).half()
    bias = torch.randn(Nn, device="cuda", generator=g)
    x0 = torch.randn(M, Nn, device="cuda", generator=g) * 3 + 0.7
    mean_prev = x0.mean(-1) + 0.01 * torch.randn(M, device="cuda", generator=g)
    plain = ops.linear(a, w, bias, N.EPI_RESID_F32, out=x0.clone(), half_m=half_m)
    parts = (Nn + 127) // 128
    ldh = (Nn + 63) // 64 * 64
    h16 = torch.zeros(M, ldh, dtype=torch.float16, device="cuda")
    part = torch.full((M, parts, 2), float("nan"), device="cuda")
    out = linear_ln(a, w, bias, None, x0.clone(), N.EPI_RESID_F32, h16=h16, part=part, parts=parts, mean=mean_prev, half_m=half_m)
    assert torch.equal(out, plain)
    d = out - mean_prev[:, None]
    assert torch.equal(h16[:, :Nn], d.half())
    assert ldh == Nn or not h16[:, Nn:].any()
    pad = parts * 128 - Nn
    dd = torch.nn.functional.pad(d.double(), (0, pad)).view(M, parts, 128)
    s1, s2 = dd.sum(-1), (dd * dd).sum(-1)
    assert torch.isfinite(part).all()
    assert (part[..., 0].double() - s1).abs().max().item() <= 1e-5 * dd.abs().sum(-1).max().item() + 1e-6
    assert ((part[..., 1].double() - s2).abs() / s2.clamp_min(1e-6)).max().item() < 1e-5
    # finalize: new mean and rstd of the rows
    mean = mean_prev.clone()
    rstd = torch.empty(M, device="cuda")
    N.check(N.lib.esmk_op_ln_finalize(N.ptr(part), N.ptr(mean), N.ptr(rstd), M, parts, Nn, N.cur_stream()))
    assert (mean.double() - out.double().mean(-1)).abs().max().item() < 2e-5
    rv = torch.rsqrt(out.double().var(-1, unbiased=False) + 1e-5)
    assert ((rstd.double() - rv).abs() / rv).max().item() < 2e-5


def test_qkv_chain_against_unfolded_ops():
    """rowstats -> folded q/k/v projection (+ RoPE, head split, V^T) against LayerNorm -> the plain fused projection."""
    B, T, E, H = 3, 160, 1280, 20
    g = _gen(5)
    x = torch.randn(B * T, E, device="cuda", generator=g) * 2 + 0.3
    gamma = 1 + 0.1 * torch.randn(E, device="cuda", generator=g)
    beta = 0.1 * torch.randn(E, device="cuda", generator=g)
    w = torch.randn(3 * E, E, device="cuda", generator=g) / math.sqrt(E)
    bias = 0.1 * torch.randn(3 * E, device="cuda", generator=g)
    qkv = ops.QkvHandle(E, H)
    hln = torch.nn.functional.layer_norm(x, (E,), gamma, beta, 1e-5).half()
    q0, k0, v0 = qkv(hln, w.half(), bias, B, T, log2_domain=True)
    y, mean, rstd = rowstats(x)
    rstd_p = torch.zeros((B * T + 255) // 256 * 256, device="cuda")
    rstd_p[: B * T] = rstd
    wf, b2 = fold_weight(w, gamma, beta)
    Tp = (T + 63) // 64 * 64
    q = torch.empty((B, H, T, 64), dtype=torch.float16, device="cuda")
    k = torch.empty_like(q)
    vt = torch.zeros((B, H, 64, Tp), dtype=torch.float16, device="cuda")
    N.check(N.lib.esmk_op_qkv_rope_ln(qkv.h, N.ptr(y), N.ptr(wf), N.ptr(bias), N.ptr(b2), N.ptr(rstd_p), N.ptr(q), N.ptr(k),
                                      N.ptr(vt), B, T, 1, N.cur_stream()))
    for name, got, ref in (("q", q, q0), ("k", k, k0), ("vt", vt[..., :T], v0[..., :T])):
        err = (got.float() - ref.float()).abs().max().item() / ref.float().abs().max().item()
        print(f"folded {name} vs unfolded: rel {err:.2e}")
        assert err < 3e-3, (name, err)


@pytest.mark.parametrize("offset", [1, 2, 3, 5])
def test_folded_v_bits_do_not_depend_on_row_offset(offset):
    """The folded projection of a token must not depend on the row of the batch it sits in: V^T of the same 200 rows
    with `offset` other rows in front is bit-equal to the rows alone (q and k carry the rotary position, which moves
    with the offset).  Found with the padded == packed test in fold mode: the compiler had fused the rstd * acc + bias
    of V elements 0 and 3 of each 4-token group with the fp16 conversion (one rounding) and left elements 1 and 2 with
    two, so V changed by an ulp with the row index modulo 4."""
    n, E, H = 200, 1280, 20
    g = _gen(3)
    x = torch.randn(n, E, device="cuda", generator=g) * 2 + 0.1
    gamma = 1 + 0.1 * torch.randn(E, device="cuda", generator=g)
    beta = 0.1 * torch.randn(E, device="cuda", generator=g)
    w = torch.randn(3 * E, E, device="cuda", generator=g) / math.sqrt(E)
    bias = 0.1 * torch.randn(3 * E, device="cuda", generator=g)
    wf, b2 = fold_weight(w, gamma, beta)
    qkv = ops.QkvHandle(E, H)

    def run(k):
        T = n + k
        xx = torch.cat([torch.randn(k, E, device="cuda", generator=g), x]) if k else x
        y, _, rstd = rowstats(xx)
        rstd_p = torch.zeros((T + 255) // 256 * 256, device="cuda")
        rstd_p[:T] = rstd
        q = torch.empty((1, H, T, 64), dtype=torch.float16, device="cuda")
        kk = torch.empty_like(q)
        vt = torch.zeros((1, H, 64, (T + 63) // 64 * 64), dtype=torch.float16, device="cuda")
        N.check(N.lib.esmk_op_qkv_rope_ln(qkv.h, N.ptr(y), N.ptr(wf), N.ptr(bias), N.ptr(b2), N.ptr(rstd_p), N.ptr(q), N.ptr(kk),
                                          N.ptr(vt), 1, T, 1, N.cur_stream()))
        t = torch.arange(T, device="cuda")
        t16 = t & 15  # the key permutation of V^T (include/esmk.h: 4-groups 1 and 2 of each 16 keys swapped)
        tp = (t & ~15) | (((t16 >> 2) & 1) << 3) | (((t16 >> 3) & 1) << 2) | (t16 & 3)
        return vt[0][:, :, tp][:, :, k:].clone()

    v0, vk = run(0), run(offset)
    assert torch.equal(v0, vk), int((v0 != vk).sum())


def build(L, E, H, seed):
    sd = synth_esm2_state_dict(L, E, H, seed=seed)
    with skip_param_init():
        m = esm.ESM2(L, E, H).eval()
    m.load_state_dict(sd)
    return m.cuda(), sd


def run_mode(model, fold, fn):
    old = os.environ.get("ESM_AMD_LN_FOLD")
    os.environ["ESM_AMD_LN_FOLD"] = "1" if fold else "0"
    try:
        with torch.no_grad():
            return fn()
    finally:
        if old is None:
            del os.environ["ESM_AMD_LN_FOLD"]
        else:
            os.environ["ESM_AMD_LN_FOLD"] = old


@pytest.mark.parametrize("L,E,H", [(3, 1280, 20), (4, 320, 20), (3, 480, 20), (2, 640, 20)])
def test_fold_against_oracle_and_unfolded_engine(L, E, H):
    """head_dim 64 / 16 / 24 (E = 480: K-padded rows) / 32; padded batch with <mask>, contacts and attention maps."""
    model, sd = build(L, E, H, seed=11)
    toks = synth_tokens(3, 150, seed=5)
    toks[1, 100] = 2
    toks[1, 101:] = 1
    toks[0, 17] = 32
    reps = list(range(L + 1))
    f = lambda: model(toks.cuda(), repr_layers=reps, return_contacts=True)
    plain = run_mode(model, False, f)
    fold = run_mode(model, True, f)
    ref = esm2_forward(sd, toks, L, H, repr_layers=reps, return_contacts=True)
    nonpad = toks.ne(1)
    rel = lambda a, b: ((a.cpu().double() - b.double())[nonpad].abs().max() / b.double()[nonpad].abs().max()).item()
    for l in reps:
        ef, ep = rel(fold["representations"][l], ref["representations"][l]), rel(plain["representations"][l], ref["representations"][l])
        print(f"L={L} E={E}: layer {l}: fold {ef:.2e} plain {ep:.2e}")
        assert ef < 2e-3, (l, ef, ep)
    assert rel(fold["logits"], ref["logits"]) < 2e-3
    assert (fold["contacts"].cpu() - ref["contacts"]).abs().max().item() < 8e-3
    assert (fold["attentions"].cpu() - ref["attentions"]).abs().max().item() < 5e-3
    assert not torch.equal(fold["representations"][L], plain["representations"][L])  # the two modes really differ


def test_fold_packed_equals_padded_equals_alone():
    model, _ = build(3, 1280, 20, seed=12)
    lens = [150, 33, 97, 128, 64]
    T = max(lens) + 2
    toks = torch.ones(len(lens), T, dtype=torch.int64)
    for i, n in enumerate(lens):
        toks[i, : n + 2] = synth_tokens(1, n, seed=20 + i)[0]
    def f():
        padded = model(toks.cuda(), repr_layers=[3])
        packed = model.forward_varlen(toks, repr_layers=[3], min_saving=None)
        alone = [model(toks[i : i + 1, : n + 2].cuda(), repr_layers=[3]) for i, n in enumerate(lens)]
        return padded, packed, alone
    padded, packed, alone = run_mode(model, True, f)
    for i, n in enumerate(lens):
        a = padded["representations"][3][i, : n + 2]
        assert torch.equal(a, packed["representations"][3][i, : n + 2]), i
        assert torch.equal(a, alone[i]["representations"][3][0]), i
        assert torch.equal(padded["logits"][i, : n + 2], alone[i]["logits"][0]), i


def test_fold_small_batch_equals_rows_of_large_batch():
    """B = 2 takes half-height tiles, B = 16 full-height ones: the fold's epilogues give the same bits in both."""
    model, _ = build(2, 1280, 20, seed=13)
    toks = synth_tokens(16, 254, seed=6)
    def f():
        return model(toks.cuda(), repr_layers=[2])["representations"][2], model(toks[:2].cuda(), repr_layers=[2])["representations"][2]
    big, small = run_mode(model, True, f)
    assert torch.equal(big[:2], small)


def test_fold_pack_order_is_enforced():
    """A q/k/v or fc1 weight packed before its LayerNorm parameters is an error, and a LayerNorm parameter packed after
    them leaves a stale fold that esmk_forward refuses."""
    L, E, H = 1, 128, 2
    cfg = N.EsmkConfig(L, E, H, 4 * E, 33, 1, 32, 0, 2, 1, 1, 1, 1, 0, 0, 0, 0, 1)
    h = ctypes.c_void_p()
    N.check(N.lib.esmk_create(ctypes.byref(cfg), ctypes.byref(h)))
    try:
        nb = ctypes.c_size_t()
        N.check(N.lib.esmk_packed_bytes(h, ctypes.byref(nb)))
        packed = torch.zeros(nb.value, dtype=torch.uint8, device="cuda")
        w = torch.randn(E, E, device="cuda")
        shape = (ctypes.c_int64 * 2)(E, E)
        rc = N.lib.esmk_pack_weight(h, N.ptr(packed), packed.numel(), b"layers.0.self_attn.q_proj.weight", N.ptr(w), 0, shape, 2,
                                    N.cur_stream())
        assert rc != 0 and b"LayerNorm" in N.lib.esmk_last_error()
    finally:
        N.lib.esmk_destroy(h)
    # weight_split and the fold exclude each other
    cfg = N.EsmkConfig(L, E, H, 4 * E, 33, 1, 32, 0, 2, 1, 1, 1, 1, 0, 0, 0, 1, 1)
    assert N.lib.esmk_create(ctypes.byref(cfg), ctypes.byref(h)) != 0


def test_gain_check_follows_the_parameters(monkeypatch):
    """ESM2._engine_ready (round 6): with ESM_AMD_LN_FOLD unset the package leaves the fold off for checkpoints whose LayerNorm
    gains silence large channels (esm_amd/esm2.py ln_fold_hazard) — decided on the CURRENT parameters: loading such weights
    into a model that has already run makes an engine of the other mode, loading ordinary ones brings the fold back; either
    way the outputs are those of a fresh model with the same weights, bit for bit.  ESM_AMD_LN_FOLD=1 overrides the check."""
    from esm_amd.synth import add_outlier_channels

    monkeypatch.delenv("ESM_AMD_LN_FOLD", raising=False)
    L, E, H = 3, 128, 2
    plain_sd = synth_esm2_state_dict(L, E, H, seed=4)
    stress_sd = synth_esm2_state_dict(L, E, H, seed=4)
    add_outlier_channels(stress_sd, L, E, magnitude=300.0)
    toks = synth_tokens(2, 30, seed=9).cuda()

    def fresh(sd):
        with skip_param_init():
            m = esm.ESM2(L, E, H).eval()
        m.load_state_dict(sd)
        return m.cuda()

    with torch.no_grad():
        model = fresh(plain_sd)
        a = model(toks, repr_layers=[L])
        assert model.ln_fold_active() is True and model._fold_hazard == 0.0
        model.load_state_dict(stress_sd)
        b = model(toks, repr_layers=[L])
        assert model.ln_fold_active() is False and model._fold_hazard > 0.5
        ref = fresh(stress_sd)
        rb = ref(toks, repr_layers=[L])
        assert ref.ln_fold_active() is False
        assert torch.equal(b["logits"], rb["logits"]) and torch.equal(b["representations"][L], rb["representations"][L])
        model.load_state_dict(plain_sd)
        c = model(toks, repr_layers=[L])
        assert model.ln_fold_active() is True
        assert torch.equal(c["logits"], a["logits"]) and torch.equal(c["representations"][L], a["representations"][L])
        monkeypatch.setenv("ESM_AMD_LN_FOLD", "1")
        model.load_state_dict(stress_sd)
        model(toks, repr_layers=[L])
        assert model.ln_fold_active() is True
