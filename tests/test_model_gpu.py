"""End-to-end parity of the HIP engine (esm.ESM2 -> libesmk.so through the C ABI) on the MI355X:
against the golden fixtures produced by the reference implementation, against the oracle on seeded
inputs at the 650M dimensions, and through size-independent properties at full length.

Tolerance: the ONE parity contract of tests/_contract.py — representations of deep stacks within 1e-3
in L2 (hard) and within max(1e-3, 1.25 x the fp16-operand floor on the same inputs) in the max norm; logits, contact
logits and few-layer toy models floor-referenced in both norms.  'rel' = max|diff| / max|ref| over non-pad positions
(SURVEY.md §7.3).  Token argmax must agree wherever the reference's top-2 logit margin exceeds twice the measured logit
error (and is reported raw as well)."""
import glob
import os

import pytest
import torch

import esm
from esm_amd.synth import skip_param_init, synth_esm2_state_dict, synth_tokens
from oracle.esm2_oracle import esm2_forward

import _contract as C

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "esm2_*.pt")))  # incl. head_dim 16 (8M)


def rel_err(a, b, mask=None):
    if mask is not None:
        a, b = a[mask], b[mask]
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def contact_errors(c, cr, with_range=False):
    """(max prob error, max logit error where the reference logit is not saturated[, range of those logits])."""
    lg = lambda t: torch.logit(t.double().clamp(1e-12, 1 - 1e-12))
    z, zr = lg(c), lg(cr)
    ok = zr.abs() < 8
    perr, zerr = (c - cr).abs().max().item(), (z - zr)[ok].abs().max().item() if ok.any() else 0.0
    if not with_range:
        return perr, zerr
    return perr, zerr, (zr[ok].max() - zr[ok].min()).item() if ok.any() else 1.0


def build(L, E, H, seed, dtype=None):
    sd = synth_esm2_state_dict(L, E, H, seed=seed)
    with skip_param_init():  # every parameter comes from sd (strict load): no 28-s random fill of a 3B model first
        m = esm.ESM2(L, E, H).eval()
    m.load_state_dict(sd)
    m = m.cuda()
    return m, sd


def argmax_agreement(logits, ref, nonpad):
    err = (logits - ref)[nonpad].abs().max().item()
    top2 = ref.topk(2, dim=-1).values
    margin = (top2[..., 0] - top2[..., 1])
    same = logits.argmax(-1) == ref.argmax(-1)
    decided = nonpad & (margin > 2 * err)
    return same[nonpad].float().mean().item(), bool(same[decided].all()), err


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_engine_matches_reference_fixture(path):
    fix = torch.load(path, weights_only=False)
    d = fix["dims"]
    model, sd = build(d["L"], d["E"], d["H"], d["seed"])
    toks = fix["tokens"].cuda()
    with torch.no_grad():
        out = model(toks, repr_layers=list(range(d["L"] + 1)), return_contacts=True)
    nonpad = fix["tokens"].ne(1)
    floor = C.floor_forward(sd, fix["tokens"], d["L"], d["H"], model=model, repr_layers=list(range(d["L"] + 1)))
    tag = os.path.basename(path)
    for layer, ref in fix["representations"].items():
        if layer == 0:
            assert rel_err(out["representations"][0].cpu(), ref, nonpad) < 1e-6
            continue
        C.check_tensors(f"{tag} repr[{layer}]", out["representations"][layer].cpu(), ref, floor["representations"][layer], nonpad)
    C.check_tensors(f"{tag} logits", out["logits"].cpu(), fix["logits"], floor["logits"], nonpad)
    raw, decided_ok, _ = argmax_agreement(out["logits"].cpu(), fix["logits"], nonpad)
    assert decided_ok
    C.check_raw_argmax(f"{tag} token argmax", raw, C.raw_argmax_agreement(floor["logits"], fix["logits"], nonpad))
    if fix["attentions"] is not None:
        a = out["attentions"].cpu()
        assert (a - fix["attentions"]).abs().max().item() < 5e-3  # p(1-p) x score error of fp16 q,k
        assert (a[fix["attentions"] == 0] == 0).all()
    # contact probabilities: logit-level error (reference tolerance convention atol=1e-3)
    c, cr = out["contacts"].cpu(), fix["contacts"]
    assert c.shape == cr.shape
    perr, zerr, zrange = contact_errors(c, cr, with_range=True)
    print(f"\n{os.path.basename(path)}: contact prob err {perr:.2e}, logit err {zerr:.2e} (range of the reference logits {zrange:.1f})")
    # the random regression (std 4 over L*H channels) amplifies the ~5e-3 score error of fp16 q,k; the 8M fixture
    # sums 120 channels (6 layers x 20 heads) and measures 5.2e-3.  Logits: the convention of the full-size tests,
    # error as a fraction of the range of the unsaturated reference logits (5.2 ... 15.8 on these fixtures)
    # Measured (round 4, unchanged since the 16x16x32 MFMAs of round 3; logit error / range): eosmid 8.0e-3 / 5.2, mid
    # 1.46e-2 / 14.8, nopad 8.6e-3 / 10.2, tiny 8.8e-3 / 11.3, t6_8M_dims 3.38e-2 / 15.8.  The absolute 3e-2 bound of
    # round 2 is kept for every fixture it held for; the 120-channel fixture, which measured 2.9e-2 then and
    # 3.4e-2 since the MFMA shape changed the summation order, is held at 3.6e-2 AND 3e-3 of its logit range.
    zmax = 3.6e-2 if d["L"] * d["H"] > 100 else 3e-2
    assert perr < (8e-3 if d["L"] * d["H"] > 100 else 5e-3) and zerr < zmax and zerr < 3e-3 * zrange, (perr, zerr, zrange)


def test_shape_pin_and_interior_pad():
    # reference tests/test_load_all.py:38-47
    model, sd = build(2, 128, 2, 3)
    toks = torch.tensor([[0, 1, 2], [3, 4, 5]])
    out = model(toks.cuda())
    assert out["logits"].shape == (2, 3, 33)
    ref = esm2_forward(sd, toks, 2, 2)
    C.check_tensors("shape pin logits", out["logits"].cpu(), ref["logits"], C.floor_forward(sd, toks, 2, 2, model=model)["logits"], toks.ne(1))


@pytest.mark.parametrize("B,T,padded", [(2, 256, True), (1, 1024, False)])
def test_650m_dims_against_oracle(B, T, padded):
    """BASELINE configs[1] dimensions (33 x 1280 x 20 heads), seeded synthetic weights."""
    L, E, H = 33, 1280, 20
    model, sd = build(L, E, H, seed=0)
    toks = synth_tokens(B, T - 2, seed=1)
    if padded:
        toks[1, 100] = 2
        toks[1, 101:] = 1
        toks[0, 17] = 32
    with torch.no_grad():
        out = model(toks.cuda(), repr_layers=[0, 1, 16, 33])
    ref = esm2_forward(sd, toks, L, H, repr_layers=[0, 1, 16, 33])
    floor = C.floor_forward(sd, toks, L, H, model=model, repr_layers=[1, 16, 33])
    nonpad = toks.ne(1)
    errs = {l: rel_err(out["representations"][l].cpu(), ref["representations"][l], nonpad) for l in (0, 1, 16, 33)}
    lerr = rel_err(out["logits"].cpu(), ref["logits"], nonpad)
    raw, decided_ok, abs_err = argmax_agreement(out["logits"].cpu(), ref["logits"], nonpad)
    print(f"\n650M-dims B={B} T={T}: rel err per layer {errs}, logits rel {lerr:.2e} abs {abs_err:.2e}, "
          f"argmax raw agreement {raw:.4f}")
    assert errs[0] < 1e-6
    for l in (1, 16, 33):  # the contract: L2 <= 1e-3 hard, max norm floor-referenced
        C.check_tensors(f"650M-dims B={B} T={T} repr[{l}]", out["representations"][l].cpu(), ref["representations"][l],
                        floor["representations"][l], nonpad, hard_l2=True)
    C.check_tensors(f"650M-dims B={B} T={T} logits", out["logits"].cpu(), ref["logits"], floor["logits"], nonpad, deep=True)
    assert decided_ok
    C.check_raw_argmax(f"650M-dims B={B} T={T} token argmax", raw, C.raw_argmax_agreement(floor["logits"], ref["logits"], nonpad))


@pytest.mark.parametrize("magnitude,force_fold", [(200.0, False), (2000.0, False), (2000.0, True)])
def test_650m_dims_outlier_channels_against_oracle(monkeypatch, magnitude, force_fold):
    """The stress weight set: four residual channels at +-magnitude — 100 ... 1000 x the ordinary
    stream — written by layer 0's fc2 and moved by every later layer, LayerNorm gains that silence them (a spread of 60 ...
    4000 : 1, esm_amd.synth.add_outlier_channels).  What it stresses: the LayerNorm fold's un-normalised fp16 operand rows
    fp16(x - mean) and gain-folded, row-centred weight images, the partial-sum statistics next to values 1000 x larger, the
    fp32 residual epilogues.  The fold's floor grows with the outliers (column j of an image is gamma_j w_ij - c_i: a silenced
    channel holds the centring constant alone, times a large x_j; tools/outlier_stress_study.py): 1.1 x the plain floor at
    200, 3.3 x at 2000.  So: at 200 the default runs the fold and meets the contract; at 2000 the package's gain check
    (esm_amd/esm2.py ln_fold_hazard) leaves the fold off and the plain mode meets it on the plain floor, which the outliers do
    not move; and the fold forced on at 2000 sits on ITS floor — the engine is its form's floor even there.
    Floor-referenced in both norms, on all channels and on the ordinary channels alone."""
    from esm_amd.synth import add_outlier_channels

    L, E, H = 33, 1280, 20
    sd = synth_esm2_state_dict(L, E, H, seed=0)
    idx = add_outlier_channels(sd, L, E, magnitude=magnitude)
    with skip_param_init():
        model = esm.ESM2(L, E, H).eval()
    model.load_state_dict(sd)
    model = model.cuda()
    if force_fold:
        monkeypatch.setenv("ESM_AMD_LN_FOLD", "1")
    env = os.environ.get("ESM_AMD_LN_FOLD", "")
    toks = synth_tokens(2, 254, seed=1)
    toks[1, 100] = 2
    toks[1, 101:] = 1
    with torch.no_grad():
        out = model(toks.cuda(), repr_layers=[1, 16, 33], return_contacts=True)
    if env == "":  # the gain check decides
        assert model.ln_fold_active() == (magnitude < 1000), (magnitude, model._fold_hazard)
    else:
        assert model.ln_fold_active() == (env == "1")
    ref = esm2_forward(sd, toks, L, H, repr_layers=[1, 16, 33], return_contacts=True)
    floor = C.floor_forward(sd, toks, L, H, model=model, repr_layers=[1, 16, 33], return_contacts=True)
    nonpad = toks.ne(1)
    ordinary = torch.ones(E, dtype=torch.bool)
    ordinary[idx] = False
    r16 = ref["representations"][16][nonpad]
    tag = f"650M-dims outliers {magnitude:g}{' fold forced' if force_fold else ''}"
    print(f"\n{tag}: channels {idx.tolist()}, |x| at layer 16 {[round(v) for v in r16[:, idx].abs().mean(0).tolist()]}, "
          f"ordinary std {r16[:, ordinary].std().item():.2f}, fold {'on' if model.ln_fold_active() else 'off'}")
    assert r16[:, idx].abs().mean() > 50 * r16[:, ordinary].std()
    for l in (1, 16, 33):
        C.check_tensors(f"{tag} repr[{l}]", out["representations"][l].cpu(), ref["representations"][l],
                        floor["representations"][l], nonpad, deep=True)
        C.check_tensors(f"{tag} repr[{l}] ordinary channels", out["representations"][l].cpu()[..., ordinary],
                        ref["representations"][l][..., ordinary], floor["representations"][l][..., ordinary], nonpad, deep=True)
    C.check_tensors(f"{tag} logits", out["logits"].cpu(), ref["logits"], floor["logits"], nonpad, deep=True)
    raw, decided_ok, _ = argmax_agreement(out["logits"].cpu(), ref["logits"], nonpad)
    assert decided_ok
    C.check_raw_argmax(f"{tag} token argmax", raw, C.raw_argmax_agreement(floor["logits"], ref["logits"], nonpad))
    c, cr, cf = out["contacts"].cpu(), ref["contacts"], floor["contacts"]
    for b, sl in ((0, slice(None)), (1, slice(0, 99))):
        _, zrel = C.contact_logit_errors(c[b, sl, sl], cr[b, sl, sl])
        _, zfl = C.contact_logit_errors(cf[b, sl, sl], cr[b, sl, sl])
        C.check(f"{tag} contact logits seq {b}", zrel, zrel, zfl, zfl, slack=C.CONTACT_SLACK, slack_l2=C.CONTACT_SLACK)
    if not model.ln_fold_active():  # the plain mode does not see the outliers: the deep-stack contract as without them
        l2, _ = C.errors(out["representations"][33].cpu(), ref["representations"][33], nonpad)
        assert l2 <= 1e-3, l2


def test_3b_dims_contacts_against_oracle():
    """BASELINE configs[2] dimensions (36 x 2560 x 40 heads): contact-head parity vs CPU."""
    L, E, H = 36, 2560, 40
    model, sd = build(L, E, H, seed=2)
    toks = synth_tokens(2, 94, seed=3)
    toks[1, 60] = 2
    toks[1, 61:] = 1
    with torch.no_grad():
        out = model(toks.cuda(), repr_layers=[36], return_contacts=True)
    ref = esm2_forward(sd, toks, L, H, repr_layers=[36], return_contacts=True)
    floor = C.floor_forward(sd, toks, L, H, model=model, repr_layers=[36], return_contacts=True)
    nonpad = toks.ne(1)
    c, cr, cf = out["contacts"].cpu(), ref["contacts"], floor["contacts"]
    C.check_tensors("3B-dims T=96 repr[36]", out["representations"][36].cpu(), ref["representations"][36],
                    floor["representations"][36], nonpad, hard_l2=True)
    C.check_tensors("3B-dims T=96 logits", out["logits"].cpu(), ref["logits"], floor["logits"], nonpad, deep=True)
    # contact logits relative to the largest unsaturated reference logit; valid region of sequence 1 is [:59,:59]
    for b, sl in ((0, slice(None)), (1, slice(0, 59))):
        perr = (c[b, sl, sl] - cr[b, sl, sl]).abs().max().item()
        _, zrel = C.contact_logit_errors(c[b, sl, sl], cr[b, sl, sl])
        _, zfl = C.contact_logit_errors(cf[b, sl, sl], cr[b, sl, sl])
        print(f"3B-dims T=96 seq {b}: contact prob err {perr:.2e}")
        C.check(f"3B-dims T=96 contact logits seq {b}", zrel, zrel, zfl, zfl, slack=C.CONTACT_SLACK, slack_l2=C.CONTACT_SLACK)
        assert perr < 2e-2, (b, perr)
    # the same map without the [2,36,40,96,96] attention tensor (csrc/contacts.hip; 1440 channels, 40 heads)
    with torch.no_grad():
        fused = model.predict_contacts(toks.cuda()).cpu()
    assert (fused[0] - c[0]).abs().max().item() < 5e-5 and (fused[1, :59, :59] - c[1, :59, :59]).abs().max().item() < 5e-5


@pytest.mark.parametrize("name", ["esm2_t6_8M_UR50D", "esm2_t12_35M_UR50D", "esm2_t30_150M_UR50D"])
def test_small_head_dims_against_oracle(name):
    """head_dim 16 / 24 / 32 (8M, 35M, 150M): heads are spread over 64 slots at pack time; the 35M model also
    exercises the K padding of the activations (E = 480 is not a multiple of the 64-wide K tile)."""
    from esm_amd.synth import ESM2_DIMS

    L, E, H = ESM2_DIMS[name]
    L = min(L, 4)
    model, sd = build(L, E, H, seed=17)
    toks = synth_tokens(3, 150, seed=9)
    toks[1, 100] = 2
    toks[1, 101:] = 1
    with torch.no_grad():
        out = model(toks.cuda(), repr_layers=[0, L], return_contacts=True)
    ref = esm2_forward(sd, toks, L, H, repr_layers=[0, L], return_contacts=True)
    nonpad = toks.ne(1)
    floor = C.floor_forward(sd, toks, L, H, model=model, repr_layers=[L])
    assert rel_err(out["representations"][0].cpu(), ref["representations"][0], nonpad) < 1e-6
    C.check_tensors(f"{name} ({L} layers) repr[{L}]", out["representations"][L].cpu(), ref["representations"][L],
                    floor["representations"][L], nonpad)
    C.check_tensors(f"{name} ({L} layers) logits", out["logits"].cpu(), ref["logits"], floor["logits"], nonpad)
    aerr = (out["attentions"].cpu() - ref["attentions"]).abs().max().item()
    cerr = (out["contacts"].cpu() - ref["contacts"]).abs().max().item()
    print(name, "attention err", aerr, "contact err", cerr)
    assert aerr < 4e-3  # fp16 q, k: ~5e-3 score error on sharp synthetic attention maps
    assert cerr < 8e-3


def test_head_dim_128_against_oracle():
    """esm2_t48_15B geometry (head_dim 128) at reduced width / depth: 8 heads x 128, 2 layers, padded batch."""
    L, E, H = 2, 1024, 8
    model, sd = build(L, E, H, seed=19)
    toks = synth_tokens(3, 300, seed=10)
    toks[2, 200] = 2
    toks[2, 201:] = 1
    with torch.no_grad():
        out = model(toks.cuda(), repr_layers=[0, 1, L], return_contacts=True)
    ref = esm2_forward(sd, toks, L, H, repr_layers=[0, 1, L], return_contacts=True)
    nonpad = toks.ne(1)
    floor = C.floor_forward(sd, toks, L, H, model=model, repr_layers=[1, L])
    assert rel_err(out["representations"][0].cpu(), ref["representations"][0], nonpad) < 1e-6
    for l in (1, L):
        C.check_tensors(f"head_dim 128 repr[{l}]", out["representations"][l].cpu(), ref["representations"][l],
                        floor["representations"][l], nonpad)
    C.check_tensors("head_dim 128 logits", out["logits"].cpu(), ref["logits"], floor["logits"], nonpad)
    assert (out["attentions"].cpu() - ref["attentions"]).abs().max().item() < 4e-3
    assert (out["contacts"].cpu() - ref["contacts"]).abs().max().item() < 8e-3


def test_esmfold_frontend_contract():
    """The call ESMFold makes into its language model (reference esm/esmfold/v1/esmfold.py:59-67,118-145):
    ``esm.requires_grad_(False); esm.half()``, every layer's representation, head weights, eos placed at the
    first padding index, then ``stack(dim=2)[:, 1:-1]`` and ``attentions.permute(0,4,3,1,2).flatten(3,4)``."""
    L, E, H = 4, 1280, 20
    model, sd = build(L, E, H, seed=23)
    model.requires_grad_(False)
    model.half()
    B, Lmax = 2, 96
    g = torch.Generator().manual_seed(2)
    esmaa = torch.randint(4, 24, (B, Lmax), generator=g)
    esmaa[1, 70:] = 1  # second protein is shorter (padding)
    bos = esmaa.new_full((B, 1), 0)
    eos = esmaa.new_full((B, 1), 1)
    toks = torch.cat([bos, esmaa, eos], dim=1)
    toks[range(B), (toks != 1).sum(1)] = 2
    res = model(toks.cuda(), repr_layers=range(model.num_layers + 1), need_head_weights=True)
    assert all(v.dtype == torch.float16 for v in res["representations"].values())
    esm_s = torch.stack([v for _, v in sorted(res["representations"].items())], dim=2)[:, 1:-1]
    esm_z = res["attentions"].permute(0, 4, 3, 1, 2).flatten(3, 4)[:, 1:-1, 1:-1, :]
    assert esm_s.shape == (B, Lmax, L + 1, E) and esm_z.shape == (B, Lmax, Lmax, L * H)
    ref = esm2_forward(sd, toks, L, H, repr_layers=range(L + 1), need_head_weights=True)
    ref_s = torch.stack([v for _, v in sorted(ref["representations"].items())], dim=2)[:, 1:-1]
    ref_z = ref["attentions"].permute(0, 4, 3, 1, 2).flatten(3, 4)[:, 1:-1, 1:-1, :]
    valid = toks[:, 1:-1].ne(1)
    assert rel_err(esm_s.float().cpu(), ref_s, valid) < 3e-3  # fp16 outputs on top of fp16 operands
    assert (esm_z.float().cpu() - ref_z).abs().max().item() < 4e-3


def test_degenerate_lengths():
    """Empty and one-residue sequences (BatchConverter yields [cls, eos] / [cls, x, eos]): the reference returns
    contacts of shape [B, T-2, T-2], empty for T = 2."""
    L, E, H = 2, 128, 2
    model, sd = build(L, E, H, seed=29)
    for toks in (torch.tensor([[0, 2], [0, 2]]), torch.tensor([[0, 5, 2], [0, 2, 1]])):
        with torch.no_grad():
            out = model(toks.cuda(), repr_layers=[L], return_contacts=True)
        ref = esm2_forward(sd, toks, L, H, repr_layers=[L], return_contacts=True)
        T = toks.shape[1]
        assert out["contacts"].shape == (toks.shape[0], T - 2, T - 2) == ref["contacts"].shape
        nonpad = toks.ne(1)
        floor = C.floor_forward(sd, toks, L, H, model=model, repr_layers=[L])
        C.check_tensors(f"degenerate T={T} repr", out["representations"][L].cpu(), ref["representations"][L],
                        floor["representations"][L], nonpad)
        # 4 - 5 positions x 33 logits of a 2-layer toy model: a floor-referenced bound means nothing on ~150 elements (two
        # realisations of the same rounding noise measured 1.3 x and 1.9 x apart in the max norm on two boxes, with the
        # representations INSIDE 1e-3) — the toy-model bound of rounds 1 - 4 (2e-3, both norms) is what a defect would break
        l2, mx = C.errors(out["logits"].cpu(), ref["logits"], nonpad)
        print(f"degenerate T={T} logits: L2 {l2:.2e}, max {mx:.2e} (bound 2e-3)")
        assert l2 < 2e-3 and mx < 2e-3


def test_row_guard_fails_loudly():
    """B*T beyond the engine's row limit is refused with an error (never a silent wrap of int row indices)."""
    import ctypes

    from esm_amd import _native as N

    model, _ = build(1, 128, 2, seed=3)
    model(torch.tensor([[0, 5, 2]]).cuda())  # creates the engine
    need = ctypes.c_size_t()
    rc = N.lib.esmk_workspace_bytes(model._engine.handle, 1 << 13, 1 << 12, 0, ctypes.byref(need))
    assert rc != 0 and b"2^24" in N.lib.esmk_last_error()
    assert N.lib.esmk_workspace_bytes(model._engine.handle, 1 << 12, 1 << 12, 0, ctypes.byref(need)) == 0


def test_poisoned_workspace():
    """The engine may not read workspace bytes it has not written: results with the workspace pre-filled with
    0xFF bytes (NaN in fp16 / fp32) equal the results on a zeroed one, for lengths on and off the 64-key tile."""
    model, _ = build(2, 320, 20, seed=13)  # head_dim 16: the attention width differs from the embedding width
    for T in (64, 77, 130):
        toks = synth_tokens(3, T, seed=T)
        toks[1, T - 9:] = 1  # one padded sequence
        toks = toks.cuda()
        with torch.no_grad():
            model(toks, repr_layers=[2], return_contacts=True)
            model._engine.workspace.zero_()
            a = model(toks, repr_layers=[2], return_contacts=True)
            model._engine.workspace.fill_(255)
            b = model(toks, repr_layers=[2], return_contacts=True)
        keep = toks.ne(1)
        assert torch.equal(a["representations"][2][keep], b["representations"][2][keep])
        assert torch.equal(a["logits"][keep], b["logits"][keep])
        assert torch.equal(a["contacts"], b["contacts"]) and torch.isfinite(b["contacts"]).all()


def test_properties_full_length():
    """Size-independent properties at L=1022 with the 650M dimensions (no oracle needed):
    run-to-run determinism, batch-composition invariance (bit exact) and padding invariance."""
    model, _ = build(4, 1280, 20, seed=7)
    toks = synth_tokens(4, 1022, seed=5).cuda()
    with torch.no_grad():
        a = model(toks, repr_layers=[4])
        b = model(toks, repr_layers=[4])
        assert torch.equal(a["logits"], b["logits"]) and torch.equal(a["representations"][4], b["representations"][4])
        one = model(toks[2:3], repr_layers=[4])
        assert torch.equal(one["representations"][4][0], a["representations"][4][2])
        # right-padding a shorter sequence must not change its non-pad outputs
        short = toks[:1, :300].clone()
        short[0, 299] = 2
        padded = torch.full((1, 1024), 1, dtype=torch.int64, device="cuda")
        padded[0, :300] = short[0]
        s = model(short, repr_layers=[4])["representations"][4]
        p = model(padded, repr_layers=[4])["representations"][4][:, :300]
        assert rel_err(p.cpu(), s.cpu()) < 1e-5
        # argmax of logits is a pure function of the sequence
        assert torch.equal(model(short)["logits"].argmax(-1), model(padded)["logits"][:, :300].argmax(-1))


def test_half_and_bf16_models():
    """model.half() (as ESMFold does, reference esm/esmfold/v1/esmfold.py:62) keeps working: outputs
    come back in the model dtype; bf16 operands are selectable."""
    model, sd = build(2, 128, 2, 21)
    toks = synth_tokens(2, 30, seed=2)
    ref = esm2_forward(sd, toks, 2, 2, repr_layers=[2])
    out32 = model(toks.cuda(), repr_layers=[2])
    fold32 = C.fold_of(model)
    mh = model.half()
    out16 = mh(toks.cuda(), repr_layers=[2])
    assert out16["logits"].dtype == torch.float16 and out16["representations"][2].dtype == torch.float16
    assert rel_err(out16["representations"][2].float().cpu(), ref["representations"][2]) < 3e-3
    C.check_tensors("fp32 model, fp16 operands", out32["representations"][2].cpu(), ref["representations"][2],
                    C.floor_forward(sd, toks, 2, 2, fold=fold32, repr_layers=[2])["representations"][2])
    mb = mh.bfloat16()
    outb = mb(toks.cuda(), repr_layers=[2])
    assert outb["logits"].dtype == torch.bfloat16
    assert rel_err(outb["representations"][2].float().cpu(), ref["representations"][2]) < 3e-2


def test_sequences_longer_than_1024_tokens():
    """ESM-2 has rotary positions, so nothing in the reference limits the length (esm2.py:77-147; extract.py truncates to
    1022 only by its own default).  A 4100-token sequence next to a 1500-token one: 65 key tiles, a padded batch whose
    second sequence skips 41 all-pad tiles, RoPE tables beyond the first 1024 rows; with contacts on a 2100-token batch."""
    L, E, H = 2, 128, 2
    model, sd = build(L, E, H, seed=31)
    toks = synth_tokens(2, 4098, seed=5)
    toks[1, 1499] = 2
    toks[1, 1500:] = 1
    with torch.no_grad():
        out = model(toks.cuda(), repr_layers=[L])
        pk = model.forward_varlen(toks.cuda(), repr_layers=[L], min_saving=None)
    ref = esm2_forward(sd, toks, L, H, repr_layers=[L])
    floor = C.floor_forward(sd, toks, L, H, model=model, repr_layers=[L])
    nonpad = toks.ne(1)
    assert torch.isfinite(out["representations"][L][nonpad.cuda()]).all()
    C.check_tensors("T=4100 repr", out["representations"][L].cpu(), ref["representations"][L], floor["representations"][L], nonpad)
    C.check_tensors("T=4100 logits", out["logits"].cpu(), ref["logits"], floor["logits"], nonpad)
    assert torch.equal(pk["representations"][L][nonpad.cuda()], out["representations"][L][nonpad.cuda()])
    toks = synth_tokens(1, 2098, seed=6)
    with torch.no_grad():
        outc = model(toks.cuda(), repr_layers=[L], return_contacts=True)
        fused = model.predict_contacts(toks.cuda())
    refc = esm2_forward(sd, toks, L, H, repr_layers=[L], return_contacts=True)
    assert outc["contacts"].shape == refc["contacts"].shape == (1, 2098, 2098)
    assert (outc["contacts"].cpu() - refc["contacts"]).abs().max().item() < 5e-3
    assert (fused.cpu() - outc["contacts"].cpu()).abs().max().item() < 1e-4
