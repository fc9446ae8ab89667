"""Parity at the FULL size of BASELINE configs 3 and 5, against fixtures produced by the reference implementation
itself (tests/golden/make_golden_large.py -> tests/golden/large_*.pt; slim fixtures: full contact maps, strided
slices of the representations, a few whole attention maps and seeded random projections of EVERY attention map).

    config 3  esm2_t36_3B dims: contacts + representations[36] + logits at T = 258 and on a padded (1022, 300) batch
    config 5  esm_msa1b_t12_100M dims, one 128 x 513 MSA: logits, representations[12], row / column attentions, contacts

GPU tests compare the HIP engine with the fixtures; the CPU tests at the bottom pin the oracle to the same fixtures
(the MSA one runs in the default CPU suite, the 3B one only with ESM_AMD_SLOW_TESTS=1: ~36 GB, minutes).

Tolerances: the parity contract of tests/_contract.py — representations L2 <= 1e-3 hard and max norm <=
max(1e-3, 1.25 x the fp16-operand floor of the same sequence, tests/golden/operand_floors.json); logits and contact
logits floor-referenced.  'rel' = max|diff| / max|ref| over the compared block.
"""
import importlib.util
import os

import pytest
import torch

import _contract as C

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def _load_gen():
    spec = importlib.util.spec_from_file_location("make_golden_large", os.path.join(GOLD, "make_golden_large.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


GEN = _load_gen()


def fixture(name):
    path = os.path.join(GOLD, f"large_{name}.pt")
    assert os.path.exists(path), f"{path} missing: regenerate with tests/golden/make_golden_large.py"
    return torch.load(path, weights_only=False)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def logit(p):
    return torch.logit(p.double().clamp(1e-12, 1 - 1e-12))


def contact_report(c, cr):
    """max probability error, max logit error where the reference is not saturated, and that error relative to
    the reference's logit range (the north star's 'contact-head logits within 1e-3 rel')."""
    c, cr = c.double().cpu(), cr.double().cpu()
    z, zr = logit(c), logit(cr)
    ok = zr.abs() < 12
    zerr = (z - zr)[ok].abs().max().item()
    return (c - cr).abs().max().item(), zerr, zerr / zr[ok].abs().max().item()


def argmax_check(logits, ref_logits, mask=None):
    """(raw argmax agreement, agreement wherever the reference's top-2 margin exceeds twice the logit error, that
    error) over the positions of ``mask`` ONLY — pad positions hold whatever the pad rows computed (the reference
    discards them, SURVEY §8 c), and an error taken over them (4.5e+1 on the padded 3B fixture) made the decided set
    empty and the check vacuous."""
    diff = (logits - ref_logits).abs()
    err = (diff[mask] if mask is not None else diff).max().item()
    top2 = ref_logits.topk(2, dim=-1).values
    decided = (top2[..., 0] - top2[..., 1]) > 2 * err
    same = logits.argmax(-1) == ref_logits.argmax(-1)
    if mask is not None:
        decided, same_m = decided & mask, same[mask]
    else:
        same_m = same
    assert decided.sum().item() > 0, "no position with a decided argmax: the check would be vacuous"
    return same_m.float().mean().item(), bool(same[decided].all()), err


# ------------------------------------------------------------------------------------------------------------
# config 3: ESM-2 3B
# ------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def model_3b():
    import esm
    from esm_amd.synth import skip_param_init, synth_esm2_state_dict

    fix = fixture("esm2_3b_T258")
    d = fix["dims"]
    sd = synth_esm2_state_dict(d["L"], d["E"], d["H"], seed=d["seed"])
    chk = GEN.checksum(sd)
    assert abs(chk - fix["weights_checksum"]) < 1e-6 * abs(chk), "synthetic weight generator drifted"
    with skip_param_init():  # strict load right below
        m = esm.ESM2(d["L"], d["E"], d["H"]).eval()
    m.load_state_dict(sd)
    del sd
    return m.cuda()


def _check_3b(model, name, mode="f16"):
    """mode "f16x2": the split-weight precision mode (ESM_AMD_OPERAND=f16x2 set by the caller) — tighter bounds."""
    fix = fixture(name)
    L = fix["dims"]["L"]
    toks = fix["tokens"].to(torch.int64)
    lengths = fix["lengths"]
    assert torch.equal(toks, GEN.esm2_3b_tokens(name)[0])
    with torch.no_grad():
        out = model(toks.cuda(), repr_layers=[L], return_contacts=True)
        fused = model.predict_contacts(toks.cuda())
    got = GEN.slim_esm2({"logits": out["logits"], "representations": out["representations"],
                         "contacts": out["contacts"]}, toks, lengths, L)
    fz = GEN.slim_esm2({"logits": out["logits"], "representations": out["representations"], "contacts": fused},
                       toks, lengths, L)
    nonpad = toks.ne(1)
    report = {}
    for b in range(toks.shape[0]):
        e_max, e_l2 = rel(got["repr"][b], fix["repr"][b]), rel_l2(got["repr"][b], fix["repr"][b])
        perr, zerr, zrel = contact_report(got["contacts"][b], fix["contacts"][b])
        fperr = (fz["contacts"][b].cpu() - got["contacts"][b].cpu()).abs().max().item()
        report[b] = dict(repr_rel_max=e_max, repr_rel_l2=e_l2, contact_prob=perr, contact_logit=zerr,
                         contact_logit_rel=zrel, fused_vs_materialised=fperr)
    raw, decided_ok, lerr = argmax_check(out["logits"].float().cpu(), fix["logits"], nonpad)
    print(f"\n{name}: {report}; logits abs err {lerr:.2e}, argmax raw {raw:.4f}, decided ok {decided_ok}")
    lrel = lerr / fix["logits"][nonpad].abs().max().item()
    print(f"{name} [{mode}]: logits rel {lrel:.2e}")
    for b, r in report.items():
        if mode == "f16x3":
            # weights and GEMM inputs split: floor of the form (q / k, v, P still fp16) 2.7e-4 / 2.2e-4 on the representations,
            # 7.9e-4 of range on the contact logits (profiles/r6_split_site_study.log): EVERY output inside 1e-3
            assert r["repr_rel_l2"] < 4e-4 and r["repr_rel_max"] < 5e-4, (b, r)
            assert r["contact_logit_rel"] < 1e-3 and r["contact_prob"] < 5e-3, (b, r)
        elif mode == "f16x2":
            # split weights: the emulated floor of this mode (weights exact, fp16 activations / q / k / v / P) is
            # 5.1 - 6.7e-4 (max norm) / 4.9 - 5.6e-4 (L2) on 650M and 3B dimensions (profiles/r3_f16x2_cpu_study.log)
            assert r["repr_rel_l2"] < 7e-4 and r["repr_rel_max"] < 8e-4, (b, r)
            assert r["contact_logit_rel"] < 2e-3 and r["contact_prob"] < 1e-2, (b, r)
        else:
            fl = C.committed_floor(name, b, fold=C.fold_of(model))  # the floor in the form the engine ran in
            C.check(f"{name} seq {b} repr[{L}]", r["repr_rel_l2"], r["repr_rel_max"], fl["repr_l2"], fl["repr_max"], hard_l2=True)
            C.check(f"{name} seq {b} contact logits", r["contact_logit_rel"], r["contact_logit_rel"], fl["contact_logit_rel"],
                    fl["contact_logit_rel"], slack=C.CONTACT_SLACK, slack_l2=C.CONTACT_SLACK)
            assert r["contact_prob"] < 1e-2, (b, r)
        assert r["fused_vs_materialised"] < 1e-4, (b, r)
    if mode == "f16x3":
        assert lrel < 6e-4 and raw == 1.0, (lrel, raw)
    elif mode == "f16x2":
        assert lrel < 1e-3, lrel  # the contract on the logits, which plain fp16 operands miss (1.1 - 1.5e-3)
    else:
        # plain fp16 operands: the logits carry the representation's error through one more LayerNorm and two GEMMs; the
        # floor itself is 1.1 - 1.6e-3 there (profiles/r4_parity_budget_study.log) — floor-referenced, per sequence
        for b in range(toks.shape[0]):
            fl = C.committed_floor(name, b, fold=C.fold_of(model))
            l2, mx = C.errors(out["logits"][b].float().cpu(), fix["logits"][b], nonpad[b])
            C.check(f"{name} seq {b} logits", l2, mx, fl["logits_l2"], fl["logits_max"], deep=True)
            C.check_raw_argmax(f"{name} seq {b} token argmax", C.raw_argmax_agreement(out["logits"][b], fix["logits"][b], nonpad[b]),
                               fl["argmax_raw"])
    assert decided_ok
    return report, lrel, raw


@pytest.mark.gpu
def test_config3_3b_contacts_T258(model_3b):
    _check_3b(model_3b, "esm2_3b_T258")


@pytest.mark.gpu
def test_config3_3b_contacts_padded_1022_300(model_3b):
    _check_3b(model_3b, "esm2_3b_padded")


@pytest.mark.gpu
def test_config3_3b_T258_split_weight_mode(model_3b, monkeypatch):
    """ESM_AMD_OPERAND=f16x2 (W = W_hi + W_lo, two MFMA passes per layer GEMM): representations AND logits inside
    1e-3 with margin on the full-size config-3 fixture; the engine re-packs its weights when the mode changes."""
    monkeypatch.setenv("ESM_AMD_OPERAND", "f16x2")
    try:
        report, lrel, raw = _check_3b(model_3b, "esm2_3b_T258", mode="f16x2")
    finally:
        monkeypatch.delenv("ESM_AMD_OPERAND")
        model_3b.refresh_engine() if hasattr(model_3b, "refresh_engine") else None


@pytest.mark.gpu
def test_config3_3b_T258_f16x3_every_output_inside_1e3(model_3b, monkeypatch):
    """ESM_AMD_OPERAND=f16x3 on the full-size config-3 fixture: representations, logits AND contact logits inside 1e-3 of the
    reference (the north star's tolerance on every output; the plain fp16 mode sits on a floor of 1.0 / 1.1 / 1.6e-3 there)."""
    monkeypatch.setenv("ESM_AMD_OPERAND", "f16x3")
    try:
        _check_3b(model_3b, "esm2_3b_T258", mode="f16x3")
        _check_3b(model_3b, "esm2_3b_padded", mode="f16x3")   # the padded (1022, 300) batch as well
    finally:
        monkeypatch.delenv("ESM_AMD_OPERAND")
        model_3b.refresh_engine() if hasattr(model_3b, "refresh_engine") else None


# ------------------------------------------------------------------------------------------------------------
# config 5: MSA Transformer, 12 x 768, one 128 x 513 MSA
# ------------------------------------------------------------------------------------------------------------
def _msa_compare(got, fix, tag):
    r = {}
    r["repr_row0_max"], r["repr_row0_l2"] = rel(got["repr_row0"], fix["repr_row0"]), rel_l2(got["repr_row0"], fix["repr_row0"])
    r["repr_sub_max"], r["repr_sub_l2"] = rel(got["repr_sub"], fix["repr_sub"]), rel_l2(got["repr_sub"], fix["repr_sub"])
    r["logits_row0"] = rel(got["logits_row0"], fix["logits_row0"])
    lerr = (got["logits_row0"].cpu() - fix["logits_row0"]).abs().max().item()
    same = got["logits_argmax"].cpu() == fix["logits_argmax"]
    decided = fix["logits_margin"].float() > 4 * lerr  # row-0 error as the scale, doubled
    r["argmax_raw"], r["argmax_decided_ok"] = same.float().mean().item(), bool(same[decided].all())
    r["contacts_prob"], r["contacts_logit"], r["contacts_logit_rel"] = contact_report(got["contacts"], fix["contacts"])
    r["row_maps"] = max((got["row_maps"][k].cpu() - v).abs().max().item() for k, v in fix["row_maps"].items())
    r["row_proj"] = (got["row_proj"].cpu() - fix["row_proj"]).abs().max().item()
    r["row_max"] = (got["row_max"].cpu() - fix["row_max"]).abs().max().item()
    # error of the deepest layer's maps vs the first layer's: the depth dependence VERDICT r1 asked about
    L = fix["row_proj"].shape[0]
    r["row_proj_first_last"] = ((got["row_proj"][0].cpu() - fix["row_proj"][0]).abs().max().item(),
                                (got["row_proj"][L - 1].cpu() - fix["row_proj"][L - 1]).abs().max().item())
    if "col_maps" in got and "col_maps" in fix:
        r["col_maps"] = max((got["col_maps"][k].cpu() - v).abs().max().item() for k, v in fix["col_maps"].items())
        r["col_proj"] = (got["col_proj"].cpu() - fix["col_proj"]).abs().max().item()
    print(f"\nconfig 5 ({tag}): {r}")
    return r


def _msa_model_and_tokens(fix):
    import argparse

    import esm
    from esm_amd.synth import synth_msa_state_dict

    d = fix["dims"]
    sd = synth_msa_state_dict(d["L"], d["E"], d["H"], d["F"], seed=d["seed"], qk_gain=d.get("qk_gain", 2.0))
    chk = GEN.checksum(sd)
    assert abs(chk - fix["weights_checksum"]) < 1e-6 * abs(chk), "synthetic weight generator drifted"
    ns = argparse.Namespace(layers=d["L"], embed_dim=d["E"], ffn_embed_dim=d["F"], attention_heads=d["H"], dropout=0.1,
                            attention_dropout=0.1, activation_dropout=0.1, max_positions=1024, embed_positions_msa=True,
                            embed_positions_msa_dim=d["E"], max_tokens=2 ** 14, max_tokens_per_msa=2 ** 14)
    model = esm.MSATransformer(ns, esm.Alphabet.from_architecture("msa_transformer")).eval()
    model.load_state_dict(sd)
    return model, sd, fix["tokens"].to(torch.int64)


# Two full-size fixtures (tests/golden/make_golden_large.py: make_msa_config5):
#   msa1b_config5_g1  calibrated synthetic weights (qk_gain 1: row-attention maxima 0.03 ... 0.13): the 1e-3 contract
#   msa1b_config5     stress weights (qk_gain 2: mean row maximum 0.46 ... 0.71, tied over 128 rows): the network
#                     itself amplifies any perturbation, fp32 summation order alone moves the maps by 7e-5
#                     (oracle vs reference above) and an fp32 run with fp16-ROUNDED OPERANDS — no engine involved —
#                     is already 5.9e-3 off (tools/msa_precision_study.py; profiles/r2_msa_precision_study.log).
#                     The engine must stay at that fp16-operand floor (bounds = 1.5 x the emulated numbers).
MSA_BOUNDS = {
    # measured (round 2): repr 5.8e-4, logits 9.8e-4, row maps 6.6e-4 (row maximum 1.8e-3), column maps 2.6e-5,
    # contact probability 1.7e-3, contact logit 2.0e-3 of the logit range
    "msa1b_config5_g1": dict(repr_max=1e-3, repr_l2=1e-3, logits=1.5e-3, row_maps=3e-3, col_maps=1e-3,
                             contacts_prob=5e-3, contacts_logit_rel=3e-3),
    "msa1b_config5": dict(repr_max=9e-3, repr_l2=7e-3, logits=1.2e-2, row_maps=8e-2, col_maps=1.7e-2,
                          contacts_prob=1e-1, contacts_logit_rel=6e-2),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(MSA_BOUNDS))
def test_config5_msa_full_size_against_reference_fixture(name):
    fix = fixture(name)
    L = fix["dims"]["L"]
    model, _, toks = _msa_model_and_tokens(fix)
    model = model.cuda()
    with torch.no_grad():
        out = model(toks.cuda(), repr_layers=[L], return_contacts=True)  # row + column attentions, 4.8 GB
    got = GEN.slim_msa(out, L)
    r = _msa_compare(got, fix, f"{name}: HIP engine vs reference fixture")
    b = MSA_BOUNDS[name]
    # the ONE parity contract (tests/_contract.py) against the MSA model's committed operand floor on this fixture
    # (tests/golden/make_floors.py msa_floor); the fixed numbers of MSA_BOUNDS stay as absolute ceilings on top
    fl = C.committed_floor(name)
    C.check(f"{name} repr row 0", r["repr_row0_l2"], r["repr_row0_max"], fl["repr_row0_l2"], fl["repr_row0_max"], deep=True)
    C.check(f"{name} repr rows ::32", r["repr_sub_l2"], r["repr_sub_max"], fl["repr_sub_l2"], fl["repr_sub_max"], deep=True)
    C.check(f"{name} logits row 0", rel_l2(got["logits_row0"], fix["logits_row0"]), r["logits_row0"], fl["logits_row0_l2"], fl["logits_row0_max"], deep=True)
    C.check_raw_argmax(f"{name} token argmax", r["argmax_raw"], fl["argmax_raw"])
    C.check(f"{name} contact logits", r["contacts_logit_rel"], r["contacts_logit_rel"], fl["contacts_logit_rel"], fl["contacts_logit_rel"],
            slack=C.CONTACT_SLACK, slack_l2=C.CONTACT_SLACK)
    assert r["repr_row0_l2"] < b["repr_l2"] and r["repr_sub_l2"] < b["repr_l2"], r
    assert r["repr_row0_max"] < b["repr_max"] and r["repr_sub_max"] < b["repr_max"], r
    assert r["logits_row0"] < b["logits"] and r["argmax_decided_ok"], r
    assert r["row_maps"] < b["row_maps"] and r["row_max"] < b["row_maps"], r
    assert r["col_maps"] < b["col_maps"], r
    assert r["contacts_prob"] < b["contacts_prob"] and r["contacts_logit_rel"] < b["contacts_logit_rel"], r


@pytest.mark.gpu
def test_config5_msa_full_size_split_weight_mode():
    """Config 5 with ESM_AMD_OPERAND=f16x2 (round 4: the MSA engine takes the split-weight mode): against the calibrated
    full-size reference fixture the representation error falls well under the plain mode's (5.2e-4 in L2, 6.1e-4 max)
    and the logits come inside 1e-3 with margin (plain: 9.2e-4 ... 9.8e-4)."""
    name = "msa1b_config5_g1"
    fix = fixture(name)
    L = fix["dims"]["L"]
    model, _, toks = _msa_model_and_tokens(fix)
    model = model.cuda()
    model.return_col_attentions = False
    old = os.environ.get("ESM_AMD_OPERAND")
    os.environ["ESM_AMD_OPERAND"] = "f16x2"
    try:
        with torch.no_grad():
            out = model(toks.cuda(), repr_layers=[L], return_contacts=True)
    finally:
        if old is None:
            os.environ.pop("ESM_AMD_OPERAND", None)
        else:
            os.environ["ESM_AMD_OPERAND"] = old
    got = GEN.slim_msa(out, L)
    r = _msa_compare(got, fix, f"{name} [f16x2]: HIP engine vs reference fixture")
    assert r["repr_row0_l2"] < 4.2e-4 and r["repr_sub_l2"] < 4.2e-4, r
    assert r["repr_row0_max"] < 5.5e-4 and r["repr_sub_max"] < 5.5e-4, r
    assert r["logits_row0"] < 8e-4 and r["argmax_decided_ok"], r


# ------------------------------------------------------------------------------------------------------------
# CPU: the oracle against the same full-size fixtures
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(MSA_BOUNDS))
def test_msa_oracle_matches_full_size_reference_fixture(name):
    from oracle.msa_oracle import msa_forward

    fix = fixture(name)
    L, H = fix["dims"]["L"], fix["dims"]["H"]
    _, sd, toks = _msa_model_and_tokens(fix)
    with torch.no_grad():
        out = msa_forward(sd, toks, L, H, repr_layers=[L], return_contacts=True)
    got = GEN.slim_msa(out, L)
    r = _msa_compare(got, fix, f"{name}: oracle vs reference fixture")
    assert r["repr_row0_max"] < 2e-5 and r["repr_sub_max"] < 2e-5 and r["logits_row0"] < 2e-5, r
    # the reference sums the tied scores over row chunks (max_tokens_per_msa, axial_attention.py:40-73): fp32
    # summation order differs from the oracle's single einsum, visible at 1e-4 on the sharpest maps of layer 12
    assert r["row_maps"] < 3e-4 and r["col_maps"] < 3e-5 and r["contacts_prob"] < 1e-4, r
    assert r["row_proj"] < 1e-3 and r["col_proj"] < 1e-3, r


@pytest.mark.skipif(os.environ.get("ESM_AMD_SLOW_TESTS") != "1", reason="3B oracle: ~36 GB and minutes (ESM_AMD_SLOW_TESTS=1)")
def test_esm2_oracle_matches_3b_reference_fixture():
    from esm_amd.synth import synth_esm2_state_dict
    from oracle.esm2_oracle import esm2_forward

    fix = fixture("esm2_3b_T258")
    d = fix["dims"]
    sd = synth_esm2_state_dict(d["L"], d["E"], d["H"], seed=d["seed"])
    toks = fix["tokens"].to(torch.int64)
    with torch.no_grad():
        out = esm2_forward(sd, toks, d["L"], d["H"], repr_layers=[d["L"]], return_contacts=True)
    got = GEN.slim_esm2(out, toks, fix["lengths"], d["L"])
    assert rel(got["repr"][0], fix["repr"][0]) < 2e-5
    assert (got["contacts"][0] - fix["contacts"][0]).abs().max().item() < 2e-5
    assert rel(got["logits"], fix["logits"]) < 2e-5
