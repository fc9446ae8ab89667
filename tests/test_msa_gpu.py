"""End-to-end parity of the MSA Transformer path (esm.MSATransformer -> esmk_msa_forward) on the MI355X:
against golden fixtures produced by the reference implementation (tests/golden/msa_*.pt) and against
the oracle (oracle/msa_oracle.py) at the 100M model's layer dimensions.

Tolerance: fp16 MFMA operands, fp32 accumulation / residual / LayerNorm / softmax; 'relative' =
max|diff| / max|ref|.  Padded positions are compared too: the reference lets the k/v of padded rows
enter the tied row attention (only q is zeroed, axial_attention.py:85-88), so they are part of the result."""
import argparse
import glob
import os

import pytest
import torch

import _contract as C
import esm
from esm_amd.synth import synth_msa_state_dict, synth_msa_tokens
from oracle.msa_oracle import msa_forward, msa_operand_floor

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "msa_*.pt")))
REL = 2e-3


def rel_err(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def build(L, E, H, F, seed):
    args = argparse.Namespace(layers=L, embed_dim=E, ffn_embed_dim=F, attention_heads=H, dropout=0.1,
                              attention_dropout=0.1, activation_dropout=0.1, max_positions=1024,
                              embed_positions_msa=True, embed_positions_msa_dim=E, max_tokens=2 ** 14,
                              max_tokens_per_msa=2 ** 14)
    alphabet = esm.Alphabet.from_architecture("msa_transformer")
    sd = synth_msa_state_dict(L, E, H, F, seed=seed)
    model = esm.MSATransformer(args, alphabet).eval()
    model.load_state_dict(sd, strict=True)
    return model.cuda(), sd


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_msa_engine_matches_reference_fixture(path):
    fix = torch.load(path, weights_only=False)
    d = fix["dims"]
    model, sd = build(d["L"], d["E"], d["H"], d["F"], d["seed"])
    with torch.no_grad():
        out = model(fix["tokens"].cuda(), repr_layers=list(range(d["L"] + 1)), return_contacts=True)
    # the ONE parity contract (tests/_contract.py) with the MSA model's own operand floor (oracle/msa_oracle.py)
    floor = msa_operand_floor(sd, fix["tokens"], d["L"], d["H"], repr_layers=list(range(d["L"] + 1)))
    tag = os.path.basename(path)
    for layer, ref in fix["representations"].items():
        if layer == 0:
            assert rel_err(out["representations"][0].cpu(), ref) < 1e-5
            continue
        C.check_tensors(f"{tag} repr[{layer}]", out["representations"][layer].cpu(), ref, floor["representations"][layer])
    C.check_tensors(f"{tag} logits", out["logits"].cpu(), fix["logits"], floor["logits"])
    assert (out["row_attentions"].cpu() - fix["row_attentions"]).abs().max().item() < 2e-3
    assert out["col_attentions"].shape == fix["col_attentions"].shape
    assert (out["col_attentions"].cpu() - fix["col_attentions"]).abs().max().item() < 2e-3
    assert (out["contacts"].cpu() - fix["contacts"]).abs().max().item() < 5e-3
    assert out["contacts"].shape == fix["contacts"].shape


@pytest.mark.parametrize("B,R,C_,pads", [(1, 32, 257, False), (2, 7, 65, True), (1, 128, 129, False)])
def test_msa_engine_matches_oracle_at_100M_dims(B, R, C_, pads):
    L, E, H, F = 2, 768, 12, 3072
    model, sd = build(L, E, H, F, seed=31)
    toks = synth_msa_tokens(B, R, C_, seed=3)
    if pads:
        toks[0, :, C_ - 5:] = 1
        toks[1, R - 2:, :] = 1
    with torch.no_grad():
        out = model(toks.cuda(), repr_layers=[0, 1, L], return_contacts=True)
    ref = msa_forward(sd, toks, L, H, repr_layers=[0, 1, L], return_contacts=True)
    ofl = msa_operand_floor(sd, toks, L, H, repr_layers=[1, L])
    assert rel_err(out["representations"][0].cpu(), ref["representations"][0]) < 1e-5
    for l in (1, L):
        C.check_tensors(f"MSA ({B},{R},{C_}) repr[{l}]", out["representations"][l].cpu(), ref["representations"][l], ofl["representations"][l])
    C.check_tensors(f"MSA ({B},{R},{C_}) logits", out["logits"].cpu(), ref["logits"], ofl["logits"])
    # tied row attention sums R*64 fp16 products per score: the probability error grows with the MSA depth
    # (measured 3.5e-3 at R = 128 on sharp synthetic attention maps)
    assert (out["row_attentions"].cpu() - ref["row_attentions"]).abs().max().item() < (2e-3 if R <= 32 else 6e-3)
    # column maps: against the emulated fp16-operand floor of the same inputs (the (2, 7, 65) case has a floor of
    # 1.9e-3 — a fixed 2e-3 sat 5 % above it and flipped with any change of the rounding pattern)
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import msa_precision_study as study

    h16 = torch.float16
    floor = study.run(sd, toks, L, H, study.Inject(weights=h16, acts=h16, qkv=h16, probs=h16))
    col_floor = (floor["col_attentions"] - ref["col_attentions"]).abs().max().item()
    col_err = (out["col_attentions"].cpu() - ref["col_attentions"]).abs().max().item()
    print(f"MSA ({B},{R},{C_}): column maps {col_err:.2e}, floor {col_floor:.2e}")
    assert col_err < max(2e-3, 1.25 * col_floor), (col_err, col_floor)
    # R = 128 with the sharp (qk_gain 2) synthetic weights is the ill-conditioned regime of DESIGN.md §2 already
    # at two layers: 5.3e-3 measured
    assert (out["contacts"].cpu() - ref["contacts"]).abs().max().item() < (5e-3 if R <= 32 else 8e-3)


def test_msa_split_weight_mode():
    """ESM_AMD_OPERAND=f16x2 on the MSA Transformer (esmk_msa_config.weight_split; round 4): every weight matrix of the
    axial layers as W_hi + W_lo, the LM head in fp32 — the weight rounding, the largest single share of the fp16-operand
    error, is gone: the error against the oracle must drop clearly below the plain mode's on the same inputs."""
    L, E, H, F = 4, 768, 12, 3072
    model, sd = build(L, E, H, F, seed=33)
    toks = synth_msa_tokens(1, 32, 257, seed=4)
    toks[0, :, 250:] = 1
    ref = msa_forward(sd, toks, L, H, repr_layers=[L], return_contacts=True)
    errs = {}
    old = os.environ.get("ESM_AMD_OPERAND")
    try:
        for mode in ("f16", "f16x2"):
            os.environ["ESM_AMD_OPERAND"] = mode
            with torch.no_grad():
                out = model(toks.cuda(), repr_layers=[L], return_contacts=True)
            d = out["representations"][L].cpu().double() - ref["representations"][L].double()
            errs[mode] = dict(repr_max=rel_err(out["representations"][L].cpu(), ref["representations"][L]),
                              repr_l2=(d.norm() / ref["representations"][L].double().norm()).item(),
                              logits=rel_err(out["logits"].cpu(), ref["logits"]),
                              row_maps=(out["row_attentions"].cpu() - ref["row_attentions"]).abs().max().item(),
                              contacts=(out["contacts"].cpu() - ref["contacts"]).abs().max().item())
    finally:
        if old is None:
            os.environ.pop("ESM_AMD_OPERAND", None)
        else:
            os.environ["ESM_AMD_OPERAND"] = old
    print(f"MSA 4 x 768, (1, 32, 257): {errs}")
    assert errs["f16x2"]["repr_l2"] < 0.8 * errs["f16"]["repr_l2"], errs
    assert errs["f16x2"]["logits"] < 0.8 * errs["f16"]["logits"], errs
    assert errs["f16x2"]["repr_max"] < REL and errs["f16x2"]["logits"] < REL, errs


def test_msa_config5_full_size_properties():
    """BASELINE config 5: esm_msa1b_t12_100M dimensions, one 128 x 513 MSA.  The CPU oracle needs minutes at this
    size, so the full-size run is checked through size-independent properties: finite outputs, row attention rows
    sum to 1, symmetric contact map in [0,1], and invariance of the query row's outputs' SHAPE / determinism."""
    from esm_amd.synth import MSA_DIMS

    L, E, H, F = MSA_DIMS["esm_msa1b_t12_100M_UR50S"]
    model, _ = build(L, E, H, F, seed=41)
    model.return_col_attentions = False  # 58 GB at this size; covered by the smaller cases
    toks = synth_msa_tokens(1, 128, 513, seed=7).cuda()
    with torch.no_grad():
        out = model(toks, repr_layers=[L], return_contacts=True)
        out2 = model(toks, repr_layers=[L], return_contacts=True)
    assert out["representations"][L].shape == (1, 128, 513, E) and out["logits"].shape == (1, 128, 513, 33)
    assert torch.isfinite(out["representations"][L]).all() and torch.isfinite(out["logits"]).all()
    ra = out["row_attentions"]
    assert ra.shape == (1, L, H, 513, 513)
    assert (ra.sum(-1) - 1).abs().max().item() < 1e-4
    c = out["contacts"]
    assert c.shape == (1, 512, 512) and (c - c.transpose(1, 2)).abs().max().item() < 1e-5
    assert c.min().item() >= 0 and c.max().item() <= 1
    assert torch.equal(out["logits"], out2["logits"])  # deterministic (no atomics on the path)


def test_msa_surface():
    model, _ = build(1, 128, 2, 256, seed=5)
    assert isinstance(model, esm.MSATransformer) and model.num_layers == 1
    with pytest.raises(RuntimeError):
        model(torch.zeros((1, 2, 8), dtype=torch.int64))  # CPU tensor: no fallback
    model.max_tokens_per_msa_(1 << 20)
    toks = synth_msa_tokens(1, 3, 9, seed=1).cuda()
    out = model(toks)
    assert out["logits"].shape == (1, 3, 9, 33) and out["representations"] == {}
    assert model.predict_contacts(toks).shape == (1, 8, 8)
