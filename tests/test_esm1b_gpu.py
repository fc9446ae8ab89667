"""ESM-1b / ESM-1v (esm.ProteinBertModel, arch "roberta_large") on the MI355X engine: parity against golden
fixtures produced by the reference and against the oracle at the 650M dimensions."""
import argparse
import glob
import os

import pytest
import torch

import _contract as C
import esm
from esm_amd.synth import synth_esm1b_state_dict, synth_tokens
from oracle.esm1b_oracle import esm1b_forward

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "esm1b_*.pt")))


def rel_err(a, b, mask):
    a, b = a[mask], b[mask]
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def build(L, E, H, seed, ln_before=True):
    args = argparse.Namespace(arch="roberta_large", layers=L, embed_dim=E, ffn_embed_dim=4 * E, attention_heads=H,
                              max_positions=1024, token_dropout=True, emb_layer_norm_before=ln_before)
    sd = synth_esm1b_state_dict(L, E, H, seed=seed, ln_before=ln_before)
    model = esm.ProteinBertModel(args, esm.Alphabet.from_architecture("roberta_large")).eval()
    model.load_state_dict(sd, strict=True)
    return model.cuda(), sd


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_esm1b_engine_matches_reference_fixture(path):
    fix = torch.load(path, weights_only=False)
    d = fix["dims"]
    model, sd = build(d["L"], d["E"], d["H"], d["seed"], d["ln_before"])
    with torch.no_grad():
        out = model(fix["tokens"].cuda(), repr_layers=list(range(d["L"] + 1)), return_contacts=True)
    nonpad = fix["tokens"].ne(1)
    # the ONE parity contract (tests/_contract.py): floor-referenced in both norms on these 2 - 3-layer toy models, the floor
    # in the form the engine runs in (oracle/esm1b_oracle.py `inject`)
    floor = C.floor_forward(sd, fix["tokens"], d["L"], d["H"], model=model, forward=esm1b_forward, repr_layers=list(range(d["L"] + 1)))
    tag = os.path.basename(path)
    for layer, ref in fix["representations"].items():
        if layer == 0:
            assert rel_err(out["representations"][0].cpu(), ref, nonpad) < 1e-5
            continue
        C.check_tensors(f"{tag} repr[{layer}]", out["representations"][layer].cpu(), ref, floor["representations"][layer], nonpad)
    C.check_tensors(f"{tag} logits", out["logits"].cpu(), fix["logits"], floor["logits"], nonpad)
    assert (out["attentions"].cpu() - fix["attentions"]).abs().max().item() < 3e-3
    assert (out["contacts"].cpu() - fix["contacts"]).abs().max().item() < 5e-3


def test_esm1b_650m_dims_against_oracle():
    L, E, H = 3, 1280, 20
    model, sd = build(L, E, H, seed=33)
    toks = synth_tokens(2, 510, seed=12)
    toks[1, 400] = 2
    toks[1, 401:] = 1
    toks[0, 17] = 32
    with torch.no_grad():
        out = model(toks.cuda(), repr_layers=[0, L])
    ref = esm1b_forward(sd, toks, L, H, repr_layers=[0, L])
    nonpad = toks.ne(1)
    assert rel_err(out["representations"][0].cpu(), ref["representations"][0], nonpad) < 1e-5  # fp32 embedding path
    floor = C.floor_forward(sd, toks, L, H, model=model, forward=esm1b_forward, repr_layers=[L])
    C.check_tensors("ESM-1b 650M-dims repr", out["representations"][L].cpu(), ref["representations"][L], floor["representations"][L], nonpad)
    C.check_tensors("ESM-1b 650M-dims logits", out["logits"].cpu(), ref["logits"], floor["logits"], nonpad)
    assert isinstance(model, esm.ProteinBertModel) and model.model_version == "ESM-1b" and model.num_layers == L
    with pytest.raises(ValueError):
        model(torch.zeros((1, 1030), dtype=torch.int64).cuda())  # above max_positions
