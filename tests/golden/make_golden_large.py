"""Full-size golden fixtures for BASELINE configs 3 and 5, produced by the REFERENCE implementation
(/root/reference/esm imported read-only; only runs where it is mounted).

    python tests/golden/make_golden_large.py [--only=msa1b_config5|msa1b_config5_g1|esm2_3b_T258|esm2_3b_padded]

The reference needs minutes (MSA config 5: ~5 min, 3B at T=1024 with attention maps: ~15 min on 8 vCPU) and
tens of GB, and its outputs are GBs (`attentions [2,36,40,1024,1024]`), so the fixtures are SLIM: full contact
maps / strided slices of the representations / a few whole attention maps / seeded random projections of every
attention map ("checksum of all entries").  `slim()` below is the single definition of what is stored; the GPU
tests (tests/test_fullsize_gpu.py) apply the same function to the engine's outputs.

Weights and tokens are regenerated deterministically by esm_amd.synth (checksum stored).

3B padded batch (lengths 1022 and 300): `ESM2.forward(return_contacts=True)` on the whole batch needs
> 62 GB in the reference (attentions 12 GB + 4 apc/symmetrize copies), so the batch runs ONCE through the
reference model with `need_head_weights=True` (padding mask, batch-level code path) and the reference's own
`model.contact_head(tokens[i:i+1], attentions[i:i+1])` is applied per sequence; `ContactPredictionHead` has no
cross-batch term (modules.py:338-357; checked on a small batch below: equal to 1 ulp).
"""
import argparse
import importlib
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"


def projection_vectors(n, seed=1234):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, generator=g, dtype=torch.float32)


# ------------------------------------------------------------------------------------------------
# what is stored (shared with tests/test_fullsize_gpu.py)
# ------------------------------------------------------------------------------------------------
def slim_esm2(out, tokens, lengths, L):
    """out: dict with logits [B,T,V], representations {L: [B,T,E]}, contacts [B,T-2,T-2] (fp32 CPU)."""
    B, T = tokens.shape
    fix = {"logits": out["logits"].float().clone()}
    rep = out["representations"][L].float()
    fix["repr_stride"] = 4
    fix["repr"] = [rep[b, : lengths[b] + 2 : 4].clone() for b in range(B)]
    c = out["contacts"].float()
    # sequences filling the row: every 4th row of the map; shorter ones: the whole valid block
    fix["contacts"] = [c[b, ::4].clone() if lengths[b] == T - 2 else c[b, : lengths[b], : lengths[b]].clone()
                       for b in range(B)]
    return fix


def slim_msa(out, L):
    """out: logits [1,R,C,V], representations {L: [1,R,C,D]}, row_attentions [1,L,H,C,C], col_attentions
    [1,L,H,C,R,R] or None, contacts [1,C-1,C-1]."""
    lg = out["logits"].float()
    top2 = lg.topk(2, dim=-1).values
    fix = {
        "logits_row0": lg[0, 0].clone(),
        "logits_argmax": lg.argmax(-1).to(torch.int8),
        "logits_margin": (top2[..., 0] - top2[..., 1]).half(),
        "repr_row0": out["representations"][L][0, 0].float().clone(),
        "repr_sub": out["representations"][L][0, ::32, ::8].float().clone(),
        "contacts": out["contacts"].float().clone(),
    }
    ra = out["row_attentions"].float()
    C = ra.shape[-1]
    fix["row_maps"] = {(l, h): ra[0, l, h].clone() for l, h in ((0, 0), (L - 1, 0), (L - 1, ra.shape[2] - 1))}
    fix["row_proj"] = ra[0] @ projection_vectors(C).to(ra.device)  # [L,H,C]: every entry of every map, weighted
    fix["row_max"] = ra[0].amax(-1)
    ca = out.get("col_attentions")
    if ca is not None:
        ca = ca.float()
        R = ca.shape[-1]
        fix["col_maps"] = {(L - 1, 0, c): ca[0, L - 1, 0, c].clone() for c in (0, 1, C // 2, C - 1)}
        w = projection_vectors(R * R, seed=4321).view(R, R).to(ca.device)
        fix["col_proj"] = torch.stack([(ca[0, l] * w).sum((-1, -2)) for l in range(ca.shape[1])])  # [L,H,C]
    return fix


# ------------------------------------------------------------------------------------------------
def import_reference():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, REFERENCE)
    for k in [k for k in sys.modules if k == "esm" or k.startswith("esm.")]:
        del sys.modules[k]
    ref = importlib.import_module("esm")
    assert ref.__file__.startswith(REFERENCE), ref.__file__
    return ref


def checksum(sd):
    return float(sum(v.double().sum() for k, v in sd.items() if k != "lm_head.weight"))


def esm2_3b_tokens(case):
    from esm_amd.synth import synth_tokens

    if case == "esm2_3b_T258":
        return synth_tokens(1, 256, seed=5), [256]
    toks = synth_tokens(2, 1022, seed=6)
    toks[1, 301] = 2
    toks[1, 302:] = 1
    return toks, [1022, 300]


def make_esm2_3b(ref, case):
    from esm_amd.synth import synth_esm2_state_dict

    L, E, H, seed = 36, 2560, 40, 2
    t0 = time.time()
    sd = synth_esm2_state_dict(L, E, H, seed=seed)
    cs = checksum(sd)
    model = ref.ESM2(num_layers=L, embed_dim=E, attention_heads=H, alphabet="ESM-1b", token_dropout=True).eval()
    model.load_state_dict(sd, strict=True)
    del sd
    toks, lengths = esm2_3b_tokens(case)
    with torch.no_grad():
        if toks.shape[0] == 1:
            out = model(toks, repr_layers=[L], return_contacts=True)
            out = {"logits": out["logits"], "representations": out["representations"], "contacts": out["contacts"]}
        else:
            res = model(toks, repr_layers=[L], need_head_weights=True)
            att = res.pop("attentions")
            contacts = torch.cat([model.contact_head(toks[i:i + 1], att[i:i + 1]) for i in range(toks.shape[0])])
            del att
            out = {"logits": res["logits"], "representations": res["representations"], "contacts": contacts}
    fix = slim_esm2(out, toks, lengths, L)
    fix.update(dims=dict(L=L, E=E, H=H, seed=seed), tokens=toks.to(torch.int16), lengths=lengths,
               weights_checksum=cs, reference_version=getattr(ref, "__version__", "?"),
               torch_version=torch.__version__, seconds=time.time() - t0)
    return fix


def check_contact_head_is_per_sequence(ref):
    from esm_amd.synth import synth_esm2_state_dict, synth_tokens

    sd = synth_esm2_state_dict(2, 128, 2, seed=3)
    model = ref.ESM2(num_layers=2, embed_dim=128, attention_heads=2, alphabet="ESM-1b", token_dropout=True).eval()
    model.load_state_dict(sd, strict=True)
    toks = synth_tokens(2, 40, seed=4)
    toks[1, 20] = 2
    toks[1, 21:] = 1
    with torch.no_grad():
        full = model(toks, return_contacts=True)
        per = torch.cat([model.contact_head(toks[i:i + 1], full["attentions"][i:i + 1]) for i in range(2)])
    assert (per - full["contacts"]).abs().max().item() < 1e-6  # reduction order of the batched sums only


def make_msa_config5(ref, qk_gain=2.0):
    """qk_gain = 2 (the synthetic-weight default, esm_amd/synth.py) gives extremely sharp tied row attention at
    depth 128 (mean row maximum 0.46 ... 0.71 per layer): a STRESS case in which the network amplifies any
    perturbation ~15x per 12 layers and 16-bit operands cannot hold 1e-3 (tools/msa_precision_study.py).
    qk_gain = 1 is the calibrated case: row maxima 0.03 ... 0.13, no amplification."""
    from esm_amd.synth import MSA_DIMS, synth_msa_state_dict, synth_msa_tokens

    L, E, H, F = MSA_DIMS["esm_msa1b_t12_100M_UR50S"]
    seed = 41
    t0 = time.time()
    sd = synth_msa_state_dict(L, E, H, F, seed=seed, qk_gain=qk_gain)
    alphabet = ref.Alphabet.from_architecture("msa_transformer")
    args = argparse.Namespace(layers=L, embed_dim=E, ffn_embed_dim=F, attention_heads=H, dropout=0.1,
                              attention_dropout=0.1, activation_dropout=0.1, max_positions=1024,
                              embed_positions_msa=True, embed_positions_msa_dim=E, max_tokens=2 ** 14,
                              max_tokens_per_msa=2 ** 14)
    model = ref.MSATransformer(args, alphabet).eval()
    model.load_state_dict(sd, strict=True)
    toks = synth_msa_tokens(1, 128, 513, seed=7)
    with torch.no_grad():
        out = model(toks, repr_layers=[L], return_contacts=True)
    fix = slim_msa(out, L)
    fix.update(dims=dict(L=L, E=E, H=H, F=F, seed=seed, qk_gain=qk_gain), tokens=toks.to(torch.int8),
               weights_checksum=checksum(sd),
               reference_version=getattr(ref, "__version__", "?"), torch_version=torch.__version__,
               seconds=time.time() - t0)
    return fix


def main():
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]
    ref = import_reference()
    check_contact_head_is_per_sequence(ref)
    for case in ("msa1b_config5", "msa1b_config5_g1", "esm2_3b_T258", "esm2_3b_padded"):
        if only and case not in only:
            continue
        if case.startswith("msa1b"):
            fix = make_msa_config5(ref, qk_gain=1.0 if case.endswith("_g1") else 2.0)
        else:
            fix = make_esm2_3b(ref, case)
        path = os.path.join(HERE, f"large_{case}.pt")
        torch.save(fix, path)
        print(case, "->", path, os.path.getsize(path) // 1024, "KiB", f"{fix['seconds']:.0f} s", flush=True)


if __name__ == "__main__":
    main()
