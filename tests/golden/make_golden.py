"""Generate the golden fixtures under tests/golden/ by running the REFERENCE implementation
(/root/reference/esm, imported read-only) on seeded synthetic weights.

    python tests/golden/make_golden.py        # only works where /root/reference is mounted

Each fixture stores the token matrix, the model dimensions / synthetic-weight seed (weights are
regenerated deterministically by esm_amd.synth, a checksum guards against generator drift) and the
reference outputs in fp32.  The fixtures pin both the oracle (tests/test_oracle.py, CPU) and the
HIP engine (tests/test_model_gpu.py, MI355X).
"""
import importlib
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"

CASES = {
    # name: dims, seed, tokens builder
    "tiny_d64": dict(L=2, E=128, H=2, seed=11, T=20, B=3, keep_attn=True),
    "mid_d64": dict(L=3, E=256, H=4, seed=12, T=70, B=2, keep_attn=True),
    "t6_8M_dims": dict(L=6, E=320, H=20, seed=13, T=24, B=2),
    "nopad_d64": dict(L=2, E=128, H=2, seed=14, T=130, B=2, nopad=True),
    # two 128-token blocks; row 1 carries an <eos> in the MIDDLE with residues after it (masked out of every
    # contact channel, modules.py:340-343), row 0 ends early (eos + pads)
    "eosmid_d64": dict(L=2, E=128, H=2, seed=15, T=150, B=2, eosmid=True),
}


MSA_CASES = {
    # name: dims, seed, MSA shape [B,R,C]; pads: a shorter row and a fully padded trailing column block
    "tiny_d64": dict(L=2, E=128, H=2, F=256, seed=21, B=2, R=5, C=19, pads=True),
    "mid_d64": dict(L=2, E=192, H=3, F=384, seed=22, B=1, R=12, C=70, pads=False),
    "onerow_d64": dict(L=1, E=128, H=2, F=256, seed=23, B=1, R=1, C=33, pads=False),
}


def build_msa_tokens(B, R, C, seed, pads):
    g = torch.Generator().manual_seed(seed)
    toks = torch.randint(4, 24, (B, R, C), generator=g, dtype=torch.int64)
    toks[torch.rand((B, R, C), generator=g) < 0.08] = 30  # gaps
    toks[:, :, 0] = 0
    if pads:
        toks[0, :, C - 3:] = 1      # MSA 0 is 3 columns shorter (batch padding)
        toks[0, 2, 4] = 32          # a <mask>
        if B > 1:
            toks[1, R - 1, :] = 1   # MSA 1 has one row less (batch padding in depth)
    return toks


def build_tokens(B, T, seed, nopad=False):
    g = torch.Generator().manual_seed(seed)
    toks = torch.randint(4, 24, (B, T), generator=g, dtype=torch.int64)
    toks[:, 0] = 0
    toks[:, -1] = 2
    if not nopad:
        # sequence 1 is shorter: eos then pads; a few <mask> tokens; sequence 0 gets a gap and X
        if B > 1:
            short = T - max(3, T // 4)
            toks[1, short] = 2
            toks[1, short + 1:] = 1
        toks[0, 3] = 32
        toks[0, 7] = 32
        if B > 1:
            toks[1, 2] = 32
        toks[0, 5] = 30
        toks[0, 6] = 24
        if B > 2:
            toks[2, T // 2] = 1  # an interior pad (tests/test_load_all.py feeds token 1 mid-row)
    return toks


def main():
    sys.path.insert(0, ROOT)
    from esm_amd.synth import synth_esm2_state_dict

    # import the reference package under its own name from /root/reference
    sys.path.insert(0, REFERENCE)
    for k in [k for k in sys.modules if k == "esm" or k.startswith("esm.")]:
        del sys.modules[k]
    ref = importlib.import_module("esm")
    assert ref.__file__.startswith(REFERENCE), ref.__file__

    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]  # e.g. --only=eosmid_d64
    for name, c in CASES.items():
        if only and name not in only:
            continue
        sd = synth_esm2_state_dict(c["L"], c["E"], c["H"], seed=c["seed"])
        model = ref.ESM2(num_layers=c["L"], embed_dim=c["E"], attention_heads=c["H"], alphabet="ESM-1b",
                         token_dropout=True).eval()
        model.load_state_dict(sd, strict=True)
        toks = build_tokens(c["B"], c["T"], c["seed"], c.get("nopad", False) or c.get("eosmid", False))
        if c.get("eosmid"):
            toks[1, 60] = 2
            toks[1, 20] = 32
            toks[0, 99] = 2
            toks[0, 100:] = 1
        with torch.no_grad():
            out = model(toks, repr_layers=list(range(c["L"] + 1)), return_contacts=True)
        fix = {
            "dims": {k: c[k] for k in ("L", "E", "H", "seed")},
            "tokens": toks,
            "weights_checksum": float(sum(v.double().sum() for k, v in sd.items() if k != "lm_head.weight")),
            "logits": out["logits"].float(),
            "representations": {k: v.float() for k, v in out["representations"].items()},
            # full attention maps only for the small cases (file size); contacts pin the rest
            "attentions": out["attentions"].float() if c.get("keep_attn") else None,
            "contacts": out["contacts"].float(),
            "reference_version": getattr(ref, "__version__", "?"),
            "torch_version": torch.__version__,
        }
        path = os.path.join(HERE, f"esm2_{name}.pt")
        torch.save(fix, path)
        print(name, "->", path, os.path.getsize(path) // 1024, "KiB")


def main_msa():
    import argparse

    sys.path.insert(0, ROOT)
    from esm_amd.synth import synth_msa_state_dict

    sys.path.insert(0, REFERENCE)
    for k in [k for k in sys.modules if k == "esm" or k.startswith("esm.")]:
        del sys.modules[k]
    ref = importlib.import_module("esm")
    assert ref.__file__.startswith(REFERENCE), ref.__file__
    alphabet = ref.Alphabet.from_architecture("msa_transformer")
    for name, c in MSA_CASES.items():
        sd = synth_msa_state_dict(c["L"], c["E"], c["H"], c["F"], seed=c["seed"])
        args = argparse.Namespace(layers=c["L"], embed_dim=c["E"], ffn_embed_dim=c["F"], attention_heads=c["H"],
                                  dropout=0.1, attention_dropout=0.1, activation_dropout=0.1, max_positions=1024,
                                  embed_positions_msa=True, embed_positions_msa_dim=c["E"], max_tokens=2 ** 14,
                                  max_tokens_per_msa=2 ** 14)
        model = ref.MSATransformer(args, alphabet).eval()
        model.load_state_dict(sd, strict=True)
        toks = build_msa_tokens(c["B"], c["R"], c["C"], c["seed"], c["pads"])
        with torch.no_grad():
            out = model(toks, repr_layers=list(range(c["L"] + 1)), return_contacts=True)
        fix = {
            "dims": {k: c[k] for k in ("L", "E", "H", "F", "seed")},
            "tokens": toks,
            "weights_checksum": float(sum(v.double().sum() for k, v in sd.items() if k != "lm_head.weight")),
            "logits": out["logits"].float(),
            "representations": {k: v.float() for k, v in out["representations"].items()},
            "row_attentions": out["row_attentions"].float(),
            "col_attentions": out["col_attentions"].float(),
            "contacts": out["contacts"].float(),
            "reference_version": getattr(ref, "__version__", "?"),
            "torch_version": torch.__version__,
        }
        path = os.path.join(HERE, f"msa_{name}.pt")
        torch.save(fix, path)
        print(name, "->", path, os.path.getsize(path) // 1024, "KiB")


ESM1B_CASES = {
    "tiny_d64": dict(L=2, E=128, H=2, seed=31, T=21, B=3, ln_before=True),
    "nolnb_d64": dict(L=2, E=192, H=3, seed=32, T=40, B=2, ln_before=False),
}


def main_esm1b():
    import argparse

    sys.path.insert(0, ROOT)
    from esm_amd.synth import synth_esm1b_state_dict

    sys.path.insert(0, REFERENCE)
    for k in [k for k in sys.modules if k == "esm" or k.startswith("esm.")]:
        del sys.modules[k]
    ref = importlib.import_module("esm")
    assert ref.__file__.startswith(REFERENCE), ref.__file__
    alphabet = ref.Alphabet.from_architecture("roberta_large")
    for name, c in ESM1B_CASES.items():
        sd = synth_esm1b_state_dict(c["L"], c["E"], c["H"], seed=c["seed"], ln_before=c["ln_before"])
        args = argparse.Namespace(arch="roberta_large", layers=c["L"], embed_dim=c["E"], ffn_embed_dim=4 * c["E"],
                                  attention_heads=c["H"], max_positions=1024, token_dropout=True,
                                  emb_layer_norm_before=c["ln_before"])
        model = ref.ProteinBertModel(args, alphabet).eval()
        model.load_state_dict(sd, strict=True)
        toks = build_tokens(c["B"], c["T"], c["seed"])
        with torch.no_grad():
            out = model(toks, repr_layers=list(range(c["L"] + 1)), return_contacts=True)
        fix = {
            "dims": {k: c[k] for k in ("L", "E", "H", "seed", "ln_before")},
            "tokens": toks,
            "weights_checksum": float(sum(v.double().sum() for k, v in sd.items() if k != "lm_head.weight")),
            "logits": out["logits"].float(),
            "representations": {k: v.float() for k, v in out["representations"].items()},
            "attentions": out["attentions"].float(),
            "contacts": out["contacts"].float(),
            "reference_version": getattr(ref, "__version__", "?"),
            "torch_version": torch.__version__,
        }
        path = os.path.join(HERE, f"esm1b_{name}.pt")
        torch.save(fix, path)
        print(name, "->", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    if "--msa-only" in sys.argv:
        main_msa()
    elif "--esm1b-only" in sys.argv:
        main_esm1b()
    else:
        main()
        if not any(a.startswith("--only=") for a in sys.argv):
            main_msa()
        main_esm1b()
