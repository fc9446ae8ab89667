"""Operand floors of the full-size config-3 fixtures (tests/golden/large_esm2_3b_*.pt), CPU only.

    python tests/golden/make_floors.py [--only esm2_3b_T258|esm2_3b_padded]      ->  tests/golden/operand_floors.json

The parity contract (tests/_contract.py) bounds an output's max-norm error by max(1e-3, SLACK x the error of the
emulated 16-bit-operand FLOOR on the same inputs): the fp32 oracle (oracle/esm2_oracle.py) re-run with every MFMA operand
(weights, GEMM inputs, q / k, v, P) rounded to fp16 before it enters a contraction — what ANY engine that feeds 11-bit
operands to fp32-accumulating matrix cores computes at best.  For the 650M-dims tests the floor is computed inside the
test (seconds); for the 3B-dims fixtures it costs minutes and tens of GB, so it is computed here once and committed.

The padded (1022, 300) batch is run one sequence at a time (the contact head has no cross-batch term and padding is
masked: the floor of a sequence does not depend on its batch mates; the batch as a whole needs > 40 GB on the CPU).
Quantities are those tests/test_fullsize_gpu.py compares: strided representation rows (`slim_esm2`), logits over the
non-pad positions, contact logits relative to the range of the unsaturated reference logits.
"""
import argparse
import importlib.util
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
OUT = os.path.join(HERE, "operand_floors.json")


def _gen():
    spec = importlib.util.spec_from_file_location("make_golden_large", os.path.join(HERE, "make_golden_large.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def contact_logit_rel(c, cr):
    lg = lambda t: torch.logit(t.double().clamp(1e-12, 1 - 1e-12))
    z, zr = lg(c), lg(cr)
    ok = zr.abs() < 12
    return ((z - zr)[ok].abs().max() / zr[ok].abs().max()).item()


def floors_of(sd, toks, lengths, L, H, GEN):
    """{form: [per-sequence floor numbers]} for the plain form and the LayerNorm-fold form ("FOLD" injection: what the engine's
    default mode rounds) — ONE fp32 reference run, two floor runs."""
    from oracle.esm2_oracle import ALL_OPERANDS, esm2_forward

    ref = esm2_forward(sd, toks, L, H, repr_layers=[L], return_contacts=True)
    ref = {"logits": ref["logits"], "representations": ref["representations"], "contacts": ref["contacts"]}
    b = GEN.slim_esm2(ref, toks, lengths, L)
    nonpad = toks.ne(1)
    res = {}
    for form, kinds in (("", ALL_OPERANDS), ("@fold", ALL_OPERANDS + ("FOLD",))):
        fl = esm2_forward(sd, toks, L, H, repr_layers=[L], return_contacts=True, inject=(frozenset(kinds), torch.float16))
        fl = {"logits": fl["logits"], "representations": fl["representations"], "contacts": fl["contacts"]}
        a = GEN.slim_esm2(fl, toks, lengths, L)
        out = []
        for i in range(toks.shape[0]):
            same = fl["logits"][i][nonpad[i]].argmax(-1) == ref["logits"][i][nonpad[i]].argmax(-1)
            out.append({"repr_max": rel(a["repr"][i], b["repr"][i]), "repr_l2": rel_l2(a["repr"][i], b["repr"][i]),
                        "logits_max": rel(fl["logits"][i][nonpad[i]], ref["logits"][i][nonpad[i]]),
                        "logits_l2": rel_l2(fl["logits"][i][nonpad[i]], ref["logits"][i][nonpad[i]]),
                        "argmax_raw": same.double().mean().item(),
                        "contact_logit_rel": contact_logit_rel(a["contacts"][i], b["contacts"][i])})
        res[form] = out
        del fl, a
    return res


def msa_floor(case, GEN):
    """Config 5 (one 128 x 513 MSA): the MSA model's operand floor (oracle/msa_oracle.py msa_operand_floor) against the
    reference fixture, in the quantities tests/test_fullsize_gpu.py compares (its _msa_compare)."""
    from esm_amd.synth import synth_msa_state_dict
    from oracle.msa_oracle import msa_operand_floor

    fix = torch.load(os.path.join(HERE, f"large_{case}.pt"), weights_only=False)
    d = fix["dims"]
    sd = synth_msa_state_dict(d["L"], d["E"], d["H"], d["F"], seed=d["seed"], qk_gain=d.get("qk_gain", 2.0))
    t0 = time.time()
    with torch.no_grad():
        out = msa_operand_floor(sd, fix["tokens"].to(torch.int64), d["L"], d["H"], repr_layers=[d["L"]], return_contacts=True)
    got = GEN.slim_msa(out, d["L"])
    del out
    r = {"repr_row0_max": rel(got["repr_row0"], fix["repr_row0"]), "repr_row0_l2": rel_l2(got["repr_row0"], fix["repr_row0"]),
         "repr_sub_max": rel(got["repr_sub"], fix["repr_sub"]), "repr_sub_l2": rel_l2(got["repr_sub"], fix["repr_sub"]),
         "logits_row0_max": rel(got["logits_row0"], fix["logits_row0"]), "logits_row0_l2": rel_l2(got["logits_row0"], fix["logits_row0"]),
         "argmax_raw": (got["logits_argmax"] == fix["logits_argmax"]).double().mean().item(),
         "row_maps": max((got["row_maps"][k] - v).abs().max().item() for k, v in fix["row_maps"].items()),
         "row_max": (got["row_max"] - fix["row_max"]).abs().max().item(),
         "col_maps": max((got["col_maps"][k] - v).abs().max().item() for k, v in fix["col_maps"].items()),
         "contacts_prob": (got["contacts"].double() - fix["contacts"].double()).abs().max().item(),
         "contacts_logit_rel": contact_logit_rel(got["contacts"], fix["contacts"])}
    print(case, r, f"{time.time() - t0:.0f} s", flush=True)
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    from esm_amd.synth import synth_esm2_state_dict

    GEN = _gen()
    res = json.load(open(OUT)) if os.path.exists(OUT) else {}
    res["_what"] = ("max-norm / L2 error of the fp32 oracle with fp16 rounding injected at every MFMA operand (W, A, QK, V, P) "
                    "against the fp32 oracle, per sequence of the full-size config-3 fixtures; tests/golden/make_floors.py")
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    for case in ("msa1b_config5_g1", "msa1b_config5"):
        if a.only and a.only != case:
            continue
        res[case] = [msa_floor(case, GEN)]
        json.dump(res, open(OUT, "w"), indent=1)
    for case in ("esm2_3b_T258", "esm2_3b_padded"):
        if a.only and a.only != case:
            continue
        fix = torch.load(os.path.join(HERE, f"large_{case}.pt"), weights_only=False)
        d = fix["dims"]
        sd = synth_esm2_state_dict(d["L"], d["E"], d["H"], seed=d["seed"])
        toks, lengths = GEN.esm2_3b_tokens(case)
        t0 = time.time()
        rows = {"": [], "@fold": []}
        for i in range(toks.shape[0]):  # one sequence at a time, cut to its own length (see the module docstring)
            n = lengths[i] + 2
            for form, r in floors_of(sd, toks[i:i + 1, :n].clone(), [lengths[i]], d["L"], d["H"], GEN).items():
                rows[form] += r
                print(case + form, i, r[-1], f"{time.time() - t0:.0f} s", flush=True)
        for form, r in rows.items():  # "<case>" = plain form, "<case>@fold" = the LayerNorm-fold form (the engine's default mode)
            res[case + form] = r
        json.dump(res, open(OUT, "w"), indent=1)


if __name__ == "__main__":
    main()
