"""The parity contract of the engine — ONE definition, used by every `-m gpu` parity test and quoted in DESIGN.md §2.

North star: "outputs within 1e-3 rel of the CPU reference", rel = max|diff| / max|ref| over non-pad positions.
An engine that feeds 11-bit (fp16) operands to fp32-accumulating matrix cores has an accuracy FLOOR that no kernel work
moves: the fp32 oracle with every MFMA operand (weights, GEMM inputs, q / k, v, P) rounded to fp16 (`oracle.esm2_oracle`
`inject`).  On a 33 - 36-layer stack that floor is 8.6 - 9.6e-4 in L2 and 0.83 - 1.18e-3 in the max norm for the
representations (the max of 10^6 - 10^8 noisy elements scatters +-15 % with the seed and re-rolls with any change of
rounding pattern) and 1.1 - 1.6e-3 for logits / contact logits (profiles/r2_esm2_precision_floor_*.log,
r4_parity_budget_study.log).  Hence:

  representations (the BASELINE metric's output):  rel L2 <= 1e-3, HARD, on every deep (>= 30-layer) configuration;
                                                   rel max <= max(1e-3, SLACK x floor_max on the SAME inputs)
  logits, contact logits, few-layer toy models:    both norms <= max(1e-3, SLACK x the floor's on the same inputs)
                                                   (plain fp16 operands do NOT reach 1e-3 there and the tests say so;
                                                   ESM_AMD_OPERAND=f16x2 does for representations and logits)

The floor is computed IN THE ENGINE'S OWN FORM: with the LayerNorm fold active (the default) the oracle's "FOLD" injection
rounds what the fold rounds (raw rows centred on the previous mean, gamma-folded row-centred weights); in the plain mode the
plain injection.  The two forms have the same number of roundings and the same expected error, but ONE model's weight-rounding
error acts on the row-independent part of its activations (91 % of the energy of the synthetic models' outputs) as a fixed
bias — one draw per (weights, form): fold / plain logits L2 = 1.24 on the 650M test weights, 0.98 on the next seed, 0.92 at
3B (profiles/r6_ln_fold_logits_study.log).  Round 5's single plain-form floor with a 1.25 slack hid that; per-form floors
let the L2 slack go to 1.10.

SLACK_L2 = 1.10, SLACK (max norm) = 1.25 (1.35 on few-layer toy models): measured over the 133 contract lines of the whole
`-m gpu` suite, engine / floor-in-its-own-form (profiles/r6_gpu_tests_contract_lines*.txt): in L2 median 0.998, largest 1.066
(fold mode) / 1.060 (plain mode) — the L2 bound is the one that sees a defect (an un-normalised row, a missed mask, a wrong
rounding point shows as 2 x or more, and in L2); in the max norm median 0.98, the four largest 1.19, 1.23, 1.24, 1.27 — two
realisations of "the maximum of the same noise over the tensor" (different summation orders, fused vs separate roundings),
all four on toy fixtures of 2 - 6 layers with 10^4 - 10^5 elements; every >= 30-layer configuration is inside 1.15.
Contact logits get CONTACT_SLACK = 1.3 (1.5 in round 5): their figure is ONE number per map — the largest logit error over
~10^5 pairs relative to the logit range — and logits of near-saturated probabilities are heavy-tailed.  Against the floor in
the engine's own form, engine / floor over the nine full-size and 3B-dims maps: 0.89 ... 1.05 in the fold mode (the
default), 0.85 ... 1.27 in the plain mode (profiles/r6_gpu_tests_contract_lines*.txt).
Integer outputs (tokens, argmax wherever the reference's top-2 margin exceeds twice the logit error) are exact.
"""
import json
import os

import torch

CONTRACT = 1e-3
SLACK = 1.25      # max norm
SLACK_TOY = 1.35  # max norm on few-layer toy models (the same statistic over 10^4 - 10^5 elements: see above)
SLACK_L2 = 1.10   # L2 norm
CONTACT_SLACK = 1.3
_FLOORS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "operand_floors.json")


def errors(got, ref, mask=None):
    """(rel L2, rel max) of got against ref over `mask`."""
    if mask is not None:
        got, ref = got[mask], ref[mask]
    got, ref = got.double().cpu(), ref.double().cpu()
    d = got - ref
    return (d.norm() / ref.norm().clamp_min(1e-30)).item(), (d.abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def fold_of(model):
    """Does this model's engine run the LayerNorm fold?  (esm_amd.esm2.ESM2.ln_fold_active; the engine is created by the
    first forward, so call this after one.)"""
    active = model.ln_fold_active()
    assert active is not None, "floor form unknown: the model has not run a forward yet"
    return bool(active)


def default_fold(E, H, weight_split=False):
    """The mode an engine created in THIS environment runs in when there is no model object to ask (a subprocess made it):
    ESM_AMD_LN_FOLD / ESMK_LN_FOLD, else the library default (on) wherever the library supports the fold."""
    v = os.environ.get("ESM_AMD_LN_FOLD", os.environ.get("ESMK_LN_FOLD", "1")).strip().lower()
    return v not in ("0", "false", "off", "no") and not weight_split and E // H <= 64


def inject_kinds(fold):
    from oracle.esm2_oracle import ALL_OPERANDS

    return frozenset(ALL_OPERANDS + (("FOLD",) if fold else ()))


def floor_forward(sd, toks, L, H, dtype=torch.float16, model=None, fold=False, forward=None, **kw):
    """The oracle with `dtype` rounding injected at every MFMA operand, in the form the engine of `model` computes in
    (LayerNorm fold or plain; `fold=` when there is no model object): the floor on these inputs."""
    from oracle.esm2_oracle import esm2_forward

    if model is not None:
        fold = fold_of(model)
    return (forward or esm2_forward)(sd, toks, L, H, inject=(inject_kinds(fold), dtype), **kw)


def committed_floor(case, seq=0, fold=False):
    """Floor numbers of a full-size fixture (tests/golden/make_floors.py -> operand_floors.json), in the engine's form."""
    with open(_FLOORS) as f:
        return json.load(f)[case + ("@fold" if fold else "")][seq]


def raw_argmax_agreement(logits, ref_logits, mask=None):
    a, b = logits.float().cpu().argmax(-1), ref_logits.float().cpu().argmax(-1)
    if mask is not None:
        a, b = a[mask], b[mask]
    return (a == b).double().mean().item()


def check_raw_argmax(name, raw, floor_raw, margin=5e-4):
    """Raw token-argmax agreement with the fp32 reference must not be below what the floor itself reaches on the same
    inputs (near-ties flip under ANY fp16-operand engine) minus 0.05 %."""
    print(f"\ncontract {name}: raw argmax agreement {raw:.5f} (floor on the same inputs {floor_raw:.5f}, bound {floor_raw - margin:.5f})")
    assert raw >= floor_raw - margin, (name, raw, floor_raw)


def check(name, l2, mx, floor_l2, floor_mx, hard_l2=False, slack=None, slack_l2=SLACK_L2, deep=None):
    """Assert the contract on measured (l2, mx) given the floor's numbers on the same inputs; prints one line that the
    evidence scripts grep ("contract ...", on a line of its own: pytest -s glues its progress dots to a test's first print)."""
    if slack is None:  # deep = the full-size configurations (BASELINE's models / fixtures); hard_l2 implies it
        slack = SLACK if (hard_l2 if deep is None else deep) else SLACK_TOY
    b_l2 = CONTRACT if hard_l2 else max(CONTRACT, slack_l2 * floor_l2)
    b_mx = max(CONTRACT, slack * floor_mx)
    print(f"\ncontract {name}: L2 {l2:.2e} (floor {floor_l2:.2e}, bound {b_l2:.2e}{' hard' if hard_l2 else ''}), "
          f"max {mx:.2e} (floor {floor_mx:.2e}, x{mx / max(floor_mx, 1e-30):.2f}, bound {b_mx:.2e})")
    assert l2 <= b_l2, (name, "L2", l2, b_l2)
    assert mx <= b_mx, (name, "max norm", mx, b_mx, floor_mx)
    return l2, mx


def check_tensors(name, got, ref, floor, mask=None, hard_l2=False, slack=None, slack_l2=SLACK_L2, deep=None):
    l2, mx = errors(got, ref, mask)
    f_l2, f_mx = errors(floor, ref, mask)
    return check(name, l2, mx, f_l2, f_mx, hard_l2=hard_l2, slack=slack, slack_l2=slack_l2, deep=deep)


def contact_logit_errors(c, cr, sat=12.0):
    """(max logit error, that error relative to the largest unsaturated reference logit) of contact probabilities."""
    lg = lambda t: torch.logit(t.double().cpu().clamp(1e-12, 1 - 1e-12))
    z, zr = lg(c), lg(cr)
    ok = zr.abs() < sat
    if not bool(ok.any()):
        return 0.0, 0.0
    e = (z - zr)[ok].abs().max().item()
    return e, e / zr[ok].abs().max().item()
