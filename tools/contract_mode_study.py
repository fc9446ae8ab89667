"""Which operand groups must be wider than fp16 for ALL outputs — representations, logits AND contact logits — to land
inside 1e-3?    CPU only: the fp32 oracle with rounding injected (test infrastructure; no engine).

Arms (what is still rounded to fp16; everything else exact — an "exact" group stands for a split operand, hi + lo, i.e. one
more MFMA pass per product):
  plain               W A QK V P                         today's default mode                         cost 1.00
  f16x2               A QK V P                           split weights (shipped precision mode)       cost 1.57 (measured)
  f16x2+maps          A QK V P, contact maps from exact q / k   + split-q/k contact sweep (3 MFMAs per product there)
  f16x2+Aqk           A(not at q/k proj) QK V P          + split activations at the q / k projection only
  f16x2+Aqk+maps
  f16x2+A             QK V P                             + split activations in every GEMM (4 passes)
  f16x2+A+maps
  f16x2+A+QK          V P                                + split q / k in the attention as well
  w8                  W = fp16 + block-scaled fp8 remainder (v_mfma_scale ..f8f6f4 at 2x the fp16 rate), lo product on
                      fp8 activations; A QK V P fp16     the cheap split: ~1.25 x GEMM time instead of 1.75 x
  w8+maps
The LM head runs in fp32 in every arm but `plain` (as the f16x2 engine does: gemm32).
    python tools/contract_mode_study.py [--cases 650m,3b_T258,3b_300] > profiles/r5_contract_mode_study.log
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esm_amd.synth import ESM2_DIMS, synth_esm2_state_dict, synth_tokens  # noqa: E402
from oracle.esm2_oracle import esm2_forward  # noqa: E402

F16 = torch.float16
ARMS = [  # name, kinds, LM head inject (None = exact), estimated cost of the step relative to plain
    ("plain", {"W", "A", "QK", "V", "P"}, "same", "1.00"),
    ("f16x2", {"A", "QK", "V", "P"}, None, "1.57 (measured)"),
    ("f16x2+maps", {"A", "QK", "V", "P", "MAPX"}, None, "1.57 + contact sweep x3"),
    ("f16x2+Aqk", {"A", "A!qk", "QK", "V", "P"}, None, "~1.85 (q/k projection 4 passes)"),
    ("f16x2+Aqk+maps", {"A", "A!qk", "QK", "V", "P", "MAPX"}, None, "~1.85 + contact sweep x3"),
    ("f16x2+A", {"QK", "V", "P"}, None, "~2.7 (every GEMM 4 passes)"),
    ("f16x2+A+maps", {"QK", "V", "P", "MAPX"}, None, "~2.7 + contact sweep x3"),
    ("f16x2+A+QK", {"V", "P"}, None, "~3 (+ attention scores 3 passes)"),
    ("w8", {"W8", "A", "QK", "V", "P"}, None, "~1.3 (lo pass at the fp8 rate)"),
    ("w8+maps", {"W8", "A", "QK", "V", "P", "MAPX"}, None, "~1.3 + contact sweep x3"),
]


def metrics(out, ref, L, nonpad):
    def e(a, b):
        a, b = a.double()[nonpad], b.double()[nonpad]
        return ((a - b).abs().max() / b.abs().max()).item(), ((a - b).norm() / b.norm()).item()
    rm, rl = e(out["representations"][L], ref["representations"][L])
    lm, ll = e(out["logits"], ref["logits"])
    am = (out["logits"].argmax(-1) == ref["logits"].argmax(-1))[nonpad].float().mean().item()
    z = lambda o: torch.logit(o["contacts"].double().clamp(1e-12, 1 - 1e-12))
    zr = z(ref)
    ok = zr.abs() < 12
    c = ((z(out) - zr)[ok].abs().max() / zr[ok].abs().max()).item()
    return rm, rl, lm, ll, am, c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="650m,3b_T258,3b_300")
    a = ap.parse_args()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    cases = {"650m": ("esm2_t33_650M_UR50D", 0, synth_tokens(2, 254, seed=1)),
             "3b_T258": ("esm2_t36_3B_UR50D", 2, synth_tokens(1, 256, seed=5)),
             "3b_300": ("esm2_t36_3B_UR50D", 2, synth_tokens(1, 300, seed=7))}
    print(__doc__.split("Arms")[0].strip())
    for cname in a.cases.split(","):
        model, seed, toks = cases[cname]
        L, E, H = ESM2_DIMS[model]
        sd = {k: v.float() for k, v in synth_esm2_state_dict(L, E, H, seed=seed).items()}
        nonpad = toks.ne(1)
        t0 = time.time()
        ref = esm2_forward(sd, toks, L, H, repr_layers=[L], return_contacts=True)
        print(f"\n== {cname}: {model} dims, tokens {tuple(toks.shape)}, weight seed {seed}")
        print(f"{'arm':16s} | repr max / L2      | logits max / L2    | argmax  | contact logits / range | est. cost")
        for name, kinds, ih, cost in ARMS:
            inj = (frozenset(kinds), F16)
            head = inj if ih == "same" else None
            out = esm2_forward(sd, toks, L, H, repr_layers=[L], return_contacts=True, inject=inj, inject_head="same" if ih == "same" else head)
            rm, rl, lm, ll, am, c = metrics(out, ref, L, nonpad)
            ok = "ALL < 1e-3" if max(rm, lm, c) < 1e-3 else ""
            print(f"{name:16s} | {rm:.2e} / {rl:.2e} | {lm:.2e} / {ll:.2e} | {am:.5f} | {c:.2e}               | {cost}  {ok}", flush=True)
        print(f"   ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
