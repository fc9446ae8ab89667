"""Which single changes bring logits / contact logits of the fp16-operand engine under 1e-3?  

CPU only, oracle with rounding injected (test infrastructure; no engine involved).  Arms:
  all            every operand group rounded to fp16, stack and LM head            (today's default mode)
  head exact     stack rounded, LM head in fp32 (weights and activations)          (gemm32 head)
  head W exact   stack rounded, LM head with exact weights, fp16 activations       (split-weight head)
  maps exact     stack rounded, contact maps from the unrounded q / k              (split-q AND split-k contact sweep)
    python tools/parity_budget_study.py [--models esm2_t33_650M_UR50D,esm2_t36_3B_UR50D] [--T 128] [--seeds 2]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esm_amd.synth import ESM2_DIMS, synth_esm2_state_dict, synth_tokens  # noqa: E402
from oracle.esm2_oracle import ALL_OPERANDS, esm2_forward  # noqa: E402


def metrics(out, ref, L):
    r = (out["representations"][L] - ref["representations"][L]).abs().max() / ref["representations"][L].abs().max()
    lg = (out["logits"] - ref["logits"]).abs().max() / ref["logits"].abs().max()
    am = (out["logits"].argmax(-1) == ref["logits"].argmax(-1)).float().mean()
    z = lambda o: torch.logit(o["contacts"].double().clamp(1e-300, 1 - 1e-16))
    zr = z(ref)
    c = (z(out) - zr).abs().max() / (zr.max() - zr.min())
    return r.item(), lg.item(), am.item(), c.item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", default="esm2_t33_650M_UR50D,esm2_t36_3B_UR50D")
    ap.add_argument("--T", type=int, default=128)
    ap.add_argument("--B", type=int, default=2)
    ap.add_argument("--seeds", type=int, default=2)
    a = ap.parse_args()
    dt = torch.float16
    allk = frozenset(ALL_OPERANDS)
    for model in a.models.split(","):
        L, E, H = ESM2_DIMS[model]
        for seed in range(a.seeds):
            sd = {k: v.float() for k, v in synth_esm2_state_dict(L, E, H, seed=seed).items()}
            toks = synth_tokens(a.B, a.T, seed=100 + seed)
            t0 = time.time()
            ref = esm2_forward(sd, toks, L, H, repr_layers=[L], return_contacts=True)
            arms = [("all", (allk, dt), "same"), ("head exact", (allk, dt), None), ("head W exact", (allk, dt), (frozenset({"A"}), dt)),
                    ("maps exact", (allk | {"MAPX"}, dt), "same")]
            for name, inj, ih in arms:
                out = esm2_forward(sd, toks, L, H, repr_layers=[L], return_contacts=True, inject=inj, inject_head=ih)
                r, lg, am, c = metrics(out, ref, L)
                print(f"{model[5:12]} s{seed} {name:13s}: repr {r:.2e} | logits rel {lg:.2e} argmax {am:.5f} | contact logit / range {c:.2e}", flush=True)
            print(f"   ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
