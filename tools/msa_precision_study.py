"""Where does the MSA Transformer (config 5: 12 x 768, 128 x 513) lose accuracy with 16-bit MFMA operands?

CPU study with the fp32 oracle (test infrastructure; this tool is not part of the product): the oracle is re-run
with fp16 (or bf16) rounding injected at chosen points — linear-layer weights, linear-layer inputs, the q / k / v
operands of the attention contractions, the softmax probabilities — and compared with the plain fp32 run.  Also
runs fp32 with a 1e-6 relative perturbation of the input embedding to measure how much the synthetic model amplifies
ANY error (condition of the network, independent of the engine).

    python tools/msa_precision_study.py [--rows 128] [--cols 513] [--layers 12] [--qk_gain 2.0]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esm_amd.synth import MSA_DIMS, synth_msa_state_dict, synth_msa_tokens  # noqa: E402
import oracle.msa_oracle as mo  # noqa: E402


class Inject(mo.OperandRounding):
    """oracle.msa_oracle.OperandRounding + an optional relative perturbation of the token embedding"""

    def __init__(self, weights=None, acts=None, qkv=None, probs=None, perturb=0.0):
        super().__init__(weights, acts, qkv, probs)
        self.perturb = perturb


def run(sd, toks, L, H, inj):
    sd2 = dict(sd)
    if inj.perturb:
        g = torch.Generator().manual_seed(99)
        e = sd["embed_tokens.weight"]
        sd2["embed_tokens.weight"] = e * (1 + inj.perturb * torch.randn(e.shape, generator=g))
    return mo.msa_forward_rounded(inj, sd2, toks, L, H, repr_layers=[L], need_head_weights=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=128)
    ap.add_argument("--cols", type=int, default=513)
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--qk_gain", type=float, default=2.0)
    ap.add_argument("--seed", type=int, default=41)
    a = ap.parse_args()
    _, E, H, Fd = MSA_DIMS["esm_msa1b_t12_100M_UR50S"]
    L = a.layers
    sd = synth_msa_state_dict(L, E, H, Fd, seed=a.seed, qk_gain=a.qk_gain)
    toks = synth_msa_tokens(1, a.rows, a.cols, seed=7)
    h = torch.float16
    cases = {
        "fp32 + 1e-6 relative perturbation of the token embedding": Inject(perturb=1e-6),
        "fp16 weights only": Inject(weights=h),
        "fp16 linear inputs only": Inject(acts=h),
        "fp16 q,k,v + probs only": Inject(qkv=h, probs=h),
        "all of the above (the engine's rounding points)": Inject(weights=h, acts=h, qkv=h, probs=h),
        "all, bf16": Inject(weights=torch.bfloat16, acts=torch.bfloat16, qkv=torch.bfloat16, probs=torch.bfloat16),
    }
    t0 = time.time()
    ref = run(sd, toks, L, H, Inject())
    print(f"fp32 reference run: {time.time() - t0:.0f} s; row attention max per layer (mean over heads / queries): "
          f"{[round(ref['row_attentions'][0, l].amax(-1).mean().item(), 3) for l in range(L)]}", flush=True)
    for name, inj in cases.items():
        out = run(sd, toks, L, H, inj)
        r, rr = out["representations"][L], ref["representations"][L]
        ra = (out["row_attentions"] - ref["row_attentions"]).abs().amax((0, 2, 3, 4))
        ca = (out["col_attentions"] - ref["col_attentions"]).abs().amax((0, 2, 3, 4, 5))
        print(json.dumps({"case": name, "repr_rel_max": ((r - rr).abs().max() / rr.abs().max()).item(),
                          "repr_rel_l2": ((r - rr).norm() / rr.norm()).item(),
                          "row_attn_err_per_layer": [round(v, 5) for v in ra.tolist()],
                          "col_attn_err_first_last": [round(ca[0].item(), 5), round(ca[-1].item(), 5)]}), flush=True)


if __name__ == "__main__":
    main()
