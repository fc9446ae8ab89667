"""Which operand SITES must be wider than fp16?  (Round 6; continues tools/contract_mode_study.py.)  CPU only: the fp32 oracle
(test infrastructure) with fp16 rounding injected everywhere EXCEPT at the named sites — "W!qk" = the q / k projection weights
exact (i.e. split, W_hi + W_lo), "W!v", "W!o", "W!fc1", "W!fc2" likewise; "A!site" = the GEMM input exact at that site.  The LM
head is exact in every arm but `plain` (the split-weight engine modes run it in fp32).

    python tools/split_site_study.py [--cases 650m,3b_T258,3b_300] > profiles/r6_split_site_study.log

Outcome: the weights of the VALUE path (v, out) carry most of the weight-rounding error of representations and logits; q / k
weights act through the softmax and matter for the attention maps (contact logits) only -> engine modes f16x2v (v, out) and
f16x2a (q, k, v, out) next to f16x2 (all).  Contact logits need split activations as well (the "A!" arms)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import contract_mode_study as S  # noqa: E402
from esm_amd.synth import ESM2_DIMS, synth_esm2_state_dict  # noqa: E402
from oracle.esm2_oracle import esm2_forward  # noqa: E402

BASE = {"W", "A", "QK", "V", "P"}
X2 = {"A", "QK", "V", "P"}  # f16x2: all weights exact
ARMS = [
    ("plain", BASE, "same", "1.00"),
    ("W exact: qk", BASE | {"W!qk"}, None, ""),
    ("W exact: v", BASE | {"W!v"}, None, ""),
    ("W exact: o", BASE | {"W!o"}, None, ""),
    ("W exact: fc1", BASE | {"W!fc1"}, None, ""),
    ("W exact: fc2", BASE | {"W!fc2"}, None, ""),
    ("W exact: v,o   (f16x2v)", BASE | {"W!v", "W!o"}, None, "1.12 measured"),
    ("W exact: qk,o", BASE | {"W!qk", "W!o"}, None, ""),
    ("W exact: qk,v,o (f16x2a)", BASE | {"W!qk", "W!v", "W!o"}, None, "1.21 measured"),
    ("W exact: fc1,fc2", BASE | {"W!fc1", "W!fc2"}, None, ""),
    ("W exact: v,o,fc2", BASE | {"W!v", "W!o", "W!fc2"}, None, ""),
    ("W exact: qk,v,o,fc2", BASE | {"W!qk", "W!v", "W!o", "W!fc2"}, None, ""),
    ("W exact: all   (f16x2)", X2, None, "1.58 measured"),
    ("f16x2 + A exact: qk", X2 | {"A!qk"}, None, ""),
    ("f16x2 + A exact: qk,fc1", X2 | {"A!qk", "A!fc1"}, None, ""),
    ("f16x2 + A exact: qk,v,fc1 (LN-fed)", X2 | {"A!qk", "A!v", "A!fc1"}, None, ""),
    ("f16x2 + A exact: qk,v,fc1,o", X2 | {"A!qk", "A!v", "A!fc1", "A!o"}, None, ""),
    ("f16x2 + A exact: all", {"QK", "V", "P"}, None, "~2.7 est."),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="650m,3b_T258,3b_300")
    a = ap.parse_args()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    from esm_amd.synth import synth_tokens

    cases = {"650m": ("esm2_t33_650M_UR50D", 0, synth_tokens(2, 254, seed=1)),
             "3b_T258": ("esm2_t36_3B_UR50D", 2, synth_tokens(1, 256, seed=5)),
             "3b_300": ("esm2_t36_3B_UR50D", 2, synth_tokens(1, 300, seed=7))}
    print(__doc__.split("    python")[0].strip())
    for cname in a.cases.split(","):
        model, seed, toks = cases[cname]
        L, E, H = ESM2_DIMS[model]
        sd = {k: v.float() for k, v in synth_esm2_state_dict(L, E, H, seed=seed).items()}
        nonpad = toks.ne(1)
        t0 = time.time()
        ref = esm2_forward(sd, toks, L, H, repr_layers=[L], return_contacts=True)
        print(f"\n== {cname}: {model} dims, tokens {tuple(toks.shape)}, weight seed {seed}")
        print(f"{'arm':38s} | repr max / L2       | logits max / L2     | argmax  | contact logits / range | cost")
        for name, kinds, ih, cost in ARMS:
            out = esm2_forward(sd, toks, L, H, repr_layers=[L], return_contacts=True, inject=(frozenset(kinds), torch.float16),
                               inject_head="same" if ih == "same" else None)
            rm, rl, lm, ll, am, c = S.metrics(out, ref, L, nonpad)
            print(f"{name:38s} | {rm:.2e} / {rl:.2e} | {lm:.2e} / {ll:.2e} | {am:.5f} | {c:.2e}               | {cost}", flush=True)
        print(f"   ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
