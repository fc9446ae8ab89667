"""Where do 16-bit MFMA operands cost ESM-2 its accuracy, and how low can an fp16-operand engine get?

CPU study on the fp32 oracle (test infrastructure; not part of the product): the ESM-2 forward is re-run with fp16
(or bf16) rounding injected at the points where the HIP engine rounds — linear-layer weights (W), linear-layer inputs
(A: LayerNorm output, attention context, GELU output), rotated q / k (QK), v (V), softmax probabilities (P) — one
group at a time and all together, and compared with the plain fp32 run.  "all" is the floor of ANY engine that feeds
fp16 operands to fp32-accumulating matrix cores; the engine's measured error is to be read against it.

With --engine (on a GPU box) the HIP engine runs the same weights and tokens, so its error is printed next to the
floor on identical inputs.

    python tools/esm2_precision_study.py [--model esm2_t33_650M_UR50D] [--T 128] [--B 2] [--seeds 3] [--dtype f16] [--engine]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esm_amd.synth import ESM2_DIMS, synth_esm2_state_dict, synth_tokens  # noqa: E402
from oracle.esm2_oracle import esm2_forward  # noqa: E402


def forward(sd, toks, L, H, inj, dt):
    """representations[L] of the oracle (reference esm/model/esm2.py:77-128) with the operand groups `inj` rounded to
    `dt` (oracle/esm2_oracle.py: `inject`); inj = () is the plain fp32 reference computation"""
    out = esm2_forward(sd, toks, L, H, repr_layers=[L], inject=(frozenset(inj), dt) if inj else None)
    return out["representations"][L]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="esm2_t33_650M_UR50D")
    ap.add_argument("--T", type=int, default=128)
    ap.add_argument("--B", type=int, default=2)
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--quick", action="store_true", help="only the all-roundings floor (and the engine)")
    ap.add_argument("--engine", action="store_true", help="also run the HIP engine (cuda:0) on the same inputs")
    a = ap.parse_args()
    dt = torch.float16 if a.dtype == "f16" else torch.bfloat16
    if a.engine:
        os.environ["ESM_AMD_OPERAND"] = a.dtype
    L, E, H = ESM2_DIMS[a.model]
    groups = [("W",), ("A",), ("QK",), ("V",), ("P",), ("W", "A"), ("QK", "V", "P"), ("W", "A", "QK", "V", "P")]
    if a.quick:
        groups = groups[-1:]
    print(f"{a.model}: {L} x {E} x {H} heads, B = {a.B}, T = {a.T}, operand dtype {a.dtype}; "
          "rel_max = max|d| / max|ref|, rel_l2 = |d|_2 / |ref|_2 of representations[L]")
    rows = {g: [] for g in groups}
    if a.engine:
        rows[("engine",)] = []
    for seed in range(a.seeds):
        sd = {k: v.float() for k, v in synth_esm2_state_dict(L, E, H, seed=seed).items()}
        toks = synth_tokens(a.B, a.T, seed=100 + seed)
        with torch.no_grad():
            t0 = time.time()
            ref = forward(sd, toks, L, H, (), dt).double()
            for g in groups:
                got = forward(sd, toks, L, H, g, dt).double()
                d = got - ref
                rows[g].append(((d.abs().max() / ref.abs().max()).item(), (d.norm() / ref.norm()).item()))
            if a.engine:
                import esm

                m = esm.ESM2(L, E, H).eval()
                m.load_state_dict(synth_esm2_state_dict(L, E, H, seed=seed))
                m = m.cuda()
                d = m(toks.cuda(), repr_layers=[L])["representations"][L].cpu().double() - ref
                rows[("engine",)].append(((d.abs().max() / ref.abs().max()).item(), (d.norm() / ref.norm()).item()))
                del m
        print(f"  seed {seed}: {time.time() - t0:.0f} s", flush=True)
    for g in rows:
        mx = [r[0] for r in rows[g]]
        l2 = [r[1] for r in rows[g]]
        print(f"{'+'.join(g):>12}: rel_max " + " ".join(f"{v:.2e}" for v in mx) + "   rel_l2 " + " ".join(f"{v:.2e}" for v in l2))


if __name__ == "__main__":
    main()
