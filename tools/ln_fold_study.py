"""CPU study: can the per-layer LayerNorm pass be folded into the GEMM that follows it?

Today:   h = fp16(LayerNorm(x));  y = h . fp16(W)^T + b                      (layernorm_kernel: 7.7 KB per row, 5.7 ms per step)
Folded:  a = fp16(x)  (written by the residual epilogue that produced x),  W' = fp16(gamma * W)  (packed once),
         y = rstd * (a . W'^T - mean * rowsum(W')) + (W . beta + b)          (mean / rstd per row in the epilogue)
The algebra is exact in real arithmetic; what changes is WHAT gets rounded to fp16: the raw residual stream instead of
its normalised value, gamma * W instead of W, and the mean enters through a cancellation.  This script runs the fp32
oracle (test infrastructure, not product) with the usual fp16 operand roundings and either form of the two LayerNorm
-> linear pairs of every layer (q/k/v and fc1), and prints the error of representations[L] / logits against the plain
fp32 run next to the standard fp16-operand floor.

    python tools/ln_fold_study.py [--model esm2_t33_650M_UR50D] [--T 128] [--B 2] [--seeds 2]
"""
import argparse
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esm_amd.synth import ESM2_DIMS, synth_esm2_state_dict, synth_tokens  # noqa: E402
from oracle import esm2_oracle as O  # noqa: E402

H16 = torch.float16
r16 = lambda t: t.to(H16).float()


def folded_linear(x, gamma, beta, w, b):
    """fp16(x) . fp16(gamma W)^T with the LayerNorm statistics applied afterwards (fp32 epilogue arithmetic)."""
    mean = x.mean(-1, keepdim=True)
    rstd = torch.rsqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-5)
    wg = r16(w * gamma[None, :])
    acc = F.linear(r16(x), wg)                       # fp32 accumulation of fp16 products
    s = wg.sum(-1)                                   # column correction, fp32, from the ROUNDED image (consistent)
    return rstd * (acc - mean * s) + (F.linear(beta[None, :], w)[0] + b)


def folded_linear_centered(x, gamma, beta, w, b):
    """The form the engine builds (round 4): the mean subtraction is folded into the weights as well —
    W'' = gamma W - rowmean(gamma W), so that  sum_k x_k W''_nk = sum_k (x_k - mean) (gamma W)_nk  exactly in real
    arithmetic — and the epilogue is ONE fma per element: y = rstd * (fp16(x) . fp16(W'')^T) + (W . beta + b).
    What is left of the mean after rounding W'' is mean * sum_k (fp16(W'')_nk - W''_nk): ~ |mean| / std * 2^-12."""
    rstd = torch.rsqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-5)
    wg = w * gamma[None, :]
    wg = r16(wg - wg.mean(-1, keepdim=True))
    return rstd * F.linear(r16(x), wg) + (F.linear(beta[None, :], w)[0] + b)


FOLD = {1: folded_linear, 2: folded_linear_centered}


def layer(sd, i, x, heads, fold, stats=None):
    p = f"layers.{i}."
    B, T, E = x.shape
    d = E // heads
    inj = (frozenset(O.ALL_OPERANDS), H16)
    g1, b1 = sd[p + "self_attn_layer_norm.weight"], sd[p + "self_attn_layer_norm.bias"]
    pa = p + "self_attn."
    if fold:
        q, k, v = (FOLD[fold](x, g1, b1, sd[pa + n + ".weight"], sd[pa + n + ".bias"]) for n in ("q_proj", "k_proj", "v_proj"))
        if stats is not None:
            stats.append((x.abs().max().item(), (x.mean(-1).abs() / x.std(-1)).max().item()))
    else:
        h = O.layer_norm(x, g1, b1)
        q, k, v = (O._linear(h, sd[pa + n + ".weight"], sd[pa + n + ".bias"], inj) for n in ("q_proj", "k_proj", "v_proj"))
    q = q * d ** -0.5
    q, k, v = (t.view(B, T, heads, d).transpose(1, 2) for t in (q, k, v))
    cos, sin = O.rope_tables(T, d)
    q, k = r16(O.apply_rope(q, cos, sin)), r16(O.apply_rope(k, cos, sin))
    probs = torch.softmax((q @ k.transpose(-1, -2)).float(), dim=-1)
    ctx = (r16(probs) @ r16(v)).transpose(1, 2).reshape(B, T, E)
    x = x + O._linear(ctx, sd[pa + "out_proj.weight"], sd[pa + "out_proj.bias"], inj)
    g2, b2 = sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"]
    if fold:
        f = FOLD[fold](x, g2, b2, sd[p + "fc1.weight"], sd[p + "fc1.bias"])
    else:
        f = O._linear(O.layer_norm(x, g2, b2), sd[p + "fc1.weight"], sd[p + "fc1.bias"], inj)
    return x + O._linear(O.gelu(f), sd[p + "fc2.weight"], sd[p + "fc2.bias"], inj)


@torch.no_grad()
def forward(sd, toks, L, heads, fold, stats=None, offset=0.0):
    # no <mask> tokens / padding in the synthetic sample: token dropout is the constant rescale of esm2.py:86-92
    x = F.embedding(toks, sd["embed_tokens.weight"]) * (1 - 0.15 * 0.8) + offset
    for i in range(L):
        x = layer(sd, i, x, heads, fold, stats)
    return O.layer_norm(x, sd["emb_layer_norm_after.weight"], sd["emb_layer_norm_after.bias"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="esm2_t33_650M_UR50D")
    ap.add_argument("--T", type=int, default=128)
    ap.add_argument("--B", type=int, default=2)
    ap.add_argument("--seeds", type=int, default=2)
    ap.add_argument("--offset", type=float, default=0.0, help="constant added to every channel of the embedding: a residual "
                    "stream whose per-row mean is not small against its spread (stress for the folded mean subtraction)")
    a = ap.parse_args()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    L, E, H = ESM2_DIMS[a.model]
    print(f"{a.model}: representations[{L}], B = {a.B}, T = {a.T}; error vs the fp32 oracle (max norm / L2)")
    for seed in range(a.seeds):
        sd = {k: v.float() for k, v in synth_esm2_state_dict(L, E, H, seed=seed).items()}
        toks = synth_tokens(a.B, a.T, seed=100 + seed)
        ref = O.esm2_forward(sd, toks, L, H, repr_layers=[L])["representations"][L].double()
        rel = lambda t: (((t.double() - ref).abs().max() / ref.abs().max()).item(), ((t.double() - ref).norm() / ref.norm()).item())
        if a.offset:  # the fp32 reference of the shifted stream
            ref = forward(sd, toks, L, H, fold=False, offset=a.offset)  # fp16-operand run: only for the |x| statistics
            inj = O.ALL_OPERANDS
            O.ALL_OPERANDS = ()
            ref = forward({**sd}, toks, L, H, fold=False, offset=a.offset).double() if False else None
            O.ALL_OPERANDS = inj
        plain = forward(sd, toks, L, H, fold=False, offset=a.offset)
        stats = []
        fold = forward(sd, toks, L, H, fold=1, stats=stats, offset=a.offset)
        foldc = forward(sd, toks, L, H, fold=2, offset=a.offset)
        if a.offset:  # no oracle for the shifted stream: the folded forms against the plain fp16-operand run
            d = lambda t: (((t.double() - plain.double()).abs().max() / plain.abs().max()).item(), ((t.double() - plain.double()).norm() / plain.double().norm()).item())
            print(f"seed {seed} (offset {a.offset}): vs the plain fp16-operand run: folded {d(fold)[0]:.2e} / {d(fold)[1]:.2e}; centred fold {d(foldc)[0]:.2e} / {d(foldc)[1]:.2e};"
                  f"  max |mean|/std per row {max(s[1] for s in stats):.2f}")
            continue
        (pm, pl), (fm, fl), (cm, cl) = rel(plain), rel(fold), rel(foldc)
        print(f"seed {seed}: fp16-operand floor {pm:.2e} / {pl:.2e};  LayerNorm folded {fm:.2e} / {fl:.2e}  (x{fl / pl:.2f} in L2);  "
              f"centred fold (1 fma) {cm:.2e} / {cl:.2e} (x{cl / pl:.2f});  stream: max|x| {max(s[0] for s in stats):.1f}, max |mean|/std per row {max(s[1] for s in stats):.2f}")


if __name__ == "__main__":
    main()
