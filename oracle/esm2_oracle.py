"""ORACLE — test infrastructure only.  NOT part of the product.

A plain-PyTorch fp32 CPU restatement of the reference's ESM-2 forward pass, written from the
behavioural spec (SURVEY.md §8 a-1 ... a-14) with the reference location of every step cited.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this module, and only as the *checker*; the shipped path (esm_amd.ESM2 -> libesmk.so) never does.

Pinning: the reference publishes no numeric fixtures for ESM-2 (SURVEY.md §8 c), so this restatement
is pinned against the reference ITSELF: ``tests/golden/make_golden.py`` imports
``/root/reference/esm`` in the build container, runs ``esm.ESM2`` on seeded synthetic weights and
stores its outputs under ``tests/golden/``; ``tests/test_oracle.py`` checks this file against those
fixtures to <= 2e-5.

It works on a state dict (reference key names) — it does not use esm_amd's modules.

``inject`` (default None: nothing is touched, the fp32 reference computation): ``(kinds, dtype)`` rounds the named
operand groups to a 16-bit dtype and back before they enter a contraction — "W" linear-layer weights, "A" linear-layer
inputs, "QK" rotated q / k, "V" values, "P" softmax probabilities ("MAPX": the returned attention maps come from the
unrounded q / k — study hook for a split-q contact sweep; "A!qk": group A is not rounded at the q / k projections (sites qk,
v, o, fc1, fc2); "W8": weights as fp16 + a block-scaled fp8 remainder; ``inject_head``: the LM head's own setting).  With all five it is the accuracy FLOOR of any engine
that feeds 16-bit operands to fp32-accumulating matrix cores; tools/esm2_precision_study.py and the parity tests
read the HIP engine's error against it.

"FOLD" in the kinds: the LayerNorm -> Linear pairs (q / k / v projections, fc1) in the engine's LayerNorm-fold FORM
(esm_amd/csrc/kernels.h GemmArgs::ln_part; same real-arithmetic value as modules.py:113,137 + the nn.Linear behind them):
the A operand is the rounded raw row  fp16(x - m_prev)  (m_prev: the row's mean at the PREVIOUS LayerNorm point — any
per-row constant cancels against centred weights), the weights are  fp16(gamma W - rowmean(gamma W)),  and the value is
rstd * acc + (b + W . beta)  with fp32 mean / rstd.  Same number of roundings as the plain form at different points: the
floor "in the fold's form" (profiles/r6_ln_fold_logits_study.log: the two forms differ by one draw of the weight-rounding
bias per model, not systematically).
"""
import math

import torch
import torch.nn.functional as F


ALL_OPERANDS = ("W", "A", "QK", "V", "P")


def _rnd(t, inject, kind, site=None):
    # "A!qk" in the kinds: group A is NOT rounded at site "qk" (study hook: which sites need wider operands)
    if inject is None or kind not in inject[0] or (site is not None and f"{kind}!{site}" in inject[0]):
        return t
    return t.to(inject[1]).float()


def _mx8(t):
    """Block-scaled fp8 (e4m3, one power-of-two scale per 32 elements of the contraction axis): the operand format of
    v_mfma_scale_f32_16x16x128_f8f6f4 — study hook only (tools/contract_mode_study.py)."""
    shp = t.shape
    K = shp[-1]
    pad = (-K) % 32
    u = F.pad(t, (0, pad)).reshape(*shp[:-1], -1, 32)
    e = torch.floor(torch.log2(u.abs().amax(-1, keepdim=True).clamp_min(1e-38)))
    scale = torch.exp2(e - 7)  # the block maximum lands in [128, 256): inside e4m3's range (448), 3 mantissa bits
    q = (u / scale).to(torch.float8_e4m3fn).float() * scale
    return q.reshape(*shp[:-1], -1)[..., :K]


def _linear(x, w, b, inject, site=None):
    if inject is not None and "W8" in inject[0]:
        # study hook: W = fp16(W) + mx8(W - fp16(W)); the lo product takes block-scaled fp8 activations as well
        hi = w.to(torch.float16).float()
        xa = _rnd(x, inject, "A", site)
        return F.linear(xa, hi, b) + F.linear(_mx8(x), _mx8(w - hi))
    return F.linear(_rnd(x, inject, "A", site), _rnd(w, inject, "W", site), b)


def gelu(x):
    # reference esm/modules.py:17-24 (exact erf form)
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def layer_norm(x, w, b):
    # reference esm/modules.py:68-81: ESM1bLayerNorm == torch.nn.LayerNorm(E), eps 1e-5
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


def rope_tables(T, head_dim, device=None):
    # reference esm/rotary_embedding.py:40-41 (inv_freq) and :47-61 (cos/sin of t x inv_freq,
    # the d/2 frequencies duplicated)
    inv_freq = 1.0 / (10000 ** (torch.arange(0, head_dim, 2, device=device).float() / head_dim))
    t = torch.arange(T, device=device).float()
    freqs = torch.outer(t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def apply_rope(x, cos, sin):
    # reference esm/rotary_embedding.py:11-20; x [..., T, d]
    half = x.shape[-1] // 2
    rot = torch.cat((-x[..., half:], x[..., :half]), dim=-1)
    return x * cos + rot * sin


def attention_layer(sd, prefix, x, heads, pad_mask, need_weights, use_rope=True, inject=None, qkv=None, shape=None):
    """Self-attention of one TransformerLayer on x [B,T,E] (reference works on [T,B,E], the math
    is layout independent).  reference esm/multihead_attention.py:256-261 (projections, q scaling),
    :280-284 (head split), :354-355 (rotary), :357 (scores), :368-374 (key padding -inf),
    :379-380 (fp32 softmax), :387-395 (PV, merge, out_proj)."""
    B, T, E = x.shape if shape is None else shape
    d = E // heads
    p = prefix + "self_attn."
    if qkv is not None:  # the projections in the LayerNorm-fold form (FoldRows.linear); x is not used
        q, k, v = qkv("q") * (d ** -0.5), qkv("k"), qkv("v")
    else:
        q = _linear(x, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"], inject, "qk") * (d ** -0.5)
        k = _linear(x, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"], inject, "qk")
        v = _linear(x, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"], inject, "v")
    q, k, v = (t.view(B, T, heads, d).transpose(1, 2) for t in (q, k, v))  # [B,H,T,d]
    if use_rope:  # ESM-2 (TransformerLayer(use_rotary_embeddings=True), esm2.py:57-66); ESM-1b has none
        cos, sin = rope_tables(T, d, q.device)
        q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
    q_x, k_x = q, k
    q, k = _rnd(q, inject, "QK"), _rnd(k, inject, "QK")
    scores = q @ k.transpose(-1, -2)  # [B,H,T,T]
    if pad_mask is not None:
        scores = scores.masked_fill(pad_mask[:, None, None, :], float("-inf"))
    probs = torch.softmax(scores.float(), dim=-1)
    maps = probs
    if need_weights and inject is not None and "MAPX" in inject[0]:
        # study hook: the RETURNED attention maps (contact head input) from the unrounded q / k of the same stream
        sx = q_x @ k_x.transpose(-1, -2)
        if pad_mask is not None:
            sx = sx.masked_fill(pad_mask[:, None, None, :], float("-inf"))
        maps = torch.softmax(sx.float(), dim=-1)
    ctx = (_rnd(probs, inject, "P") @ _rnd(v, inject, "V")).transpose(1, 2).reshape(B, T, E)
    out = _linear(ctx, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"], inject, "o")
    return out, (maps if need_weights else None)


class FoldRows:
    """Row state of the LayerNorm-fold form ("FOLD" inject kind): what the engine's residual epilogues hand to the next
    GEMM — the rounded raw rows x - m_prev, and fp32 (mean, rstd) from the UNROUNDED differences
    (esm_amd/csrc/elementwise.hip rowstats_kernel / ln_finalize_kernel)."""

    def __init__(self, x, dtype):
        self.dtype = dtype
        self.mean = x.mean(-1, keepdim=True)  # chain entry (layer 0): the row's own mean
        self.update(x)

    def update(self, x):
        d = x - self.mean                     # centred on the PREVIOUS mean
        dm = d.mean(-1, keepdim=True)
        var = ((d * d).mean(-1, keepdim=True) - dm * dm).clamp_min(0.0)
        self.mean = self.mean + dm
        self.rstd = torch.rsqrt(var + 1e-5)
        self.rows = d.to(self.dtype).float()

    def linear(self, gamma, beta, w, b):
        """LayerNorm(x; gamma, beta) . w^T + b in the fold's form."""
        wg = w * gamma[None, :]
        wg = (wg - wg.mean(-1, keepdim=True)).to(self.dtype).float()
        return self.rstd * F.linear(self.rows, wg) + (F.linear(beta[None, :], w)[0] + b)


def _is_fold(inject):
    return inject is not None and "FOLD" in inject[0]


def transformer_layer(sd, i, x, heads, pad_mask, need_weights, use_rope=True, inject=None, fold=None, last=False):
    # reference esm/modules.py:120-142 (pre-LN residual blocks)
    p = f"layers.{i}."
    if fold is not None:
        # the same layer with its two LayerNorm -> Linear pairs in the fold's form (FoldRows); everything else as below
        g, b = sd[p + "self_attn_layer_norm.weight"], sd[p + "self_attn_layer_norm.bias"]
        a, probs = attention_layer(sd, p, None, heads, pad_mask, need_weights, use_rope, inject,
                                   qkv=lambda n: fold.linear(g, b, sd[p + f"self_attn.{n}_proj.weight"], sd[p + f"self_attn.{n}_proj.bias"]),
                                   shape=x.shape)
        x = x + a
        fold.update(x)
        h = gelu(fold.linear(sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], sd[p + "fc1.weight"], sd[p + "fc1.bias"]))
        x = x + _linear(h, sd[p + "fc2.weight"], sd[p + "fc2.bias"], inject, "fc2")
        if not last:
            fold.update(x)
        return x, probs
    h = layer_norm(x, sd[p + "self_attn_layer_norm.weight"], sd[p + "self_attn_layer_norm.bias"])
    a, probs = attention_layer(sd, p, h, heads, pad_mask, need_weights, use_rope, inject)
    x = x + a
    h = layer_norm(x, sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"])
    h = gelu(_linear(h, sd[p + "fc1.weight"], sd[p + "fc1.bias"], inject, "fc1"))
    h = _linear(h, sd[p + "fc2.weight"], sd[p + "fc2.bias"], inject, "fc2")
    return x + h, probs


def contact_head(sd, tokens, attentions, eos_idx=2, prepend_bos=True, append_eos=True):
    # reference esm/modules.py:338-357 with symmetrize / apc (modules.py:27-41)
    if append_eos:
        keep = tokens.ne(eos_idx).to(attentions)
        attentions = attentions * (keep[:, :, None] * keep[:, None, :])[:, None, None]
        attentions = attentions[..., :-1, :-1]
    if prepend_bos:
        attentions = attentions[..., 1:, 1:]
    B, L, H, S, _ = attentions.shape
    a = attentions.reshape(B, L * H, S, S)
    a = a + a.transpose(-1, -2)
    a1 = a.sum(-1, keepdim=True)
    a2 = a.sum(-2, keepdim=True)
    a12 = a.sum((-1, -2), keepdim=True)
    a = a - (a1 * a2) / a12
    z = F.linear(a.permute(0, 2, 3, 1), sd["contact_head.regression.weight"], sd["contact_head.regression.bias"])
    return torch.sigmoid(z.squeeze(3))


@torch.no_grad()
def esm2_forward(
    sd, tokens, num_layers, heads, repr_layers=(), need_head_weights=False, return_contacts=False,
    token_dropout=True, padding_idx=1, mask_idx=32, eos_idx=2, prepend_bos=True, append_eos=True, inject=None,
    inject_head="same",
):
    """reference esm/model/esm2.py:77-144.  ``sd``: fp32 state dict with the reference's keys."""
    if return_contacts:
        need_head_weights = True
    assert tokens.ndim == 2
    pad = tokens.eq(padding_idx)  # esm2.py:82
    x = sd["embed_tokens.weight"][tokens]  # esm2.py:84 (embed_scale = 1)
    if token_dropout:  # esm2.py:86-92
        x = x.masked_fill((tokens == mask_idx).unsqueeze(-1), 0.0)
        mask_ratio_train = 0.15 * 0.8
        src_lengths = (~pad).sum(-1)
        ratio = (tokens == mask_idx).sum(-1).to(x.dtype) / src_lengths
        x = x * (1 - mask_ratio_train) / (1 - ratio)[:, None, None]
    x = x * (1 - pad.unsqueeze(-1).type_as(x))  # esm2.py:94-95
    wanted = set(int(i) for i in repr_layers)
    reps = {}
    if 0 in wanted:
        reps[0] = x
    pad_mask = pad if bool(pad.any()) else None  # esm2.py:108-109
    attn = []
    fold = FoldRows(x, inject[1]) if _is_fold(inject) else None
    for i in range(num_layers):  # esm2.py:111-121
        x, probs = transformer_layer(sd, i, x, heads, pad_mask, need_head_weights, inject=inject, fold=fold, last=i + 1 == num_layers)
        if (i + 1) in wanted:
            reps[i + 1] = x
        if need_head_weights:
            attn.append(probs)
    x = layer_norm(x, sd["emb_layer_norm_after.weight"], sd["emb_layer_norm_after.bias"])  # :123
    if num_layers in wanted:
        reps[num_layers] = x  # esm2.py:127-128
    # RobertaLMHead, reference esm/modules.py:308-314 (weight tied to the embedding, esm2.py:71-75)
    ih = inject if isinstance(inject_head, str) else inject_head  # the LM head may round differently from the stack
    h = gelu(_linear(x, sd["lm_head.dense.weight"], sd["lm_head.dense.bias"], ih))
    h = layer_norm(h, sd["lm_head.layer_norm.weight"], sd["lm_head.layer_norm.bias"])
    logits = _linear(h, sd["embed_tokens.weight"], None, ih) + sd["lm_head.bias"]
    out = {"logits": logits, "representations": reps}
    if need_head_weights:
        attentions = torch.stack(attn, 1)  # [B,L,H,T,T], esm2.py:132-139
        if pad_mask is not None:
            keep = 1 - pad_mask.type_as(attentions)
            attentions = attentions * (keep[:, None, :] * keep[:, :, None])[:, None, None]
        out["attentions"] = attentions
        if return_contacts:
            out["contacts"] = contact_head(sd, tokens, attentions, eos_idx, prepend_bos, append_eos)
    return out
