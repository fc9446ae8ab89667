"""ORACLE — test infrastructure only.  NOT part of the product.

A plain-PyTorch fp32 CPU restatement of the reference's MSA Transformer forward pass (axial row /
column attention, SURVEY.md §8 a-15 ... a-18), every step citing the reference code it follows.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` may import it, as the
checker.  Pinned against fixtures produced by the reference itself (tests/golden/make_golden.py ->
tests/golden/msa_*.pt; tests/test_oracle.py checks <= 2e-5).

Works on a state dict with the reference's key names; eval mode (all dropouts are identity).
The reference's ``max_tokens_per_msa`` chunking (axial_attention.py:40-73,160-183) is a memory
optimisation that sums / concatenates the same terms, so it is not restated.
"""
import math

import torch
import torch.nn.functional as F

from .esm2_oracle import contact_head, gelu, layer_norm


def row_attention(sd, p, x, heads, pad_mask):
    """RowSelfAttention (reference esm/axial_attention.py:75-130): x [R,C,B,D]; one attention map per
    head shared by all rows (tied), scaled by d^-1/2 / sqrt(R) (:36-38)."""
    R, C, B, D = x.shape
    d = D // heads
    scaling = (d ** -0.5) / math.sqrt(R)
    q = F.linear(x, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"]).view(R, C, B, heads, d) * scaling  # :82-84
    k = F.linear(x, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"]).view(R, C, B, heads, d)
    v = F.linear(x, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"]).view(R, C, B, heads, d)
    if pad_mask is not None:  # :85-88 q zeroed at padded positions (pad_mask [B,R,C])
        q = q * (1 - pad_mask.permute(1, 2, 0).unsqueeze(3).unsqueeze(4).to(q))
    w = torch.einsum("rinhd,rjnhd->hnij", q, k)  # :90
    if pad_mask is not None:  # :96-100 columns padded in row 0
        w = w.masked_fill(pad_mask[:, 0].unsqueeze(0).unsqueeze(2), -10000)
    probs = w.softmax(-1)  # :127
    ctx = torch.einsum("hnij,rjnhd->rinhd", probs, v).contiguous().view(R, C, B, D)  # :111-112
    return F.linear(ctx, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"]), probs  # :113


def column_attention(sd, p, x, heads, pad_mask):
    """ColumnSelfAttention (reference esm/axial_attention.py:185-239)."""
    R, C, B, D = x.shape
    d = D // heads
    if R == 1:  # :189-200
        probs = torch.ones(heads, C, B, R, R, dtype=x.dtype)
        v = F.linear(x, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"])
        return F.linear(v, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"]), probs
    q = F.linear(x, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"]).view(R, C, B, heads, d) * (d ** -0.5)
    k = F.linear(x, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"]).view(R, C, B, heads, d)
    v = F.linear(x, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"]).view(R, C, B, heads, d)
    w = torch.einsum("icnhd,jcnhd->hcnij", q, k)  # :207
    if pad_mask is not None:  # :211-215
        w = w.masked_fill(pad_mask.permute(2, 0, 1).unsqueeze(0).unsqueeze(3), -10000)
    probs = w.softmax(-1)
    ctx = torch.einsum("hcnij,jcnhd->icnhd", probs, v).contiguous().view(R, C, B, D)
    return F.linear(ctx, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"]), probs


def axial_layer(sd, i, x, heads, pad_mask):
    """AxialTransformerLayer + NormalizedResidualBlock (reference esm/modules.py:196-221,360-392):
    x += f(LN(x)) for f = row attention, column attention, FFN (modules.py:395-418)."""
    p = f"layers.{i}.row_self_attention."
    a, row_probs = row_attention(sd, p + "layer.", layer_norm(x, sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"]),
                                 heads, pad_mask)
    x = x + a
    p = f"layers.{i}.column_self_attention."
    a, col_probs = column_attention(sd, p + "layer.", layer_norm(x, sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"]),
                                    heads, pad_mask)
    x = x + a
    p = f"layers.{i}.feed_forward_layer."
    h = layer_norm(x, sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"])
    h = F.linear(gelu(F.linear(h, sd[p + "layer.fc1.weight"], sd[p + "layer.fc1.bias"])),
                 sd[p + "layer.fc2.weight"], sd[p + "layer.fc2.bias"])
    return x + h, col_probs, row_probs


@torch.no_grad()
def msa_forward(sd, tokens, num_layers, heads, repr_layers=(), need_head_weights=False, return_contacts=False,
                padding_idx=1, eos_idx=2, prepend_bos=True, append_eos=False):
    """reference esm/model/msa_transformer.py:146-220; tokens [B,R,C]."""
    if return_contacts:
        need_head_weights = True
    assert tokens.ndim == 3
    B, R, C = tokens.shape
    pad = tokens.eq(padding_idx)
    pad_mask = pad if bool(pad.any()) else None  # :153-155
    x = sd["embed_tokens.weight"][tokens]  # :157
    # LearnedPositionalEmbedding (modules.py:240-257): positions count non-pad tokens, offset by pad idx
    flat = tokens.view(B * R, C)
    m = flat.ne(padding_idx).int()
    positions = (torch.cumsum(m, dim=1).type_as(m) * m).long() + padding_idx
    x = x + sd["embed_positions.weight"][positions].view(x.shape)  # :158
    if "msa_position_embedding" in sd:  # :159-165
        if R > 1024:
            raise RuntimeError("MSA depth above 1024")
        x = x + sd["msa_position_embedding"][:, :R]
    x = layer_norm(x, sd["emb_layer_norm_before.weight"], sd["emb_layer_norm_before.bias"])  # :167
    if pad_mask is not None:
        x = x * (1 - pad_mask.unsqueeze(-1).type_as(x))  # :171-172
    wanted = set(int(i) for i in repr_layers)
    reps = {}
    if 0 in wanted:
        reps[0] = x
    rows, cols = [], []
    x = x.permute(1, 2, 0, 3)  # B R C D -> R C B D (:183)
    for i in range(num_layers):
        x, col_probs, row_probs = axial_layer(sd, i, x, heads, pad_mask)
        if need_head_weights:
            cols.append(col_probs.permute(2, 0, 1, 3, 4))  # H C B R R -> B H C R R (:193-194)
            rows.append(row_probs.permute(1, 0, 2, 3))     # H B C C -> B H C C (:195-196)
        if (i + 1) in wanted:
            reps[i + 1] = x.permute(2, 0, 1, 3)
    x = layer_norm(x, sd["emb_layer_norm_after.weight"], sd["emb_layer_norm_after.bias"])  # :200
    x = x.permute(2, 0, 1, 3)
    if num_layers in wanted:
        reps[num_layers] = x  # :204-205
    h = gelu(F.linear(x, sd["lm_head.dense.weight"], sd["lm_head.dense.bias"]))  # modules.py:308-314
    h = layer_norm(h, sd["lm_head.layer_norm.weight"], sd["lm_head.layer_norm.bias"])
    logits = F.linear(h, sd["embed_tokens.weight"]) + sd["lm_head.bias"]
    out = {"logits": logits, "representations": reps}
    if need_head_weights:
        out["col_attentions"] = torch.stack(cols, 1)  # B L H C R R
        out["row_attentions"] = torch.stack(rows, 1)  # B L H C C
        if return_contacts:
            out["contacts"] = contact_head(sd, tokens, out["row_attentions"], eos_idx, prepend_bos, append_eos)
    return out


# ---- the 16-bit-operand FLOOR of this model (as esm2_oracle's `inject`): the same forward with rounding injected at the
# engine's rounding points — linear-layer weights and inputs, the q / k / v operands of the attention contractions, the
# softmax probabilities.  tools/msa_precision_study.py (per-group study) and bench.py (floor next to the engine's
# parity numbers on the same sample) use it; nothing in the product does.
import torch as _torch  # the real modules: `F` / `torch` of this module are swapped for proxies during a rounded run
import torch.nn.functional as _F


class OperandRounding:
    def __init__(self, weights=None, acts=None, qkv=None, probs=None):
        self.weights, self.acts, self.qkv, self.probs = weights, acts, qkv, probs

    @staticmethod
    def _r(t, dt):
        return t.to(dt).float() if dt is not None else t

    def linear(self, x, w, b=None):
        return _F.linear(self._r(x, self.acts), self._r(w, self.weights), b)

    def einsum(self, eq, a, b):
        if eq.startswith("hnij") or eq.startswith("hcnij"):  # probs x v
            return _torch.einsum(eq, self._r(a, self.probs), self._r(b, self.qkv))
        return _torch.einsum(eq, self._r(a, self.qkv), self._r(b, self.qkv))


class _Proxy:
    def __init__(self, real, name, fn):
        self._real, self._name, self._fn = real, name, fn

    def __getattr__(self, k):
        return self._fn if k == self._name else getattr(self._real, k)


def msa_forward_rounded(rounding, sd, tokens, num_layers, heads, **kw):
    """msa_forward with `rounding` (an OperandRounding) applied at every F.linear / torch.einsum of this module."""
    g = globals()
    real_F, real_torch = g["F"], g["torch"]
    g["F"], g["torch"] = _Proxy(real_F, "linear", rounding.linear), _Proxy(real_torch, "einsum", rounding.einsum)
    try:
        return msa_forward(sd, tokens, num_layers, heads, **kw)
    finally:
        g["F"], g["torch"] = real_F, real_torch


def msa_operand_floor(sd, tokens, num_layers, heads, dtype=None, **kw):
    """The floor: every MFMA operand of the engine rounded to `dtype` (default fp16)."""
    dt = dtype or _torch.float16
    return msa_forward_rounded(OperandRounding(weights=dt, acts=dt, qkv=dt, probs=dt), sd, tokens, num_layers, heads, **kw)
