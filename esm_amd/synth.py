"""Deterministic synthetic ESM-2 parameters and checkpoints.

No pretrained weights exist offline (no network on either box), so parity tests and the benchmark
use seeded random parameters: every tensor is drawn from its own generator, seeded by the CRC32 of
its state-dict key XOR a global seed, independent of any module constructor.  The q/k projection
weights are scaled by ``qk_gain`` (default 2) so that softmax rows are clearly non-uniform —
with small random weights attention is ~1/T everywhere and RoPE / masking / head-layout bugs are
numerically muted — while the stack stays non-chaotic end to end.
"""
import argparse
import contextlib
import os
import zlib

import torch

ESM2_DIMS = {
    # name: (layers, embed_dim, heads)                                   
    "esm2_t6_8M_UR50D": (6, 320, 20),
    "esm2_t12_35M_UR50D": (12, 480, 20),
    "esm2_t30_150M_UR50D": (30, 640, 20),
    "esm2_t33_650M_UR50D": (33, 1280, 20),
    "esm2_t36_3B_UR50D": (36, 2560, 40),
    "esm2_t48_15B_UR50D": (48, 5120, 40),
}



@contextlib.contextmanager
def skip_param_init():
    """Construct a model WITHOUT the random fill that nn.Linear / nn.LayerNorm / nn.Embedding constructors give their
    parameters (28 s for the 3B architecture on 8 threads) when a strict ``load_state_dict`` of a complete synthetic
    state dict follows right away — which is what checks that nothing stays uninitialised.  Bench and tests only: it
    patches ``reset_parameters`` of the three classes PROCESS-WIDE for the duration of the block, so it is not thread
    safe and must not wrap the construction of modules that are not fully overwritten afterwards."""
    import torch.nn as nn

    classes = (nn.Linear, nn.LayerNorm, nn.Embedding)
    saved = [c.reset_parameters for c in classes]
    for c in classes:
        c.reset_parameters = lambda self: None
    try:
        yield
    finally:
        for c, f in zip(classes, saved):
            c.reset_parameters = f

def esm2_param_shapes(num_layers, embed_dim, heads, vocab=33):
    """State-dict keys and shapes of ESM2 (reference esm/model/esm2.py:40-75)."""
    E, F = embed_dim, 4 * embed_dim
    shapes = {"embed_tokens.weight": (vocab, E)}
    for i in range(num_layers):
        p = f"layers.{i}."
        for proj in ("k_proj", "v_proj", "q_proj", "out_proj"):
            shapes[p + f"self_attn.{proj}.weight"] = (E, E)
            shapes[p + f"self_attn.{proj}.bias"] = (E,)
        shapes[p + "self_attn.rot_emb.inv_freq"] = (E // heads // 2,)
        shapes[p + "self_attn_layer_norm.weight"] = (E,)
        shapes[p + "self_attn_layer_norm.bias"] = (E,)
        shapes[p + "fc1.weight"] = (F, E)
        shapes[p + "fc1.bias"] = (F,)
        shapes[p + "fc2.weight"] = (E, F)
        shapes[p + "fc2.bias"] = (E,)
        shapes[p + "final_layer_norm.weight"] = (E,)
        shapes[p + "final_layer_norm.bias"] = (E,)
    shapes["contact_head.regression.weight"] = (1, num_layers * heads)
    shapes["contact_head.regression.bias"] = (1,)
    shapes["emb_layer_norm_after.weight"] = (E,)
    shapes["emb_layer_norm_after.bias"] = (E,)
    shapes["lm_head.weight"] = (vocab, E)
    shapes["lm_head.bias"] = (vocab,)
    shapes["lm_head.dense.weight"] = (E, E)
    shapes["lm_head.dense.bias"] = (E,)
    shapes["lm_head.layer_norm.weight"] = (E,)
    shapes["lm_head.layer_norm.bias"] = (E,)
    return shapes


def _draw(key, shape, seed, std, mean=0.0, device="cpu"):
    g = torch.Generator(device=device)
    g.manual_seed((zlib.crc32(key.encode()) ^ seed) & 0x7FFFFFFF)
    return torch.randn(shape, generator=g, device=device, dtype=torch.float32) * std + mean


def synth_esm2_state_dict(num_layers, embed_dim, heads, seed=0, qk_gain=2.0, device="cpu", vocab=33, ln_gamma_std=0.02):
    """fp32 state dict with the reference's key names (tied lm_head.weight included).  ``qk_gain`` / ``ln_gamma_std``: the
    two knobs of the data-sensitivity line of bench.py (sharper attention maps, wider spread of the LayerNorm gains)."""
    sd = {}
    d = embed_dim // heads
    for key, shape in esm2_param_shapes(num_layers, embed_dim, heads, vocab).items():
        if key.endswith("inv_freq"):
            sd[key] = (1.0 / (10000 ** (torch.arange(0, d, 2).float() / d))).to(device)
        elif key == "lm_head.weight":
            continue
        elif key == "embed_tokens.weight":
            sd[key] = _draw(key, shape, seed, 0.25, device=device)
        elif key.startswith("contact_head.regression.weight"):
            sd[key] = _draw(key, shape, seed, 4.0, device=device)
        elif key.startswith("contact_head.regression.bias"):
            sd[key] = _draw(key, shape, seed, 0.5, mean=-1.0, device=device)
        elif "layer_norm" in key and key.endswith(".weight"):
            sd[key] = _draw(key, shape, seed, ln_gamma_std, mean=1.0, device=device)
        elif len(shape) == 2:
            # std 0.02 at fan-in 1280, scaled with fan-in^-1/2 so that activation statistics (and
            # the sharpness of the attention) do not depend on the model width
            gain = qk_gain if (".q_proj." in key or ".k_proj." in key) else 1.0
            sd[key] = _draw(key, shape, seed, 0.02 * (1280.0 / shape[1]) ** 0.5 * gain, device=device)
        else:
            sd[key] = _draw(key, shape, seed, 0.02, device=device)
    sd["lm_head.weight"] = sd["embed_tokens.weight"]
    return sd


def add_outlier_channels(sd, num_layers, embed_dim, channels=4, magnitude=200.0, token_spread=10.0, seed=0, balanced=False):
    """The "stress" weight set of SURVEY.md §7.4: large-magnitude outlier channels in the residual stream, to approximate the
    dynamic range a trained checkpoint may have (a few channels hundreds of times the rest, LayerNorm gains that undo them).
    In place on an ESM-2 state dict of ``synth_esm2_state_dict``; returns the channel indices.

    * layer 0's ``fc2`` writes ``+-magnitude`` into ``channels`` seeded channels through its bias (the same for every token)
      and ``token_spread`` times its usual weight rows (different for every token); every later layer's out-projection and
      ``fc2`` keep writing into those channels at ``token_spread`` times the usual scale, so the outliers move with depth;
    * the outliers add ``channels * magnitude^2 / E`` to a row's variance.  Every LayerNorm that sees them gets its gains
      multiplied by sqrt(1 + that / s^2) on the ordinary channels — s(p)^2 = 0.22 p^1.1 is the variance of the ordinary
      stream of these synthetic weights at depth p (measured: 0.468, 0.963, 1.43, 2.15, 3.13 at p = 1, 4, 8, 16, 32 for every
      width) — so that the normalised ordinary channels keep the scale they have without the outliers (the network stays
      as sharp and as non-chaotic as the plain synthetic one), and set so that the normalised outliers come out near 1: a
      gain spread of about magnitude / s : 1 inside one LayerNorm.  Of every four outliers three point up and one down
      (``balanced``: two and two), so the row mean moves by sum(sign) * magnitude / E; the LayerNorm biases of the
      ordinary channels take that shift back.  Whether trained checkpoints look like this cannot be checked offline: the
      set is a stress case for the engine's operand forms, not a model of one."""
    g = torch.Generator()
    g.manual_seed(0x5EED ^ seed)
    idx = torch.randperm(embed_dim, generator=g)[:channels].sort().values
    # three up, one down: the row mean moves too (by magnitude / 640 at E = 1280) and the LayerNorm biases take that back;
    # balanced: two up, two down — the same outliers without a common offset of the ordinary channels
    sign = torch.where(torch.arange(channels) % 2 == 1, -1.0, 1.0) if balanced else torch.where(torch.arange(channels) % 4 == 3, -1.0, 1.0)
    dev = sd["layers.0.fc2.bias"].device
    idx_d = idx.to(dev)
    sd["layers.0.fc2.bias"][idx_d] = (sign * magnitude).to(dev)
    for i in range(num_layers):
        sd[f"layers.{i}.fc2.weight"][idx_d] *= token_spread
        if i > 0:
            sd[f"layers.{i}.self_attn.out_proj.weight"][idx_d] *= token_spread
    out_var = channels * magnitude ** 2 / embed_dim

    mean_shift = float(sign.sum()) * magnitude / embed_dim  # what the constant part of the outliers adds to a row's mean

    def regain(name, depth):
        s2 = 0.22 * depth ** 1.1
        w, b = sd[name + ".weight"], sd[name + ".bias"]
        row_std = (out_var - mean_shift ** 2 + s2) ** 0.5
        w *= row_std / s2 ** 0.5
        w[idx_d] = w[idx_d] * (s2 ** 0.5 / magnitude)
        keep = b[idx_d].clone()
        b += w * (mean_shift / row_std)  # the ordinary channels' (x - mean) / std * gain is lower by exactly this
        b[idx_d] = keep

    for i in range(1, num_layers):  # layer 0's two LayerNorms come before the first outlier is written
        regain(f"layers.{i}.self_attn_layer_norm", i)
        regain(f"layers.{i}.final_layer_norm", i + 0.5)
    regain("emb_layer_norm_after", num_layers)
    return idx


def synth_tokens(batch, length, seed=1, device="cpu"):
    """BASELINE synthetic batch: <cls> + `length` uniform ids in 4..23 + <eos> (SURVEY §8 d)."""
    g = torch.Generator()
    g.manual_seed(seed)
    toks = torch.randint(4, 24, (batch, length + 2), generator=g, dtype=torch.int64)
    toks[:, 0] = 0
    toks[:, -1] = 2
    return toks.to(device)


def write_esm2_checkpoint(directory, name, num_layers, embed_dim, heads, seed=0, qk_gain=2.0, token_dropout=True):
    """Write ``<name>.pt`` + ``<name>-contact-regression.pt`` in the reference's ESM-2 checkpoint
    format (SURVEY.md Appendix A / reference esm/pretrained.py:67-77,164-188)."""
    os.makedirs(directory, exist_ok=True)
    sd = synth_esm2_state_dict(num_layers, embed_dim, heads, seed, qk_gain)
    regression = {k: v for k, v in sd.items() if k.startswith("contact_head.")}
    body = {"encoder.sentence_encoder." + k: v for k, v in sd.items() if not k.startswith("contact_head.")}
    cfg = argparse.Namespace(
        encoder_layers=num_layers, encoder_embed_dim=embed_dim, encoder_attention_heads=heads,
        token_dropout=token_dropout,
    )
    path = os.path.join(directory, name + ".pt")
    torch.save({"cfg": {"model": cfg}, "model": body}, path)
    torch.save({"model": regression}, os.path.join(directory, name + "-contact-regression.pt"))
    return path


# ---------------------------------------------------------------------------------------------------
# MSA Transformer (reference esm/model/msa_transformer.py:88-144)
# ---------------------------------------------------------------------------------------------------
MSA_DIMS = {
    # name: (layers, embed_dim, heads, ffn)                               
    "esm_msa1b_t12_100M_UR50S": (12, 768, 12, 3072),
}


def msa_param_shapes(num_layers, embed_dim, heads, ffn_dim, max_positions=1024, vocab=33, padding_idx=1):
    """State-dict keys and shapes of MSATransformer with embed_positions_msa=True."""
    E, F = embed_dim, ffn_dim
    shapes = {"msa_position_embedding": (1, 1024, 1, E), "embed_tokens.weight": (vocab, E)}
    for i in range(num_layers):
        for blk in ("row_self_attention", "column_self_attention"):
            p = f"layers.{i}.{blk}."
            for proj in ("k_proj", "v_proj", "q_proj", "out_proj"):
                shapes[p + f"layer.{proj}.weight"] = (E, E)
                shapes[p + f"layer.{proj}.bias"] = (E,)
            shapes[p + "layer_norm.weight"] = (E,)
            shapes[p + "layer_norm.bias"] = (E,)
        p = f"layers.{i}.feed_forward_layer."
        shapes[p + "layer.fc1.weight"] = (F, E)
        shapes[p + "layer.fc1.bias"] = (F,)
        shapes[p + "layer.fc2.weight"] = (E, F)
        shapes[p + "layer.fc2.bias"] = (E,)
        shapes[p + "layer_norm.weight"] = (E,)
        shapes[p + "layer_norm.bias"] = (E,)
    shapes["contact_head.regression.weight"] = (1, num_layers * heads)
    shapes["contact_head.regression.bias"] = (1,)
    shapes["embed_positions.weight"] = (max_positions + padding_idx + 1, E)
    for k in ("emb_layer_norm_before", "emb_layer_norm_after", "lm_head.layer_norm"):
        shapes[k + ".weight"] = (E,)
        shapes[k + ".bias"] = (E,)
    shapes["lm_head.weight"] = (vocab, E)
    shapes["lm_head.bias"] = (vocab,)
    shapes["lm_head.dense.weight"] = (E, E)
    shapes["lm_head.dense.bias"] = (E,)
    return shapes


def synth_msa_state_dict(num_layers, embed_dim, heads, ffn_dim, seed=0, qk_gain=2.0, device="cpu", max_positions=1024):
    """fp32 MSA-Transformer state dict with the reference's key names."""
    sd = {}
    for key, shape in msa_param_shapes(num_layers, embed_dim, heads, ffn_dim, max_positions).items():
        if key == "lm_head.weight":
            continue
        if key == "embed_tokens.weight":
            sd[key] = _draw(key, shape, seed, 0.25, device=device)
        elif key == "embed_positions.weight":
            sd[key] = _draw(key, shape, seed, 0.1, device=device)
            sd[key][1] = 0.0  # padding_idx row of nn.Embedding
        elif key == "msa_position_embedding":
            sd[key] = _draw(key, shape, seed, 0.1, device=device)
        elif key.startswith("contact_head.regression.weight"):
            sd[key] = _draw(key, shape, seed, 4.0, device=device)
        elif key.startswith("contact_head.regression.bias"):
            sd[key] = _draw(key, shape, seed, 0.5, mean=-1.0, device=device)
        elif "layer_norm" in key and key.endswith(".weight"):
            sd[key] = _draw(key, shape, seed, 0.02, mean=1.0, device=device)
        elif len(shape) == 2:
            gain = qk_gain if (".q_proj." in key or ".k_proj." in key) else 1.0
            sd[key] = _draw(key, shape, seed, 0.02 * (1280.0 / shape[1]) ** 0.5 * gain, device=device)
        else:
            sd[key] = _draw(key, shape, seed, 0.02, device=device)
    sd["lm_head.weight"] = sd["embed_tokens.weight"]
    return sd


def synth_msa_tokens(batch, rows, cols, seed=1, gap_frac=0.05, device="cpu"):
    """BASELINE config 5 input: column 0 = <cls>, the rest uniform over the 20 amino acids with
    ``gap_frac`` '-' (id 30) tokens, no pads (SURVEY.md §8 d)."""
    g = torch.Generator()
    g.manual_seed(seed)
    toks = torch.randint(4, 24, (batch, rows, cols), generator=g, dtype=torch.int64)
    gaps = torch.rand((batch, rows, cols), generator=g) < gap_frac
    toks[gaps] = 30
    toks[:, :, 0] = 0
    return toks.to(device)


# ---------------------------------------------------------------------------------------------------
# ESM-1b / ESM-1v (reference esm/model/esm1.py, arch "roberta_large")
# ---------------------------------------------------------------------------------------------------
def synth_esm1b_state_dict(num_layers, embed_dim, heads, ffn_dim=None, seed=0, qk_gain=2.0, max_positions=1024,
                           ln_before=True):
    """fp32 ESM-1b state dict with the reference's key names: the ESM-2 keys without rot_emb.inv_freq, plus
    embed_positions.weight and (optionally) emb_layer_norm_before.*."""
    ffn_dim = ffn_dim or 4 * embed_dim
    sd = {k: v for k, v in synth_esm2_state_dict(num_layers, embed_dim, heads, seed=seed, qk_gain=qk_gain).items()
          if not k.endswith("inv_freq")}
    if ffn_dim != 4 * embed_dim:
        for i in range(num_layers):
            for key, shape in ((f"layers.{i}.fc1.weight", (ffn_dim, embed_dim)), (f"layers.{i}.fc1.bias", (ffn_dim,)),
                               (f"layers.{i}.fc2.weight", (embed_dim, ffn_dim))):
                std = 0.02 * (1280.0 / shape[1]) ** 0.5 if len(shape) == 2 else 0.02
                sd[key] = _draw(key, shape, seed, std)
    sd["embed_positions.weight"] = _draw("embed_positions.weight", (max_positions + 2, embed_dim), seed, 0.1)
    sd["embed_positions.weight"][1] = 0.0
    if ln_before:
        sd["emb_layer_norm_before.weight"] = _draw("emb_layer_norm_before.weight", (embed_dim,), seed, 0.02, mean=1.0)
        sd["emb_layer_norm_before.bias"] = _draw("emb_layer_norm_before.bias", (embed_dim,), seed, 0.02)
    sd["embed_tokens.weight"] = sd["embed_tokens.weight"].clone()
    sd["embed_tokens.weight"][32] = 0.0  # pretrained.py:97 zeroes the <mask> row for token dropout
    sd["lm_head.weight"] = sd["embed_tokens.weight"]
    return sd
