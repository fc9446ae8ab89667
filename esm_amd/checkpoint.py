"""Model loading with the entry points of the reference's ``esm.pretrained`` (reference
esm/pretrained.py:18-221 and the zero-argument factories at :224-552), written from the format
description in SURVEY.md Appendix A.

Checkpoint formats understood
  * ESM-2: ``{"cfg": {"model": ns(encoder_layers, encoder_embed_dim, encoder_attention_heads,
    token_dropout)}, "model": state}`` with the key prefixes ``encoder.sentence_encoder.`` /
    ``encoder.`` removed; dispatched by the file stem starting with ``esm2``;
  * optional sibling ``<stem>-contact-regression.pt`` holding ``contact_head.regression.*``.
  * ``{"args": Namespace(arch=...), "model": state}``: ``msa_transformer`` (MSA Transformer) and
    ``roberta_large`` (ESM-1b / ESM-1v), with the reference's key-prefix upgrades.
Other model families of the reference (ESM-1 ``protein_bert_base``, ESM-IF1, ESMFold) are outside the MI355X
engine's scope and raise ``NotImplementedError`` when a checkpoint asks for them.
"""
import re
import urllib
import warnings
from pathlib import Path

import torch

from .alphabet import Alphabet
from .esm2 import ESM2

_HUB = "https://dl.fbaipublicfiles.com/fair-esm"
_NO_REGRESSION_MARKERS = ("esm1v", "esm_if", "270K", "500K")


def _has_regression_weights(model_name):
    """All released models ship contact-regression weights except ESM-1v, ESM-IF1 and the
    partially trained ESM-2 checkpoints (270K / 500K updates)."""
    return not any(m in model_name for m in _NO_REGRESSION_MARKERS)


def load_model_and_alphabet(model_name):
    if model_name.endswith(".pt"):
        return load_model_and_alphabet_local(model_name)
    return load_model_and_alphabet_hub(model_name)


def _torch_load(path):
    # checkpoints carry argparse.Namespace objects -> need the full unpickler on torch >= 2.6
    try:
        return torch.load(str(path), map_location="cpu", weights_only=False)
    except TypeError:  # very old torch without the keyword
        return torch.load(str(path), map_location="cpu")


def load_hub_workaround(url):
    try:
        return torch.hub.load_state_dict_from_url(url, progress=False, map_location="cpu")
    except RuntimeError:
        return _torch_load(f"{torch.hub.get_dir()}/checkpoints/{Path(url).name}")
    except urllib.error.HTTPError:
        raise Exception(f"Could not load {url}, check if you specified a correct model name?")


def load_regression_hub(model_name):
    return load_hub_workaround(f"{_HUB}/regression/{model_name}-contact-regression.pt")


def _download_model_and_regression_data(model_name):
    model_data = load_hub_workaround(f"{_HUB}/models/{model_name}.pt")
    regression = load_regression_hub(model_name) if _has_regression_weights(model_name) else None
    return model_data, regression


def load_model_and_alphabet_hub(model_name):
    model_data, regression = _download_model_and_regression_data(model_name)
    return load_model_and_alphabet_core(model_name, model_data, regression)


def load_model_and_alphabet_local(model_location):
    """Load a ``.pt`` checkpoint; the contact-regression file must sit next to it."""
    path = Path(model_location)
    model_data = _torch_load(path)
    name = path.stem
    regression = None
    if _has_regression_weights(name):
        regression = _torch_load(str(path.with_suffix("")) + "-contact-regression.pt")
    return load_model_and_alphabet_core(name, model_data, regression)


def has_emb_layer_norm_before(model_state):
    return any(k.startswith("emb_layer_norm_before") for k in model_state)


_PREFIX = re.compile(r"^(encoder\.sentence_encoder\.|encoder\.|sentence_encoder\.)")
_ARG_PREFIX = re.compile(r"^encoder_")


def strip_key_prefix(key):
    """'encoder.sentence_encoder.layers.0.fc1.weight' / 'encoder.lm_head.bias' -> module-relative key."""
    return _PREFIX.sub("", key)


def strip_arg_prefix(name):
    """'encoder_embed_dim' -> 'embed_dim' (hyper-parameter names of the v1 checkpoints)."""
    return _ARG_PREFIX.sub("", name)


def _build_esm2(model_data):
    cfg = model_data["cfg"]["model"]
    state = {strip_key_prefix(k): v for k, v in model_data["model"].items()}
    alphabet = Alphabet.from_architecture("ESM-1b")
    model = ESM2(
        num_layers=cfg.encoder_layers,
        embed_dim=cfg.encoder_embed_dim,
        attention_heads=cfg.encoder_attention_heads,
        alphabet=alphabet,
        token_dropout=cfg.token_dropout,
    )
    return model, alphabet, state


def _build_v1(model_data):
    arch = getattr(model_data.get("args"), "arch", None)
    if arch == "msa_transformer":
        from .msa_transformer import build_from_checkpoint

        return build_from_checkpoint(model_data)
    if arch == "roberta_large":  # ESM-1b / ESM-1v
        from .esm1 import build_from_checkpoint

        return build_from_checkpoint(model_data)
    raise NotImplementedError(
        f"architecture {arch!r} is outside the scope of the MI355X ESM-2 engine "
        "(ESM-1 / ESM-1b / ESM-1v / ESM-IF1 are not implemented)"
    )


def load_model_and_alphabet_core(model_name, model_data, regression_data=None):
    if regression_data is not None:
        model_data["model"].update(regression_data["model"])
    if model_name.startswith("esm2"):
        model, alphabet, state = _build_esm2(model_data)
    else:
        model, alphabet, state = _build_v1(model_data)

    expected, found = set(model.state_dict().keys()), set(state.keys())
    if regression_data is None:
        optional = {"contact_head.regression.weight", "contact_head.regression.bias"}
        problems = []
        missing = (expected - found) - optional
        if missing:
            problems.append(f"Missing key(s) in state_dict: {missing}.")
        unexpected = found - expected
        if unexpected:
            problems.append(f"Unexpected key(s) in state_dict: {unexpected}.")
        if problems:
            raise RuntimeError(
                "Error(s) in loading state_dict for {}:\n\t{}".format(type(model).__name__, "\n\t".join(problems))
            )
        if optional - found:
            warnings.warn("Regression weights not found, predicting contacts will not produce correct results.")
    model.load_state_dict(state, strict=regression_data is not None)
    return model, alphabet


def _factory(name, doc):
    def load():
        return load_model_and_alphabet_hub(name)

    load.__name__ = name
    load.__qualname__ = name
    load.__doc__ = doc
    return load


_RELEASED = {
    "esm2_t6_8M_UR50D": "6 layer ESM-2 model with 8M params, trained on UniRef50.",
    "esm2_t12_35M_UR50D": "12 layer ESM-2 model with 35M params, trained on UniRef50.",
    "esm2_t30_150M_UR50D": "30 layer ESM-2 model with 150M params, trained on UniRef50.",
    "esm2_t33_650M_UR50D": "33 layer ESM-2 model with 650M params, trained on UniRef50.",
    "esm2_t36_3B_UR50D": "36 layer ESM-2 model with 3B params, trained on UniRef50.",
    "esm2_t48_15B_UR50D": "48 layer ESM-2 model with 15B params, trained on UniRef50.",
    "esm1b_t33_650M_UR50S": "33 layer ESM-1b model with 650M params, trained on UniRef50.",
    "esm1v_t33_650M_UR90S_1": "33 layer ESM-1v model with 650M params, trained on UniRef90 (ensemble member 1).",
    "esm1v_t33_650M_UR90S_2": "33 layer ESM-1v model with 650M params, trained on UniRef90 (ensemble member 2).",
    "esm1v_t33_650M_UR90S_3": "33 layer ESM-1v model with 650M params, trained on UniRef90 (ensemble member 3).",
    "esm1v_t33_650M_UR90S_4": "33 layer ESM-1v model with 650M params, trained on UniRef90 (ensemble member 4).",
    "esm1v_t33_650M_UR90S_5": "33 layer ESM-1v model with 650M params, trained on UniRef90 (ensemble member 5).",
    "esm_msa1_t12_100M_UR50S": "MSA Transformer (ESM-MSA-1), 12 layers, 100M params.",
    "esm_msa1b_t12_100M_UR50S": "MSA Transformer (ESM-MSA-1b), 12 layers, 100M params.",
}
for _name, _doc in _RELEASED.items():
    globals()[_name] = _factory(_name, _doc)
del _name, _doc
