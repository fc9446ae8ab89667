"""ESM-1b / ESM-1v (``ProteinBertModel`` with ``args.arch == "roberta_large"``) on the MI355X engine.

Same public surface as the reference class (reference esm/model/esm1.py:22-200): ``__init__(args, alphabet)``,
state-dict key names, ``forward(tokens, repr_layers, need_head_weights, return_contacts)``,
``predict_contacts``, ``num_layers``.  The layer stack is the ESM-2 one without rotary embeddings
(``TransformerLayer(use_rotary_embeddings=False)``, esm1.py:71-82) plus a learned positional embedding and
``emb_layer_norm_before`` (esm1.py:88-104,133-139); ``forward`` is inherited from ``esm_amd.ESM2`` and runs in
``esmk_forward`` with the ``no_rope / num_positions / ln_before`` fields of ``esmk_config`` set.

The original ESM-1 models (arch ``protein_bert_base``: sinusoidal positions, bias_kv, untied output
embedding) are not implemented.
"""
import torch
import torch.nn as nn

from .esm2 import ESM2, ContactPredictionHead, RobertaLMHead, TransformerLayer
from .msa_transformer import LearnedPositionalEmbedding


class ProteinBertModel(ESM2):
    @classmethod
    def add_args(cls, parser):
        # reference esm/model/esm1.py:23-46
        parser.add_argument("--num_layers", default=36, type=int, metavar="N", help="number of layers")
        parser.add_argument("--embed_dim", default=1280, type=int, metavar="N", help="embedding dimension")
        parser.add_argument("--logit_bias", action="store_true", help="whether to apply bias to logits")
        parser.add_argument("--ffn_embed_dim", default=5120, type=int, metavar="N", help="embedding dimension for FFN")
        parser.add_argument("--attention_heads", default=20, type=int, metavar="N", help="number of attention heads")

    def __init__(self, args, alphabet):
        if getattr(args, "arch", None) != "roberta_large":
            raise NotImplementedError(
                "only the ESM-1b / ESM-1v architecture (arch 'roberta_large') runs on the MI355X engine; "
                f"arch {getattr(args, 'arch', None)!r} (ESM-1: sinusoidal positions, bias_kv) is not implemented")
        nn.Module.__init__(self)
        self.args = args
        self.model_version = "ESM-1b"
        self.num_layers_ = args.layers
        self.embed_dim = args.embed_dim
        self.ffn_embed_dim = args.ffn_embed_dim
        self.attention_heads = args.attention_heads
        self.alphabet = alphabet
        self.alphabet_size = len(alphabet)
        self.padding_idx = alphabet.padding_idx
        self.mask_idx = alphabet.mask_idx
        self.cls_idx = alphabet.cls_idx
        self.eos_idx = alphabet.eos_idx
        self.prepend_bos = alphabet.prepend_bos
        self.append_eos = alphabet.append_eos
        self.token_dropout = bool(getattr(args, "token_dropout", False))
        ln_before = bool(getattr(args, "emb_layer_norm_before", False))
        E = self.embed_dim
        self.embed_scale = 1
        self.embed_tokens = nn.Embedding(self.alphabet_size, E, padding_idx=self.padding_idx)
        self.layers = nn.ModuleList([TransformerLayer(E, self.ffn_embed_dim, self.attention_heads)
                                     for _ in range(args.layers)])
        for layer in self.layers:  # no rotary embedding in ESM-1b: drop the inv_freq buffer from the state dict
            del layer.self_attn.rot_emb
        self.contact_head = ContactPredictionHead(args.layers * self.attention_heads, self.prepend_bos,
                                                  self.append_eos, eos_idx=self.eos_idx)
        self.embed_positions = LearnedPositionalEmbedding(args.max_positions, E, self.padding_idx)
        self.emb_layer_norm_before = nn.LayerNorm(E) if ln_before else None
        self.emb_layer_norm_after = nn.LayerNorm(E)
        self.lm_head = RobertaLMHead(E, self.alphabet_size, self.embed_tokens.weight)
        self._engine = None
        # picked up by esm_amd.esm2._Engine
        self._engine_no_rope = 1
        self._engine_num_positions = self.embed_positions.weight.shape[0]
        self._engine_ln_before = int(ln_before)

    @property
    def num_layers(self):
        return self.args.layers

    @num_layers.setter
    def num_layers(self, v):  # ESM2.__init__ is bypassed; kept so generic code may assign
        self.args.layers = v

    def forward(self, tokens, repr_layers=[], need_head_weights=False, return_contacts=False, **kw):
        if tokens.ndim == 2 and tokens.size(1) > self.embed_positions.max_positions:
            raise ValueError(f"Sequence length {tokens.size(1)} above maximum  sequence length of "
                             f"{self.embed_positions.max_positions}")
        return super().forward(tokens, repr_layers, need_head_weights, return_contacts, **kw)


def build_from_checkpoint(model_data):
    """``{"args": Namespace(arch="roberta_large", ...), "model": state}`` -> (model, alphabet, state), following
    reference esm/pretrained.py:87-99."""
    import argparse

    from .alphabet import Alphabet
    from .checkpoint import strip_arg_prefix, strip_key_prefix

    alphabet = Alphabet.from_architecture(model_data["args"].arch)
    # fairseq-era checkpoints: hyper-parameters are "encoder_<name>", tensors "encoder.sentence_encoder.<key>" /
    # "encoder.<key>" 
    model_args = {strip_arg_prefix(k): v for k, v in vars(model_data["args"]).items()}
    state = {strip_key_prefix(k): v for k, v in model_data["model"].items()}
    state["embed_tokens.weight"][alphabet.mask_idx].zero_()  # for token dropout (pretrained.py:97)
    model_args["emb_layer_norm_before"] = any(k.startswith("emb_layer_norm_before") for k in state)
    model = ProteinBertModel(argparse.Namespace(**model_args), alphabet)
    return model, alphabet, state
