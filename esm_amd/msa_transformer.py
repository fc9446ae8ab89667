"""MSA Transformer as a drop-in ``nn.Module`` whose forward pass runs in libesmk.so on the MI355X.

Keeps the public surface of the reference ``esm.model.msa_transformer.MSATransformer`` (reference
esm/model/msa_transformer.py:20-238): ``__init__(args, alphabet)``, attribute and state-dict key names
(``layers.{i}.row_self_attention.layer.q_proj.weight`` ...), ``forward(tokens [B,R,C], repr_layers,
need_head_weights, return_contacts)``, ``predict_contacts``, ``num_layers``, ``max_tokens_per_msa_``.
Sub-modules are parameter containers; the axial layer math (tied row attention, column attention, FFN:
reference esm/axial_attention.py, esm/modules.py:145-221,360-418) runs inside ``esmk_msa_forward``.
There is no CPU path.

``need_head_weights=True`` returns ``row_attentions [B,L,H,C,C]`` and ``col_attentions [B,L,H,C,R,R]`` as the
reference does (the latter is 58 GB in fp32 for a 12-layer 128x513 MSA: it fits the MI355X's 288 GB).  Set
``model.return_col_attentions = False`` to skip it, e.g. for ``predict_contacts`` on deep MSAs.
"""
import ctypes
import re

import torch
import torch.nn as nn

from .esm2 import (ContactPredictionHead, RobertaLMHead, _Container, _operand_dtype_for, _weight_split, live_tensors,
                   warn_if_grad_expected)

_AXIS = re.compile(r"row|column")


class _AxialAttention(_Container):
    def __init__(self, embed_dim, num_heads, max_tokens_per_msa):
        super().__init__()
        self.num_heads, self.head_dim = num_heads, embed_dim // num_heads
        self.max_tokens_per_msa = max_tokens_per_msa
        self.k_proj = nn.Linear(embed_dim, embed_dim)
        self.v_proj = nn.Linear(embed_dim, embed_dim)
        self.q_proj = nn.Linear(embed_dim, embed_dim)
        self.out_proj = nn.Linear(embed_dim, embed_dim)


class RowSelfAttention(_AxialAttention):
    pass


class ColumnSelfAttention(_AxialAttention):
    pass


class FeedForwardNetwork(_Container):
    def __init__(self, embed_dim, ffn_dim):
        super().__init__()
        self.fc1 = nn.Linear(embed_dim, ffn_dim)
        self.fc2 = nn.Linear(ffn_dim, embed_dim)


class NormalizedResidualBlock(_Container):
    def __init__(self, layer, embed_dim):
        super().__init__()
        self.layer = layer
        self.layer_norm = nn.LayerNorm(embed_dim)


class AxialTransformerLayer(_Container):
    def __init__(self, embed_dim, ffn_dim, heads, max_tokens_per_msa):
        super().__init__()
        self.row_self_attention = NormalizedResidualBlock(RowSelfAttention(embed_dim, heads, max_tokens_per_msa), embed_dim)
        self.column_self_attention = NormalizedResidualBlock(ColumnSelfAttention(embed_dim, heads, max_tokens_per_msa), embed_dim)
        self.feed_forward_layer = NormalizedResidualBlock(FeedForwardNetwork(embed_dim, ffn_dim), embed_dim)


class LearnedPositionalEmbedding(nn.Embedding):
    """Parameter container with the reference's table size (reference esm/modules.py:232-238)."""

    def __init__(self, num_embeddings, embedding_dim, padding_idx):
        super().__init__(num_embeddings + padding_idx + 1, embedding_dim, padding_idx)
        self.max_positions = num_embeddings


class _MsaEngine:
    def __init__(self, model, device, operand_dtype, weight_split=0):
        from . import _native as N

        self.N, self.device, self.operand_dtype = N, device, operand_dtype
        self.weight_split = int(weight_split)  # ESM_AMD_OPERAND=f16x2 / f16x2a (esmk_msa_config.weight_split: 1 / 2)
        a = model.args
        cfg = N.EsmkMsaConfig(
            a.layers, a.embed_dim, a.attention_heads, a.ffn_embed_dim, model.alphabet_size, model.padding_idx,
            model.mask_idx, model.cls_idx, model.eos_idx if model.eos_idx is not None else -1,
            int(bool(model.prepend_bos)), int(bool(model.append_eos)), model.embed_positions.weight.shape[0],
            int(model.msa_position_embedding is not None), N.dtype_code(operand_dtype), int(self.weight_split))
        self.handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            N.check(N.lib.esmk_msa_create(ctypes.byref(cfg), ctypes.byref(self.handle)))
            nbytes = ctypes.c_size_t()
            N.check(N.lib.esmk_packed_bytes(self.handle, ctypes.byref(nbytes)))
            self.packed = torch.zeros(nbytes.value, dtype=torch.uint8, device=device)
        self.fingerprint, self.workspace, self._named = None, None, None

    def close(self):
        if self.handle:
            self.N.lib.esmk_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync_weights(self, model):
        """See esm_amd.esm2._Engine.sync_weights (same detection rules; ``refresh_engine()`` after ``.data`` edits)."""
        N = self.N
        named = live_tensors(self, model, skip=lambda k: k == "lm_head.weight")
        fp = tuple((id(t), t.data_ptr(), t._version, t.dtype) for _, t in named)
        if fp == self.fingerprint:
            return
        stream = N.cur_stream()
        for key, t in named:
            t = t.detach()
            if key == "msa_position_embedding":  # [1,1024,1,D] (or [1,1024,1,1] in the first release) -> [1024,D]
                t = t.expand(1, t.shape[1], 1, model.args.embed_dim).reshape(t.shape[1], model.args.embed_dim)
            t = t.contiguous()
            shape = (ctypes.c_int64 * t.dim())(*t.shape)
            N.check(N.lib.esmk_pack_weight(self.handle, N.ptr(self.packed), self.packed.numel(), key.encode(),
                                           N.ptr(t), N.dtype_code(t.dtype), shape, t.dim(), stream))
        self.fingerprint = fp

    def workspace_for(self, B, R, C, flags):
        N = self.N
        need = ctypes.c_size_t()
        N.check(N.lib.esmk_msa_workspace_bytes(self.handle, B, R, C, flags, ctypes.byref(need)))
        if self.workspace is None or self.workspace.numel() < need.value:
            self.workspace = None
            self.workspace = torch.empty(need.value, dtype=torch.uint8, device=self.device)
        return self.workspace


class MSATransformer(nn.Module):
    @classmethod
    def add_args(cls, parser):
        # reference esm/model/msa_transformer.py:21-86
        parser.add_argument("--num_layers", default=12, type=int, metavar="N", help="number of layers")
        parser.add_argument("--embed_dim", default=768, type=int, metavar="N", help="embedding dimension")
        parser.add_argument("--logit_bias", action="store_true", help="whether to apply bias to logits")
        parser.add_argument("--ffn_embed_dim", default=3072, type=int, metavar="N", help="embedding dimension for FFN")
        parser.add_argument("--attention_heads", default=12, type=int, metavar="N", help="number of attention heads")
        parser.add_argument("--dropout", default=0.1, type=float, help="Dropout to apply.")
        parser.add_argument("--attention_dropout", default=0.1, type=float, help="Dropout to apply.")
        parser.add_argument("--activation_dropout", default=0.1, type=float, help="Dropout to apply.")
        parser.add_argument("--max_tokens_per_msa", default=2 ** 14, type=int,
                            help="kept for compatibility: the engine never needs to chunk the attention")

    def __init__(self, args, alphabet):
        super().__init__()
        self.args = args
        self.alphabet_size = len(alphabet)
        self.padding_idx = alphabet.padding_idx
        self.mask_idx = alphabet.mask_idx
        self.cls_idx = alphabet.cls_idx
        self.eos_idx = alphabet.eos_idx
        self.prepend_bos = alphabet.prepend_bos
        self.append_eos = alphabet.append_eos
        E = args.embed_dim
        self.embed_tokens = nn.Embedding(self.alphabet_size, E, padding_idx=self.padding_idx)
        if getattr(args, "embed_positions_msa", False):
            emb_dim = getattr(args, "embed_positions_msa_dim", E)
            self.msa_position_embedding = nn.Parameter(0.01 * torch.randn(1, 1024, 1, emb_dim), requires_grad=True)
        else:
            self.register_parameter("msa_position_embedding", None)
        # reference msa_transformer.py:114: a parameter-less child kept for the module surface (named_children order);
        # the engine is forward-only and refuses training mode with non-zero dropout (see forward)
        self.dropout_module = nn.Dropout(getattr(args, "dropout", 0.0))
        mtpm = getattr(args, "max_tokens_per_msa", getattr(args, "max_tokens", 2 ** 14))
        self.layers = nn.ModuleList(
            [AxialTransformerLayer(E, args.ffn_embed_dim, args.attention_heads, mtpm) for _ in range(args.layers)])
        self.contact_head = ContactPredictionHead(args.layers * args.attention_heads, self.prepend_bos, self.append_eos,
                                                  eos_idx=self.eos_idx)
        self.embed_positions = LearnedPositionalEmbedding(args.max_positions, E, self.padding_idx)
        self.emb_layer_norm_before = nn.LayerNorm(E)
        self.emb_layer_norm_after = nn.LayerNorm(E)
        self.lm_head = RobertaLMHead(E, self.alphabet_size, self.embed_tokens.weight)
        self._engine = None
        self.return_col_attentions = True

    @property
    def num_layers(self):
        return self.args.layers

    def max_tokens_per_msa_(self, value: int) -> None:
        """reference msa_transformer.py:229-238 (the engine holds the whole MSA in HBM; kept as a no-op knob)."""
        for module in self.modules():
            if isinstance(module, (RowSelfAttention, ColumnSelfAttention)):
                module.max_tokens_per_msa = value

    def _get_engine(self, device):
        odt = _operand_dtype_for(self.embed_tokens.weight.dtype)
        split = _weight_split()
        eng = self._engine
        if eng is None or eng.device != device or eng.operand_dtype != odt or eng.weight_split != split:
            if eng is not None:
                eng.close()
            eng = _MsaEngine(self, device, odt, split)
            object.__setattr__(self, "_engine", eng)
        return eng

    def forward(self, tokens, repr_layers=[], need_head_weights=False, return_contacts=False):
        if return_contacts:
            need_head_weights = True
        assert tokens.ndim == 3
        if self.training and (self.args.dropout or self.args.attention_dropout or self.args.activation_dropout):
            raise RuntimeError("esm_amd.MSATransformer is forward-only: call .eval() (dropout is not implemented)")
        if not tokens.is_cuda:
            raise RuntimeError("esm_amd.MSATransformer runs only on an MI355X (ROCm) device; there is no CPU fallback")
        w = self.embed_tokens.weight
        if w.device != tokens.device:
            raise RuntimeError(f"model parameters are on {w.device} but tokens on {tokens.device}")
        warn_if_grad_expected(self)
        from . import _native as N

        dev = tokens.device
        B, R, C = tokens.shape
        if self.msa_position_embedding is not None and R > 1024:
            raise RuntimeError("Using model with MSA position embedding trained on maximum MSA "
                               f"depth of 1024, but received {R} alignments.")
        if C > self.embed_positions.max_positions:
            raise ValueError(f"Sequence length {C} above maximum  sequence length of {self.embed_positions.max_positions}")
        L, E, H, V = self.args.layers, self.args.embed_dim, self.args.attention_heads, self.alphabet_size
        repr_set = sorted({int(i) for i in repr_layers if 0 <= int(i) <= L})
        with torch.cuda.device(dev):
            eng = self._get_engine(dev)
            eng.sync_weights(self)
            tok = tokens.to(torch.int64).contiguous()
            flags = N.OUT_LOGITS
            f32 = dict(dtype=torch.float32, device=dev)
            logits = torch.empty((B, R, C, V), **f32)
            reps = [torch.empty((B, R, C, E), **f32) for _ in repr_set]
            row_attn = col_attn = contacts = None
            if need_head_weights:
                flags |= N.OUT_ATTN
                row_attn = torch.empty((B, L, H, C, C), **f32)
                if self.return_col_attentions:
                    flags |= N.OUT_COL_ATTN
                    col_attn = torch.empty((B, L, H, C, R, R), **f32)
            if return_contacts:
                flags |= N.OUT_CONTACTS
                S = C - int(self.prepend_bos) - int(self.append_eos)
                contacts = torch.empty((B, S, S), **f32)
            ws = eng.workspace_for(B, R, C, flags)
            layers_arr = (ctypes.c_int32 * max(1, len(repr_set)))(*repr_set)
            outs_arr = (ctypes.c_void_p * max(1, len(repr_set)))(*[r.data_ptr() for r in reps])
            N.check(N.lib.esmk_msa_forward(
                eng.handle, N.ptr(eng.packed), N.ptr(tok), B, R, C, layers_arr, len(repr_set), outs_arr, flags,
                N.ptr(logits), N.ptr(row_attn), N.ptr(col_attn), N.ptr(contacts), N.ptr(ws), ws.numel(), N.cur_stream()))
        out_dt = w.dtype
        cast = (lambda t: t) if out_dt == torch.float32 else (lambda t: t.to(out_dt))
        result = {"logits": cast(logits), "representations": {l: cast(r) for l, r in zip(repr_set, reps)}}
        if need_head_weights:
            if col_attn is not None:
                result["col_attentions"] = cast(col_attn)
            result["row_attentions"] = cast(row_attn)
            if return_contacts:
                result["contacts"] = cast(contacts)
        return result

    def predict_contacts(self, tokens):
        return self(tokens, return_contacts=True)["contacts"]

    def profile_begin(self):
        """Arm per-kernel-class HIP-event timing of the following forward calls (bench.py --workload msa1b)."""
        from .esm2 import ESM2

        ESM2.profile_begin(self)

    def profile_end(self):
        from .esm2 import ESM2

        return ESM2.profile_end(self)

    def refresh_engine(self):
        if self._engine is not None:
            self._engine.close()
        object.__setattr__(self, "_engine", None)

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_engine"] = None
        return state


def build_from_checkpoint(model_data):
    """``{"args": Namespace(arch="msa_transformer", ...), "model": state}`` -> (model, alphabet), following
    reference esm/pretrained.py:111-127 (prefix stripping, embed_positions_msa_dim from the tensor)."""
    from .alphabet import Alphabet
    from .checkpoint import strip_arg_prefix, strip_key_prefix

    args = model_data["args"]
    # the released checkpoints name the two axial attentions the other way round from the module attributes
    # (pretrained.py:111-124): "row" <-> "column" in every tensor name, after the usual prefix stripping
    swap_axes = lambda key: _AXIS.sub(lambda m: "column" if m.group(0) == "row" else "row", key)
    model_args = {strip_arg_prefix(k): v for k, v in vars(args).items()}
    state = {strip_key_prefix(swap_axes(k)): v for k, v in model_data["model"].items()}
    if model_args.get("embed_positions_msa", False):
        model_args["embed_positions_msa_dim"] = state["msa_position_embedding"].size(-1)
    import argparse

    alphabet = Alphabet.from_architecture("msa_transformer")
    model = MSATransformer(argparse.Namespace(**model_args), alphabet)
    return model, alphabet, state
