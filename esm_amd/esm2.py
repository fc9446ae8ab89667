"""ESM-2 as a drop-in ``nn.Module`` whose forward pass runs in libesmk.so on the MI355X.

The class keeps the public surface of the reference ``esm.model.esm2.ESM2`` (reference
esm/model/esm2.py:15-147): constructor arguments, attribute names, state-dict key names
(``layers.{i}.self_attn.q_proj.weight`` ...), ``forward(tokens, repr_layers, need_head_weights,
return_contacts)`` and ``predict_contacts``.  The sub-modules below are *parameter containers*
with the reference's names; no layer math is done in Python/torch — ``forward`` hands raw device
pointers to ``esmk_forward`` (include/esmk.h).  There is no CPU path: CPU tensors raise.
"""
import ctypes
import os
import warnings
from typing import Union

import torch
import torch.nn as nn

from .alphabet import Alphabet


class _Container(nn.Module):
    """Holds parameters under the reference's names; calling it is an error by design."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError(
            f"{type(self).__name__} is a parameter container of the MI355X engine; "
            "the layer math runs inside ESM2.forward (libesmk.so), not per sub-module"
        )


class RotaryEmbedding(_Container):
    """Carries the ``inv_freq`` buffer of reference esm/rotary_embedding.py:40-41."""

    def __init__(self, dim: int):
        super().__init__()
        inv_freq = 1.0 / (10000 ** (torch.arange(0, dim, 2).float() / dim))
        self.register_buffer("inv_freq", inv_freq)


class MultiheadAttention(_Container):
    def __init__(self, embed_dim, num_heads):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == embed_dim, "embed_dim must be divisible by num_heads"
        self.scaling = self.head_dim ** -0.5
        self.k_proj = nn.Linear(embed_dim, embed_dim)
        self.v_proj = nn.Linear(embed_dim, embed_dim)
        self.q_proj = nn.Linear(embed_dim, embed_dim)
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        self.rot_emb = RotaryEmbedding(self.head_dim)


class TransformerLayer(_Container):
    def __init__(self, embed_dim, ffn_embed_dim, attention_heads):
        super().__init__()
        self.embed_dim, self.ffn_embed_dim, self.attention_heads = embed_dim, ffn_embed_dim, attention_heads
        self.self_attn = MultiheadAttention(embed_dim, attention_heads)
        self.self_attn_layer_norm = nn.LayerNorm(embed_dim)
        self.fc1 = nn.Linear(embed_dim, ffn_embed_dim)
        self.fc2 = nn.Linear(ffn_embed_dim, embed_dim)
        self.final_layer_norm = nn.LayerNorm(embed_dim)


class RobertaLMHead(_Container):
    def __init__(self, embed_dim, output_dim, weight):
        super().__init__()
        self.dense = nn.Linear(embed_dim, embed_dim)
        self.layer_norm = nn.LayerNorm(embed_dim)
        self.weight = weight  # tied to embed_tokens.weight
        self.bias = nn.Parameter(torch.zeros(output_dim))


class ContactPredictionHead(_Container):
    def __init__(self, in_features, prepend_bos, append_eos, bias=True, eos_idx=None):
        super().__init__()
        self.in_features, self.prepend_bos, self.append_eos = in_features, prepend_bos, append_eos
        if append_eos and eos_idx is None:
            raise ValueError("Using an alphabet with eos token, but no eos token was passed in.")
        self.eos_idx = eos_idx
        self.regression = nn.Linear(in_features, 1, bias)


def _native_lowp(param_dtype, operand_dtype):
    """``.half()`` / ``.bfloat16()`` models return fp16 / bf16 tensors (the reference runs the whole module in that
    dtype; ESMFold's front end does so, esmfold/v1/esmfold.py:61-67).  When the model dtype is the engine's operand
    dtype the engine writes representations / attention maps in it directly (ESMK_OUT_REPR_LOWP / _ATTN_LOWP);
    ``ESM_AMD_NATIVE_LOWP=0`` falls back to fp32 outputs + a cast (same bits, one more pass; used by the tests)."""
    if os.environ.get("ESM_AMD_NATIVE_LOWP", "1") == "0":
        return False
    return param_dtype in (torch.float16, torch.bfloat16) and param_dtype == operand_dtype


def _weight_split():
    """``ESM_AMD_OPERAND=f16x2``: precision mode with split weights (W = W_hi + W_lo, both fp16, two MFMA passes per
    layer GEMM): removes the weight rounding — two thirds of the fp16-operand error of a deep stack — at 2x the GEMM
    time.  ESM-2, ESM-1b and (since round 4) the MSA Transformer engine.
    ``ESM_AMD_OPERAND=f16x2a`` (round 6): the same for the ATTENTION projections only (q, k, v, out: a third of the GEMM
    work) — representations and logits inside 1e-3 in both norms at ~1.3x the plain step instead of 1.6x.
    ``ESM_AMD_OPERAND=f16x2v``: the VALUE path only (v, out: a sixth of the GEMM work, ~1.2x) — most of f16x2a's gain on
    representations and logits; q / k rounding matters for the attention maps / contact logits only.
    ``ESM_AMD_OPERAND=f16x3``: weights AND GEMM inputs split (every layer GEMM a plain launch over K' = 3 K: A_hi W_hi +
    A_hi W_lo + A_lo W_hi) — the mode that holds 1e-3 on EVERY output, contact logits included, at ~2.4x the step;
    head_dim-64 models, padded batches (``forward_varlen`` falls back to ``forward``).
    Returns esmk_config.weight_split: 0 off, 1 f16x2, 2 f16x2a, 3 f16x2v, 4 f16x3."""
    env = os.environ.get("ESM_AMD_OPERAND", "").lower()
    return {"f16x2": 1, "fp16x2": 1, "f16x2a": 2, "fp16x2a": 2, "f16x2v": 3, "fp16x2v": 3, "f16x3": 4, "fp16x3": 4}.get(env, 0)


def _ln_fold():
    """``ESM_AMD_LN_FOLD=1|0``: LayerNorm fold of the engine (esmk_config.ln_fold) on / off; unset = the
    library's default.  ESM-2 / ESM-1b engines with plain fp16 / bf16 operands and head_dim <= 64."""
    v = os.environ.get("ESM_AMD_LN_FOLD", "")
    return 0 if v == "" else (1 if v not in ("0", "off", "false") else -1)


# The LayerNorm fold and small LayerNorm gains (round 6, tools/outlier_stress_study.py, profiles/r6_outlier_stress_study.log).
# The fold's consumers run on gain-folded, row-centred weight images: column j of an image holds gamma_j w_ij - c_i with
# c_i = mean_k(gamma_k w_ik), and its operand rows are the un-normalised fp16(x - mean).  A channel whose gain is far below
# the others (|gamma_j| << median / sqrt(E)) holds almost nothing but -c_i; if the checkpoint uses that small gain to silence
# a large activation (the "massive activation" channels of trained transformers), the fp16 rounding of x_j and of c_i is
# multiplied by that large x_j: the fold's error grows with x_j / (E s) (s: the spread of the ordinary channels) while the
# plain mode — which rounds the normalised value gamma_j (x_j - mean) rstd — does not see the channel at all.  (Same-signed
# outliers add a second term: they shift the row mean, the LayerNorm bias takes the shift back — exactly, as fp32 W . beta, in
# the fold, against a counterpart that went through the rounded image)  Measured on
# the stress weights of esm_amd.synth.add_outlier_channels (650M dims, four channels): gain ratio 133 (outliers 200 x the
# stream) -> fold / plain floor 1.1; 1333 -> 3.3 ... 4.2; 13333 -> 34.  The hazard of one LayerNorm, from its gains alone:
#     h = sum over channels with |gamma_j| < median / 8 of (median / |gamma_j|) / E
# and of a model: the mean over its folded LayerNorms (0.35 / 3.5 / 35 on those three sets; the stream of the first layers
# is small, so their ratios are the largest).  When ESM_AMD_LN_FOLD is unset, a model whose h exceeds 0.5 runs WITHOUT the
# fold (the standalone LayerNorm passes:
# - 1.1 % at B = 64, - 6 % at B = 4); ESM_AMD_LN_FOLD=1 forces it on, =0 off.  Callers of the C ABI choose esmk_config.ln_fold
# themselves.
_FOLD_HAZARD_MAX = 0.5


def ln_fold_hazard(gains):
    """``gains``: [n_layernorms, E] LayerNorm weights whose outputs feed folded GEMMs.  Returns their mean h (see above)."""
    g = gains.detach().float().abs()
    med = g.median(dim=-1, keepdim=True).values
    small = g < med / 8
    h = torch.where(small, med / g.clamp_min(1e-30), torch.zeros_like(g)).sum(-1) / g.shape[-1]
    return float(h.mean().item()) if h.numel() else 0.0


# Small and medium batches as TWO half-batches on two HIP streams (round 6).  Below ~56 k rows the persistent GEMMs end in
# partly filled rounds of tiles over the 256 CUs (B = 8 x 1024 tokens: fc2 has 320 half-height tiles = 1.25 rounds);
# workgroups without a tile exit at once, so the kernels of a second, independent half-batch take the idle CUs.  Measured on
# one box (650M dims, profiles/r6_dual_stream_probe.log), rows -> gain: 4096 + 3.9 %, 6144 - 1.2 %, 8192 + 9.4 %, 12288
# + 2.4 %, 16384 + 7.9 %, 24576 + 2.4 %, 32768 + 2.1 %, 40960 + 6.9 %, 49152 + 2.2 %, 65536 + 0.6 % (whole rounds already);
# the same per row count for other (B, T) shapes.  Sequences are independent and every kernel of the forward is batch-invariant
# bit for bit, so the results are the bits of the one-stream forward (the fused contact map of predict_contacts, whose head
# grouping depends on the batch size, stays on one stream).  ``ESM_AMD_DUAL_STREAM=0`` switches it off,
# ``=lo:hi[,lo:hi...]`` sets the row windows (tokens per forward call).
def _dual_stream_window():
    v = os.environ.get("ESM_AMD_DUAL_STREAM", "")
    if v in ("0", "off", "false"):
        return None
    if ":" in v:
        return [tuple(int(x) for x in w.split(":", 1)) for w in v.split(",")]
    return [(3584, 5120), (7168, 57344)]


def _dual_stream_wanted(rows):
    win = _dual_stream_window()
    return win is not None and any(lo <= rows <= hi for lo, hi in win)


def _operand_dtype_for(param_dtype):
    env = os.environ.get("ESM_AMD_OPERAND", "").lower()
    if env in ("bf16", "bfloat16"):
        return torch.bfloat16
    if env in ("f16", "fp16", "float16", "half", "f16x2", "fp16x2", "f16x2a", "fp16x2a", "f16x2v", "fp16x2v", "f16x3", "fp16x3"):
        return torch.float16
    # fp16 operands keep the 33-layer stack within 1e-3 of the fp32 reference (bf16: ~5e-3)
    return torch.bfloat16 if param_dtype == torch.bfloat16 else torch.float16


def live_tensors(engine, model, skip):
    """[(state-dict key, tensor)] of the model's CURRENT parameters and buffers.  The (owner dict, name) slots are
    collected once per engine; reading them back costs a dict lookup per tensor, so a replaced Parameter object is
    picked up without walking the module tree on every forward."""
    if engine._named is None:
        slots = []
        for prefix, mod in model.named_modules():
            for store in (mod._parameters, mod._buffers):
                for name, t in store.items():
                    key = f"{prefix}.{name}" if prefix else name
                    if t is not None and not skip(key) and name not in getattr(mod, "_non_persistent_buffers_set", ()):
                        slots.append((key, store, name))
        engine._named = slots
    return [(key, store[name]) for key, store, name in engine._named]


def check_finite(result):
    """``ESM_AMD_CHECK_FINITE=1`` (debug aid, synchronises): raise if an output holds inf / NaN.  The engine rounds
    GEMM operands to fp16 (range 65504) also for fp32 models; this has been validated on seeded synthetic weights
    only (no released checkpoint is available offline), so a first run on real weights can be checked this way.
    Pad positions are included: the reference leaves finite garbage there as well."""
    if os.environ.get("ESM_AMD_CHECK_FINITE", "0") != "1":
        return
    def walk(prefix, v):
        if isinstance(v, dict):
            for k, t in v.items():
                walk(f"{prefix}[{k!r}]", t)
        elif torch.is_tensor(v) and v.is_floating_point() and not bool(torch.isfinite(v).all()):
            raise FloatingPointError(f"esm_amd: {prefix} contains inf / NaN (fp16 operand overflow? try ESM_AMD_OPERAND=bf16)")
    walk("out", result)


def warn_if_grad_expected(model):
    """The engine is forward-only: outputs carry no grad_fn.  The reference's own tests call forward without
    ``no_grad`` (tests/test_load_all.py:39-47), so this warns — once per model — instead of raising."""
    if torch.is_grad_enabled() and not getattr(model, "_warned_no_grad", False):
        if any(p.requires_grad for p in model.parameters()):
            warnings.warn(
                "esm_amd: the MI355X engine is forward-only — the tensors returned by forward() have no grad_fn, so "
                "backward() through this model yields no parameter gradients. Wrap inference in torch.no_grad() or "
                "call model.requires_grad_(False) to silence this warning.", RuntimeWarning, stacklevel=3)
            # only once the warning was actually emitted: a later model.requires_grad_(True) must still be told
            object.__setattr__(model, "_warned_no_grad", True)


class _Engine:
    """One esmk_model handle + packed parameter image + workspace for one (device, dtype)."""

    def __init__(self, model: "ESM2", device, operand_dtype, weight_split=0, ln_fold=None):
        from . import _native as N

        self.N = N
        self.device = device
        self.operand_dtype = operand_dtype
        self.weight_split = int(weight_split)  # esmk_config.weight_split: 0 off, 1 f16x2, 2 f16x2a, 3 f16x2v
        # ESM_AMD_LN_FOLD (or the gain check of ESM2._fold_setting) at creation: a changed setting makes a new engine
        self.ln_fold = _ln_fold() if ln_fold is None else int(ln_fold)
        # ESM-1b / ESM-1v (esm_amd.esm1.ProteinBertModel) set these; ESM-2 leaves them at zero
        self.no_rope = int(getattr(model, "_engine_no_rope", 0))
        num_positions = int(getattr(model, "_engine_num_positions", 0))
        ln_before = int(getattr(model, "_engine_ln_before", 0))
        cfg = N.EsmkConfig(
            model.num_layers, model.embed_dim, model.attention_heads, int(getattr(model, "ffn_embed_dim", 4 * model.embed_dim)),
            model.alphabet_size, model.padding_idx, model.mask_idx, model.cls_idx, model.eos_idx,
            int(bool(model.token_dropout)), int(bool(model.prepend_bos)), int(bool(model.append_eos)),
            N.dtype_code(operand_dtype), self.no_rope, num_positions, ln_before, int(self.weight_split),
            # the fold is asked for only where the library supports it (plain operands, head_dim <= 64); elsewhere "default"
            self.ln_fold if (not self.weight_split and model.embed_dim // model.attention_heads <= 64) or self.ln_fold < 0 else 0,
        )
        self.handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            N.check(N.lib.esmk_create(ctypes.byref(cfg), ctypes.byref(self.handle)))
            if not self.no_rope:
                d = model.embed_dim // model.attention_heads
                inv = (1.0 / (10000 ** (torch.arange(0, d, 2).float() / d))).tolist()
                arr = (ctypes.c_float * len(inv))(*inv)
                N.check(N.lib.esmk_set_rope_inv_freq(self.handle, arr, len(inv)))
            nbytes = ctypes.c_size_t()
            N.check(N.lib.esmk_packed_bytes(self.handle, ctypes.byref(nbytes)))
            # zero-initialised: padded head slots / K columns of the packed image must stay zero
            self.packed = torch.zeros(nbytes.value, dtype=torch.uint8, device=device)
        self.fingerprint = None
        self.workspace = None
        self.workspace2 = None   # second half-batch of the dual-stream forward
        self.stream2 = None
        self.max_T = 0           # longest row a finished forward call has seen (its RoPE table exists and is ordered before us)
        self.profiling = False
        self.dual_calls = 0
        self._named = None

    def close(self):
        if self.handle:
            self.N.lib.esmk_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def workspace_for_packed(self, n_seg, rows, flags):
        N = self.N
        need = ctypes.c_size_t()
        N.check(N.lib.esmk_packed_workspace_bytes(self.handle, n_seg, rows, flags, ctypes.byref(need)))
        if self.workspace is None or self.workspace.numel() < need.value:
            self.workspace = None
            self.workspace = torch.empty(need.value, dtype=torch.uint8, device=self.device)
        return self.workspace

    def sync_weights(self, model):
        """Re-pack the parameter image when the parameters changed: ``.cuda()`` / ``.half()`` /
        ``load_state_dict`` (also with ``assign=True``), ``module.weight = nn.Parameter(...)``, swapped tensors and
        tracked in-place edits are all seen (live tensors are looked up on every call; fingerprint = object id,
        storage address, version counter, dtype).  NOT seen: writes through ``param.data`` (they bypass the version
        counter) and replaced sub-modules — call ``model.refresh_engine()`` after those."""
        N = self.N
        named = live_tensors(self, model, skip=lambda k: k == "lm_head.weight" or k.endswith("inv_freq"))
        fp = tuple((id(t), t.data_ptr(), t._version, t.dtype) for _, t in named)
        if fp == self.fingerprint:
            return False
        repacked = self.fingerprint is not None  # (the first pack of a new engine is not a change of the parameters)
        stream = N.cur_stream()
        # LayerNorm parameters first: with the LayerNorm fold the q/k/v and fc1 weights are folded with them at pack time
        for key, t in sorted(named, key=lambda kt: 0 if "layer_norm" in kt[0] else 1):
            t = t.detach()
            if not t.is_contiguous():
                t = t.contiguous()
            shape = (ctypes.c_int64 * t.dim())(*t.shape)
            N.check(N.lib.esmk_pack_weight(self.handle, N.ptr(self.packed), self.packed.numel(),
                                           key.encode(), N.ptr(t), N.dtype_code(t.dtype), shape, t.dim(),
                                           stream))
        self.fingerprint = fp
        return repacked

    def workspace_for(self, B, T, flags):
        N = self.N
        need = ctypes.c_size_t()
        N.check(N.lib.esmk_workspace_bytes(self.handle, B, T, flags, ctypes.byref(need)))
        if self.workspace is None or self.workspace.numel() < need.value:
            self.workspace = None
            self.workspace = torch.empty(need.value, dtype=torch.uint8, device=self.device)
        return self.workspace


class ESM2(nn.Module):
    def __init__(
        self,
        num_layers: int = 33,
        embed_dim: int = 1280,
        attention_heads: int = 20,
        alphabet: Union[Alphabet, str] = "ESM-1b",
        token_dropout: bool = True,
    ):
        super().__init__()
        self.num_layers = num_layers
        self.embed_dim = embed_dim
        self.attention_heads = attention_heads
        if not isinstance(alphabet, Alphabet):
            alphabet = Alphabet.from_architecture(alphabet)
        self.alphabet = alphabet
        self.alphabet_size = len(alphabet)
        self.padding_idx = alphabet.padding_idx
        self.mask_idx = alphabet.mask_idx
        self.cls_idx = alphabet.cls_idx
        self.eos_idx = alphabet.eos_idx
        self.prepend_bos = alphabet.prepend_bos
        self.append_eos = alphabet.append_eos
        self.token_dropout = token_dropout
        self._engine = None
        self._init_submodules()

    def _init_submodules(self):
        self.embed_scale = 1
        self.embed_tokens = nn.Embedding(self.alphabet_size, self.embed_dim, padding_idx=self.padding_idx)
        self.layers = nn.ModuleList(
            [TransformerLayer(self.embed_dim, 4 * self.embed_dim, self.attention_heads) for _ in range(self.num_layers)]
        )
        self.contact_head = ContactPredictionHead(
            self.num_layers * self.attention_heads, self.prepend_bos, self.append_eos, eos_idx=self.eos_idx
        )
        self.emb_layer_norm_after = nn.LayerNorm(self.embed_dim)
        self.lm_head = RobertaLMHead(self.embed_dim, self.alphabet_size, self.embed_tokens.weight)

    # ------------------------------------------------------------------------------------------
    def _get_engine(self, device):
        pdt = self.embed_tokens.weight.dtype
        odt = _operand_dtype_for(pdt)
        split = _weight_split()
        fold = self._fold_setting()
        eng = self._engine
        if (eng is None or eng.device != device or eng.operand_dtype != odt or eng.weight_split != split
                or eng.ln_fold != fold):
            if eng is not None:
                eng.close()
            eng = _Engine(self, device, odt, split, fold)
            object.__setattr__(self, "_engine", eng)
        return eng

    def _fold_setting(self):
        """esmk_config.ln_fold for this model: ESM_AMD_LN_FOLD when set; otherwise 0 (the library's default: on where it is
        supported) unless the LayerNorm gains in front of the q/k/v and fc1 projections make the fold's operand form lossy
        (``ln_fold_hazard`` above) — then -1.  The gains are read when the engine's parameter fingerprint changes
        (``_engine_ready``), not per call."""
        env = _ln_fold()
        if env != 0:
            return env
        h = self.__dict__.get("_fold_hazard")
        if h is None:
            gains = [l.weight.detach() for layer in self.layers for l in (layer.self_attn_layer_norm, layer.final_layer_norm)]
            h = ln_fold_hazard(torch.stack(gains)) if gains else 0.0
            object.__setattr__(self, "_fold_hazard", h)
        return -1 if h > _FOLD_HAZARD_MAX else 0

    def _engine_ready(self, device):
        """The engine for this call with the current parameters packed.  Changed parameters (``_Engine.sync_weights``) may have
        changed the LayerNorm gains: the fold decision is taken again, and an engine of the other mode is made if it flipped."""
        eng = self._get_engine(device)
        if eng.sync_weights(self) and _ln_fold() == 0:
            object.__setattr__(self, "_fold_hazard", None)
            again = self._get_engine(device)
            if again is not eng:
                again.sync_weights(self)
            eng = again
        return eng

    def forward(self, tokens, repr_layers=[], need_head_weights=False, return_contacts=False, contacts_only=False):
        """Reference esm/model/esm2.py:77-144.  ``contacts_only=True`` (engine extension, used by ``predict_contacts``
        and the extraction driver) returns ``{"contacts", "representations"}`` only: no logits and no [B,L,H,T,T]
        "attentions" tensor is built — the contact map is accumulated layer by layer (csrc/contacts.hip)."""
        if contacts_only:
            return_contacts, need_head_weights = True, False
        if return_contacts and not contacts_only:
            need_head_weights = True
        assert tokens.ndim == 2
        if not tokens.is_cuda:
            raise RuntimeError(
                "esm_amd.ESM2 runs only on an MI355X (ROCm) device: move the model and tokens to "
                "'cuda' first; the engine has no CPU fallback"
            )
        w = self.embed_tokens.weight
        if w.device != tokens.device:
            raise RuntimeError(f"model parameters are on {w.device} but tokens on {tokens.device}")
        warn_if_grad_expected(self)
        from . import _native as N

        dev = tokens.device
        B, T = tokens.shape
        L, E, H, V = self.num_layers, self.embed_dim, self.attention_heads, self.alphabet_size
        repr_set = sorted({int(i) for i in repr_layers if 0 <= int(i) <= L})
        with torch.cuda.device(dev):
            eng = self._engine_ready(dev)
            tok = tokens.to(torch.int64).contiguous()
            # predict_contacts: no logits, no attention tensor (contacts.hip accumulates the map layer by layer)
            flags = 0 if contacts_only else N.OUT_LOGITS
            f32 = dict(dtype=torch.float32, device=dev)
            lowp = _native_lowp(w.dtype, eng.operand_dtype)
            logits = None if contacts_only else torch.empty((B, T, V), **f32)
            if lowp and repr_set:
                flags |= N.OUT_REPR_LOWP
            reps = [torch.empty((B, T, E), dtype=w.dtype if lowp else torch.float32, device=dev) for _ in repr_set]
            attn = contacts = None
            if need_head_weights:
                flags |= N.OUT_ATTN
                if lowp and not return_contacts:  # the contact kernels read fp32 maps
                    flags |= N.OUT_ATTN_LOWP
                    attn = torch.empty((B, L, H, T, T), dtype=w.dtype, device=dev)
                else:
                    attn = torch.empty((B, L, H, T, T), **f32)
            if return_contacts:
                S = max(T - int(self.prepend_bos) - int(self.append_eos), 0)
                contacts = torch.empty((B, S, S), **f32)
                if S > 0:  # empty sequences: the reference returns an empty [B,0,0] map
                    flags |= N.OUT_CONTACTS
            layers_arr = (ctypes.c_int32 * max(1, len(repr_set)))(*repr_set)
            if (B >= 2 and not contacts_only and _dual_stream_wanted(B * T) and T <= eng.max_T and not eng.profiling
                    and not torch.cuda.is_current_stream_capturing()):
                # two half-batches, the second on the engine's own stream (see _dual_stream_window): same bits
                cur = torch.cuda.current_stream(dev)
                if eng.stream2 is None:
                    eng.stream2 = torch.cuda.Stream(dev)
                h = (B + 1) // 2
                need = ctypes.c_size_t()
                N.check(N.lib.esmk_workspace_bytes(eng.handle, h, T, flags, ctypes.byref(need)))
                ws = eng.workspace_for(h, T, flags)
                if eng.workspace2 is None or eng.workspace2.numel() < need.value:
                    eng.workspace2 = None
                    eng.workspace2 = torch.empty(need.value, dtype=torch.uint8, device=dev)
                ready = torch.cuda.Event()
                ready.record(cur)                      # tokens, weights and the output allocations are ordered before the side stream
                eng.stream2.wait_event(ready)
                for lo, hi, wsp, st in ((0, h, ws, cur), (h, B, eng.workspace2, eng.stream2)):
                    part = lambda t: None if t is None else t[lo:hi]
                    outs_arr = (ctypes.c_void_p * max(1, len(repr_set)))(*[r[lo:hi].data_ptr() for r in reps])
                    N.check(N.lib.esmk_forward(
                        eng.handle, N.ptr(eng.packed), N.ptr(tok[lo:hi]), hi - lo, T, layers_arr, len(repr_set), outs_arr,
                        flags, N.ptr(part(logits)), N.ptr(part(attn)), N.ptr(part(contacts)), N.ptr(wsp), wsp.numel(),
                        ctypes.c_void_p(st.cuda_stream)))
                done = torch.cuda.Event()
                done.record(eng.stream2)
                cur.wait_event(done)                   # the caller's stream sees both halves
                eng.dual_calls += 1
            else:
                ws = eng.workspace_for(B, T, flags)
                outs_arr = (ctypes.c_void_p * max(1, len(repr_set)))(*[r.data_ptr() for r in reps])
                N.check(N.lib.esmk_forward(
                    eng.handle, N.ptr(eng.packed), N.ptr(tok), B, T, layers_arr, len(repr_set), outs_arr,
                    flags, N.ptr(logits), N.ptr(attn), N.ptr(contacts), N.ptr(ws), ws.numel(), N.cur_stream()))
                eng.max_T = max(eng.max_T, T)
        out_dt = w.dtype
        cast = lambda t: t if t.dtype == out_dt else t.to(out_dt)
        if contacts_only:
            result = {"contacts": cast(contacts), "representations": {l: cast(r) for l, r in zip(repr_set, reps)}}
            check_finite(result)
            return result
        result = {"logits": cast(logits), "representations": {l: cast(r) for l, r in zip(repr_set, reps)}}
        if need_head_weights:
            result["attentions"] = cast(attn)
            if return_contacts:
                result["contacts"] = cast(contacts)
        check_finite(result)
        return result

    # ------------------------------------------------------------------------------------------
    # token-packed batches: no compute on padding (include/esmk.h, esmk_forward_packed)
    supports_varlen = True  # ESM-2 (all sizes) and ESM-1b / ESM-1v; the MSA Transformer has no such path
    supports_contacts_only = True  # forward(contacts_only=True): contact maps without the attention tensor

    def forward_varlen(self, tokens, repr_layers=[], lengths=None, min_saving=0.08, unpack=True):
        """Same results as ``forward(tokens, repr_layers)`` on the non-pad positions of a RIGHT-padded batch
        (what ``BatchConverter`` yields, reference esm/data.py:262-297), but the sequences are laid back to back
        in one row space and the engine does no work on padding.  Pad positions of the returned tensors are zero
        (the reference leaves the values the pad rows happened to compute there).

        tokens   [B,T] int64, CPU or device; from a CPU tensor the lengths are read without a device sync
        lengths  optional per-row token counts (incl. <cls>/<eos>); default: up to the last non-pad token
        min_saving  fall back to ``forward`` when packing saves less than this fraction of the rows
                    (None: always pack)
        unpack   False: return the packed tensors ([rows, .]) plus ``segments`` ([B,2] first row, length)

        Attention maps and contacts are per-sequence [T,T] objects: use ``forward`` for those."""
        assert tokens.ndim == 2
        from . import _native as N
        from .packing import pack_plan

        w = self.embed_tokens.weight
        if not w.is_cuda:
            raise RuntimeError("esm_amd.ESM2 runs only on an MI355X (ROCm) device; the engine has no CPU fallback")
        warn_if_grad_expected(self)
        dev = w.device
        B, T = tokens.shape
        L, E, V = self.num_layers, self.embed_dim, self.alphabet_size
        plan = pack_plan(tokens, self.padding_idx, lengths)
        if unpack and ((min_saving is not None and plan.rows > (1.0 - min_saving) * B * T) or _weight_split() == 4):
            return self.forward(tokens.to(dev), repr_layers=repr_layers)  # (f16x3 has no token-packed form)
        repr_set = sorted({int(i) for i in repr_layers if 0 <= int(i) <= L})
        with torch.cuda.device(dev):
            eng = self._engine_ready(dev)
            idx, keep = plan.index(dev)
            flat = plan.pack(tokens, self.padding_idx, idx)
            f32 = dict(dtype=torch.float32, device=dev)
            lowp = _native_lowp(w.dtype, eng.operand_dtype) and bool(repr_set)
            pflags = N.OUT_LOGITS | (N.OUT_REPR_LOWP if lowp else 0)
            logits = torch.empty((plan.rows, V), **f32)
            reps = [torch.empty((plan.rows, E), dtype=w.dtype if lowp else torch.float32, device=dev) for _ in repr_set]
            ws = eng.workspace_for_packed(B, plan.rows, pflags)
            seg = plan.segments  # int32 [B,2], CPU, contiguous
            layers_arr = (ctypes.c_int32 * max(1, len(repr_set)))(*repr_set)
            outs_arr = (ctypes.c_void_p * max(1, len(repr_set)))(*[r.data_ptr() for r in reps])
            N.check(N.lib.esmk_forward_packed(
                eng.handle, N.ptr(eng.packed), N.ptr(flat),
                ctypes.cast(seg.data_ptr(), ctypes.POINTER(ctypes.c_int32)), B, plan.rows,
                layers_arr, len(repr_set), outs_arr, pflags, N.ptr(logits), N.ptr(ws), ws.numel(),
                N.cur_stream()))
        out_dt = w.dtype
        cast = lambda t: t if t.dtype == out_dt else t.to(out_dt)
        if not unpack:
            return {"logits": cast(logits), "representations": {l: cast(r) for l, r in zip(repr_set, reps)},
                    "segments": seg}
        un = lambda t: plan.unpack(t, idx, keep)
        return {"logits": cast(un(logits)), "representations": {l: cast(un(r)) for l, r in zip(repr_set, reps)}}

    def profile_begin(self):
        """Arm per-kernel-class HIP-event timing of the following forward calls (bench.py)."""
        from . import _native as N

        if self._engine is None:
            raise RuntimeError("run one forward before profiling")
        N.check(N.lib.esmk_profile_begin(self._engine.handle))
        self._engine.profiling = True  # per-class events live on ONE stream: no dual-stream forward while armed

    def profile_end(self):
        """Stop profiling; returns [{name, launches, ms, flops, bytes}] summed over the calls."""
        from . import _native as N

        buf = (N.EsmkProfileEntry * 32)()
        n = ctypes.c_int()
        N.check(N.lib.esmk_profile_end(self._engine.handle, buf, 32, ctypes.byref(n)))
        self._engine.profiling = False
        return [dict(name=buf[i].name.decode(), launches=buf[i].launches, ms=buf[i].ms, flops=buf[i].flops,
                     bytes=buf[i].bytes) for i in range(n.value)]

    def ln_fold_active(self):
        """True / False: the engine of this model runs with / without the LayerNorm fold; None before the
        first forward."""
        if self._engine is None:
            return None
        return bool(self._engine.N.lib.esmk_ln_fold_enabled(self._engine.handle) == 1)

    def refresh_engine(self):
        """Drop the engine state (call after replacing Parameter objects or sub-modules)."""
        if self._engine is not None:
            self._engine.close()
        object.__setattr__(self, "_engine", None)
        object.__setattr__(self, "_fold_hazard", None)

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_engine"] = None  # the native handle is rebuilt lazily
        return state

    def predict_contacts(self, tokens):
        """Reference esm2.py:146-147 returns ``self(tokens, return_contacts=True)["contacts"]``, which first builds
        the [B,L,H,T,T] attention tensor (2.8 GB per 1024-token sequence at 650M).  Only the map leaves this call, so
        the engine accumulates it layer by layer instead (csrc/contacts.hip) — same formula, no attention tensor."""
        return self(tokens, return_contacts=True, contacts_only=True)["contacts"]
