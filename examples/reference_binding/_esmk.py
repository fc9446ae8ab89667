"""The binding a maintainer of facebookresearch/esm would drop next to esm/model/esm2.py:
ctypes over the C ABI of libesmk.so (include/esmk.h), no dependency on this repo's Python package.

    from esm.model import _esmk                     # in the reference tree
    class ESM2(nn.Module):
        def forward(self, tokens, repr_layers=[], need_head_weights=False, return_contacts=False):
            if tokens.is_cuda and not torch.is_grad_enabled():
                return _esmk.forward(self, tokens, repr_layers, need_head_weights, return_contacts)
            ...                                      # the existing PyTorch path (esm2.py:77-144)

``forward`` takes any nn.Module with the reference ESM2's attributes and state-dict keys.  The engine copy of the
weights is re-packed whenever a parameter's storage, version or dtype changed (.cuda(), .half(), load_state_dict).
"""
import ctypes
import os

import torch

_LIB = None
OUT_LOGITS, OUT_ATTN, OUT_CONTACTS = 1, 2, 4  # esmk.h: ESMK_OUT_*
_DTYPE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}  # esmk.h: ESMK_F32 / F16 / BF16


class Config(ctypes.Structure):  # struct esmk_config
    _fields_ = [(n, ctypes.c_int32) for n in (
        "num_layers", "embed_dim", "num_heads", "ffn_dim", "vocab", "pad_idx", "mask_idx", "cls_idx", "eos_idx",
        "token_dropout", "prepend_bos", "append_eos", "operand_dtype", "no_rope", "num_positions", "ln_before",
        "weight_split", "ln_fold")]


def lib(path=None):
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(path or os.environ.get("LIBESMK", "libesmk.so"))
        _LIB.esmk_last_error.restype = ctypes.c_char_p
    return _LIB


def _chk(rc):
    if rc:
        raise RuntimeError(lib().esmk_last_error().decode())


def config_for(m, operand_dtype=torch.float16, weight_split=0, ln_fold=0):
    """esmk_config from the attributes ESM2.__init__ sets (esm/model/esm2.py:24-38).  weight_split = 1: the engine's
    f16x2 precision mode (split weights, 2x GEMM time, ~40 % lower error; esmk.h).  ln_fold: 0 = the library's default,
    1 / -1 = LayerNorm fold on / off (esmk.h)."""
    return Config(m.num_layers, m.embed_dim, m.attention_heads, 4 * m.embed_dim, m.alphabet_size, m.padding_idx,
                  m.mask_idx, m.cls_idx, m.eos_idx, int(bool(m.token_dropout)), int(bool(m.prepend_bos)),
                  int(bool(m.append_eos)), _DTYPE[operand_dtype], 0, 0, 0, int(weight_split), int(ln_fold))


def gain_hazard(m):
    """The check INTEGRATION.md asks of a binding before esmk_create: LayerNorm gains that silence very large channels make
    the fold's un-normalised fp16 operand rows lossy -> run such checkpoints with ln_fold = -1."""
    g = torch.stack([ln.weight.detach().float().abs() for layer in m.layers for ln in (layer.self_attn_layer_norm, layer.final_layer_norm)])
    med = g.median(dim=-1, keepdim=True).values
    return float((torch.where(g < med / 8, med / g.clamp_min(1e-30), torch.zeros_like(g)).sum(-1) / g.shape[-1]).mean())


class Engine:
    """One esmk_model handle + packed parameter image + workspace for a model on one device."""

    def __init__(self, m, device, operand_dtype=torch.float16):
        L = lib()
        self.device, self.h = device, ctypes.c_void_p()
        cfg = config_for(m, operand_dtype, ln_fold=-1 if gain_hazard(m) > 0.5 else 0)
        with torch.cuda.device(device):
            _chk(L.esmk_create(ctypes.byref(cfg), ctypes.byref(self.h)))
            d = m.embed_dim // m.attention_heads
            inv = (1.0 / (10000 ** (torch.arange(0, d, 2).float() / d))).tolist()  # rotary_embedding.py:40-41
            _chk(L.esmk_set_rope_inv_freq(self.h, (ctypes.c_float * len(inv))(*inv), len(inv)))
            n = ctypes.c_size_t()
            _chk(L.esmk_packed_bytes(self.h, ctypes.byref(n)))
            self.packed = torch.zeros(n.value, dtype=torch.uint8, device=device)  # padded slots must stay zero
        self.fingerprint, self.workspace = None, None

    def __del__(self):
        if getattr(self, "h", None):
            lib().esmk_destroy(self.h)
            self.h = None

    def sync_weights(self, m, stream):
        named = [(k, t) for k, t in m.state_dict(keep_vars=True).items() if not k.endswith("inv_freq")]
        fp = tuple((t.data_ptr(), t._version, t.dtype) for _, t in named)
        if fp == self.fingerprint:
            return
        # LayerNorm parameters first: with the engine's LayerNorm fold (esmk.h) q/k/v and fc1 weights are folded with them
        for k, t in sorted(named, key=lambda kt: 0 if "layer_norm" in kt[0] else 1):
            t = t.detach().contiguous()
            shape = (ctypes.c_int64 * t.dim())(*t.shape)
            _chk(lib().esmk_pack_weight(self.h, ctypes.c_void_p(self.packed.data_ptr()),
                                        ctypes.c_size_t(self.packed.numel()), k.encode(),
                                        ctypes.c_void_p(t.data_ptr()), _DTYPE[t.dtype], shape, t.dim(), stream))
        self.fingerprint = fp


def forward(m, tokens, repr_layers=(), need_head_weights=False, return_contacts=False):
    """Drop-in body of ESM2.forward for HIP tensors in inference (esm/model/esm2.py:77-144)."""
    if return_contacts:
        need_head_weights = True
    assert tokens.ndim == 2
    L_ = lib()
    dev = tokens.device
    eng = m.__dict__.get("_esmk_engine")
    if eng is None or eng.device != dev:
        eng = Engine(m, dev)
        m.__dict__["_esmk_engine"] = eng
    B, T = tokens.shape
    nl, E, H, V = m.num_layers, m.embed_dim, m.attention_heads, m.alphabet_size
    layers = sorted({int(i) for i in repr_layers if 0 <= int(i) <= nl})
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        eng.sync_weights(m, stream)
        f32 = dict(dtype=torch.float32, device=dev)
        tok = tokens.to(torch.int64).contiguous()
        flags = OUT_LOGITS
        logits = torch.empty((B, T, V), **f32)
        reps = [torch.empty((B, T, E), **f32) for _ in layers]
        attn = contacts = None
        if need_head_weights:
            flags |= OUT_ATTN
            attn = torch.empty((B, nl, H, T, T), **f32)
        if return_contacts:
            S = max(T - int(m.prepend_bos) - int(m.append_eos), 0)
            contacts = torch.empty((B, S, S), **f32)
            if S > 0:
                flags |= OUT_CONTACTS
        need = ctypes.c_size_t()
        _chk(L_.esmk_workspace_bytes(eng.h, B, T, ctypes.c_uint32(flags), ctypes.byref(need)))
        if eng.workspace is None or eng.workspace.numel() < need.value:
            eng.workspace = torch.empty(need.value, dtype=torch.uint8, device=dev)
        ptr = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
        larr = (ctypes.c_int32 * max(1, len(layers)))(*layers)
        oarr = (ctypes.c_void_p * max(1, len(layers)))(*[r.data_ptr() for r in reps])
        _chk(L_.esmk_forward(eng.h, ptr(eng.packed), ptr(tok), B, T, larr, len(layers), oarr, ctypes.c_uint32(flags),
                             ptr(logits), ptr(attn), ptr(contacts), ptr(eng.workspace),
                             ctypes.c_size_t(eng.workspace.numel()), stream))
    out_dt = next(m.parameters()).dtype
    cast = lambda t: t if t.dtype == out_dt else t.to(out_dt)
    result = {"logits": cast(logits), "representations": {l: cast(r) for l, r in zip(layers, reps)}}
    if need_head_weights:
        result["attentions"] = cast(attn)
        if return_contacts:
            result["contacts"] = cast(contacts)
    return result
