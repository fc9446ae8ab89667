"""Thin torch-tensor wrappers over the single-kernel entry points of libesmk.so.

Used by the parity tests and micro-benchmarks; the model path (esm_amd.esm2.ESM2) calls
``esmk_forward`` directly.  Every function requires CUDA (HIP) tensors and launches on the
current stream; nothing here falls back to torch math.
"""
import ctypes

import torch

from . import _native as N


def _req_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("esm_amd.ops: tensors must live on the GPU (no CPU fallback)")
        if t is not None and not t.is_contiguous():
            raise RuntimeError("esm_amd.ops: tensors must be contiguous")


def layernorm(x, gamma, beta, operand_dtype=torch.float16, want_op=True, want_f32=False, variant=None):
    """torch.nn.LayerNorm(E, eps=1e-5) on fp32 rows (reference esm/modules.py:68-81)."""
    _req_cuda(x, gamma, beta)
    assert x.dtype == torch.float32 and gamma.dtype == torch.float32 and beta.dtype == torch.float32
    rows, E = x.reshape(-1, x.shape[-1]).shape
    y = torch.empty_like(x, dtype=operand_dtype) if want_op else None
    y32 = torch.empty_like(x) if want_f32 else None
    code = N.dtype_code(operand_dtype) | (0 if variant is None else (variant + 1) << 8)
    N.check(N.lib.esmk_op_layernorm(N.ptr(x), N.ptr(gamma), N.ptr(beta), N.ptr(y), N.ptr(y32), rows, E,
                                    code, N.cur_stream()))
    return y, y32


def masked_row_mean(x, counts, first_row=1):
    """out[b] = x[b, first_row : first_row + counts[b]].mean(0) in fp32 (reference scripts/extract.py:113-116);
    x [B,T,E] fp32 / fp16 / bf16, counts int32 [B] on the same device.  Empty slices give NaN, as torch.mean."""
    _req_cuda(x, counts)
    assert x.dim() == 3 and counts.dtype == torch.int32 and counts.numel() == x.shape[0]
    B, T, E = x.shape
    out = torch.empty((B, E), dtype=torch.float32, device=x.device)
    N.check(N.lib.esmk_op_masked_row_mean(N.ptr(x), N.dtype_code(x.dtype), N.ptr(counts), N.ptr(out), B, T, E,
                                          first_row, N.cur_stream()))
    return out


def linear(a, w, bias=None, epilogue=N.EPI_STORE_T, out=None, force_generic=False, dbg=0, force_old=False,
           panel_c=0, half_m=0):
    """nn.Linear with fused epilogue: a [M,K], w [N,K] (both f16 or bf16), bias fp32 [N]."""
    _req_cuda(a, w, bias, out)
    assert a.dtype == w.dtype and a.dtype in (torch.float16, torch.bfloat16)
    M, K = a.shape
    Nn, K2 = w.shape
    assert K == K2
    if epilogue == N.EPI_RESID_F32:
        assert out is not None and out.dtype == torch.float32 and tuple(out.shape) == (M, Nn)
    elif out is None:
        odt = a.dtype if epilogue in (N.EPI_STORE_T, N.EPI_GELU_T) else torch.float32
        out = torch.empty((M, Nn), dtype=odt, device=a.device)
    # 0x100: generic 64x64 kernel, 0x200: one-tile-per-workgroup 256x256 kernel (gemm.hip) instead of the
    # persistent kernel (gemm8.hip); bits 20..27: tile-order panel width of the persistent kernel
    code = (N.dtype_code(a.dtype) | (0x100 if force_generic else 0) | (0x200 if force_old else 0) | (dbg << 12)
            | (panel_c << 20) | ({0: 0, 1: 1, -1: 2}[half_m] << 28))  # half_m: 1 force 128-row tiles, -1 never
    N.check(N.lib.esmk_op_linear(N.ptr(a), N.ptr(w), N.ptr(bias), N.ptr(out), M, Nn, K, epilogue, code,
                                 N.cur_stream()))
    return out


def split_weight(w):
    """w [N,K] (fp32 / fp16 / bf16) -> fp16 [N,2K]: 64-column K tiles interleaved hi | lo, hi = fp16(w), lo = fp16(w - hi)
    (the weight image of the f16x2 precision mode)."""
    _req_cuda(w)
    Nn, K = w.shape
    out = torch.zeros((Nn, 2 * K), dtype=torch.float16, device=w.device)
    N.check(N.lib.esmk_op_split_weight(N.ptr(w.contiguous()), N.dtype_code(w.dtype), N.ptr(out), Nn, K, N.cur_stream()))
    return out


def linear_split(a, w2, bias=None, epilogue=N.EPI_STORE_T, out=None):
    """a [M,K] fp16, w2 = split_weight(w) [N,2K]: a . (w_hi + w_lo)^T + bias with the epilogues of ``linear``."""
    _req_cuda(a, w2, bias, out)
    assert a.dtype == torch.float16 and w2.dtype == torch.float16
    M, K = a.shape
    Nn = w2.shape[0]
    assert w2.shape[1] == 2 * K
    if epilogue == N.EPI_RESID_F32:
        assert out is not None and out.dtype == torch.float32 and tuple(out.shape) == (M, Nn)
    elif out is None:
        out = torch.empty((M, Nn), dtype=torch.float16 if epilogue in (N.EPI_STORE_T, N.EPI_GELU_T) else torch.float32,
                          device=a.device)
    N.check(N.lib.esmk_op_linear_split(N.ptr(a), N.ptr(w2), N.ptr(bias), N.ptr(out), M, Nn, K, epilogue, N.cur_stream()))
    return out


LOG2E = 1.4426950408889634


def to_log2_domain(q, dtype=None):
    """The attention kernels take q with log2(e) folded in (the engine's QKV epilogue does that in fp32 before the
    single rounding to the operand dtype).  Returns (q_kernel, q_effective): the operand-dtype tensor to hand to the
    kernels and the fp32 natural-domain q it represents exactly (q_kernel / log2 e) — what a reference must use."""
    qk = (q.float() * LOG2E).to(dtype or q.dtype)
    return qk, qk.float() / LOG2E


def attention(q, k, vt, key_bias=None, want_lse=False):
    """softmax(q k^T + key_bias) v for head_dim 64.  q [B,H,T,64] in the LOG2 domain (``to_log2_domain``), k
    [B,H,T,64]; vt [B,H,64,Tp] (layout of the fused QKV epilogue, see csrc/attention.hip).  Returns ctx
    [B*T, H*64] (+ the row log-sum-exp converted to the natural log, [B,H,T])."""
    _req_cuda(q, k, vt, key_bias)
    B, H, T, D = q.shape
    assert D == 64 and vt.shape[-1] == (T + 63) // 64 * 64
    ctx = torch.empty((B * T, H * 64), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, H, T), dtype=torch.float32, device=q.device) if want_lse else None
    N.check(N.lib.esmk_op_attention(N.ptr(q), N.ptr(k), N.ptr(vt), N.ptr(key_bias), N.ptr(ctx), N.ptr(lse),
                                    B, H, T, N.dtype_code(q.dtype), N.cur_stream()))
    return (ctx, lse / LOG2E) if want_lse else ctx  # the kernel's lse is log2-domain


def attention_probs(q, k, lse, key_bias=None, out=None, layer=0, num_layers=1):
    """q in the log2 domain; lse: natural-log row log-sum-exp as returned by ``attention(..., want_lse=True)``."""
    _req_cuda(q, k, lse, key_bias, out)
    lse = (lse * LOG2E).contiguous()
    B, H, T, D = q.shape
    if out is None:
        out = torch.empty((B, num_layers, H, T, T), dtype=torch.float32, device=q.device)
    N.check(N.lib.esmk_op_attention_probs(N.ptr(q), N.ptr(k), N.ptr(lse), N.ptr(key_bias), N.ptr(out), B, H, T,
                                          layer, num_layers, N.dtype_code(q.dtype), N.cur_stream()))
    return out


def contacts(attn, tokens, w, b, eos_idx=2, prepend_bos=True, append_eos=True):
    """ContactPredictionHead.forward (reference esm/modules.py:338-357)."""
    _req_cuda(attn, tokens, w, b)
    B, L, H, T, _ = attn.shape
    C = L * H
    S = T - int(prepend_bos) - int(append_eos)
    scratch = torch.empty((B * C * (S + 1),), dtype=torch.float32, device=attn.device)
    out = torch.empty((B, S, S), dtype=torch.float32, device=attn.device)
    N.check(N.lib.esmk_op_contacts(N.ptr(attn), N.ptr(tokens), N.ptr(w.reshape(-1)), N.ptr(b.reshape(-1)),
                                   N.ptr(scratch), N.ptr(out), B, C, T, eos_idx, int(prepend_bos),
                                   int(append_eos), N.cur_stream()))
    return out


def permute_keys16(t):
    """Key position used by the V^T layout: inside each group of 16 keys the 4-groups 1 and 2
    are swapped (position p holds key perm[p])."""
    idx = torch.arange(t)
    t16 = idx & 15
    pos = (idx & ~15) | (((t16 >> 2) & 1) << 3) | (((t16 >> 3) & 1) << 2) | (t16 & 3)
    return pos


def make_vt(v):
    """Reference layout helper: v [B,H,T,64] -> vt [B,H,64,Tp] as the QKV epilogue writes it."""
    B, H, T, D = v.shape
    Tp = (T + 63) // 64 * 64
    vt = torch.zeros((B, H, D, Tp), dtype=v.dtype, device=v.device)
    pos = permute_keys16(T).to(v.device)
    vt[:, :, :, pos] = v.transpose(2, 3)
    return vt


class QkvHandle:
    """Owns an esmk_model handle for the fused QKV + RoPE op (tests / micro-benchmarks)."""

    def __init__(self, embed_dim, num_heads, operand_dtype=torch.float16):
        cfg = N.EsmkConfig(1, embed_dim, num_heads, 4 * embed_dim, 33, 1, 32, 0, 2, 1, 1, 1,
                           N.dtype_code(operand_dtype))
        self.h = ctypes.c_void_p()
        N.check(N.lib.esmk_create(ctypes.byref(cfg), ctypes.byref(self.h)))
        d = embed_dim // num_heads
        inv = 1.0 / (10000 ** (torch.arange(0, d, 2).float() / d))
        self.inv_freq = inv
        arr = (ctypes.c_float * inv.numel())(*inv.tolist())
        N.check(N.lib.esmk_set_rope_inv_freq(self.h, arr, inv.numel()))
        self.E, self.H, self.dtype = embed_dim, num_heads, operand_dtype

    def __call__(self, a, wqkv, bias, B, T, log2_domain=False):
        """log2_domain: q also carries log2(e) — the q ``attention`` / ``attention_probs`` take (esmk_op_qkv_rope2)."""
        _req_cuda(a, wqkv, bias)
        H, dev = self.H, a.device
        Tp = (T + 63) // 64 * 64
        q = torch.empty((B, H, T, 64), dtype=self.dtype, device=dev)
        k = torch.empty_like(q)
        vt = torch.empty((B, H, 64, Tp), dtype=self.dtype, device=dev)
        N.check(N.lib.esmk_op_qkv_rope2(self.h, N.ptr(a), N.ptr(wqkv), N.ptr(bias), N.ptr(q), N.ptr(k), N.ptr(vt),
                                        B, T, int(bool(log2_domain)), N.cur_stream()))
        return q, k, vt

    def __del__(self):
        if getattr(self, "h", None):
            N.lib.esmk_destroy(self.h)
            self.h = None
