"""Token-packed batches (esmk_forward_packed / ESM2.forward_varlen) on the MI355X.

The packed forward must give every sequence exactly what the padded engine gives it alone (same kernels, same
tile order inside the segment => bit equality), and therefore the oracle's values within the usual tolerance."""
import ctypes

import pytest
import torch

import esm
from esm_amd.packing import pack_plan
from esm_amd.synth import synth_esm2_state_dict
from oracle.esm2_oracle import esm2_forward

import _contract as C

pytestmark = pytest.mark.gpu
PAD, MASK, CLS, EOS = 1, 32, 0, 2


def rel_err(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def build(L, E, H, seed):
    sd = synth_esm2_state_dict(L, E, H, seed=seed)
    m = esm.ESM2(L, E, H).eval()
    m.load_state_dict(sd)
    return m.cuda(), sd


def ragged_batch(lengths, seed, masks=(), interior_pad=()):
    """Right-padded [B, max len] batch as BatchConverter yields it: <cls> residues <eos> <pad>..."""
    g = torch.Generator().manual_seed(seed)
    T = max(lengths)
    toks = torch.full((len(lengths), T), PAD, dtype=torch.int64)
    for b, n in enumerate(lengths):
        toks[b, 0] = CLS
        if n > 2:
            toks[b, 1:n - 1] = torch.randint(4, 24, (n - 2,), generator=g)
        toks[b, n - 1] = EOS
    for b, t in masks:
        toks[b, t] = MASK
    for b, t in interior_pad:
        toks[b, t] = PAD
    return toks


LENGTHS = [2, 3, 17, 64, 65, 127, 128, 129, 300, 16, 33, 257]


@pytest.mark.parametrize("dims", [(2, 128, 2), (2, 320, 20), (2, 1280, 20)], ids=["d64_small", "d16_8M", "d64_650M"])
def test_packed_equals_each_sequence_alone(dims):
    L, E, H = dims
    model, _ = build(L, E, H, seed=11)
    toks = ragged_batch(LENGTHS, seed=3, masks=[(8, 5), (8, 40), (3, 7)], interior_pad=[(8, 100)])
    with torch.no_grad():
        model.forward_varlen(toks, repr_layers=[L], min_saving=None)
        # every byte of the workspace = 0xFF (NaN in fp16 / fp32): nothing may be read before it is written,
        # and gap rows must not leak into their neighbours
        model._engine.workspace.fill_(255)
        out = model.forward_varlen(toks, repr_layers=[0, 1, L], min_saving=None)
        for b, n in enumerate(LENGTHS):
            one = model(toks[b:b + 1, :n].cuda(), repr_layers=[0, 1, L])
            for layer in (0, 1, L):
                assert torch.equal(out["representations"][layer][b, :n], one["representations"][layer][0]), (b, n, layer)
            assert torch.equal(out["logits"][b, :n], one["logits"][0]), (b, n)
            assert out["logits"][b, n:].abs().max().item() == 0 if n < toks.shape[1] else True


def test_packed_against_oracle_and_padded_engine():
    L, E, H = 3, 256, 4
    model, sd = build(L, E, H, seed=5)
    lengths = [50, 2, 200, 131, 64, 7]
    toks = ragged_batch(lengths, seed=9, masks=[(0, 3), (0, 4), (2, 150)], interior_pad=[(2, 20)])
    ref = esm2_forward(sd, toks, L, H, repr_layers=[L])
    with torch.no_grad():
        pk = model.forward_varlen(toks, repr_layers=[L], min_saving=None)
        pd = model(toks.cuda(), repr_layers=[L])
    nonpad = toks.ne(PAD)
    floor = C.floor_forward(sd, toks, L, H, model=model, repr_layers=[L])  # the parity contract (tests/_contract.py), toy model
    C.check_tensors("packed repr", pk["representations"][L].cpu(), ref["representations"][L], floor["representations"][L], nonpad)
    C.check_tensors("packed logits", pk["logits"].cpu(), ref["logits"], floor["logits"], nonpad)
    # the padded engine computes the same rows (plus the pad rows)
    m = nonpad.cuda()
    assert torch.equal(pk["representations"][L][m], pd["representations"][L][m])
    assert torch.equal(pk["logits"][m], pd["logits"][m])


def test_packed_layout_and_fallback():
    model, _ = build(1, 128, 2, seed=2)
    toks = ragged_batch([40, 40, 39], seed=1)
    with torch.no_grad():
        a = model.forward_varlen(toks, repr_layers=[1])                  # nothing to save: padded path
        b = model.forward_varlen(toks, repr_layers=[1], min_saving=None)  # forced packing
        raw = model.forward_varlen(toks, repr_layers=[1], unpack=False)
    nonpad = toks.ne(PAD).cuda()
    assert torch.equal(a["representations"][1][nonpad], b["representations"][1][nonpad])
    plan = pack_plan(toks, PAD)
    assert raw["segments"].tolist() == plan.segments.tolist() == [[0, 40], [48, 40], [96, 39]]
    assert raw["representations"][1].shape == (plan.rows, 128) and plan.rows % 128 == 0
    assert torch.equal(raw["representations"][1][48:88], b["representations"][1][1, :40])
    # device tokens and explicit lengths are accepted as well
    with torch.no_grad():
        c = model.forward_varlen(toks.cuda(), repr_layers=[1], lengths=[40, 40, 39], min_saving=None)
    assert torch.equal(c["logits"], b["logits"])


def test_packed_errors_are_loud():
    from esm_amd import _native as N
    model, _ = build(1, 128, 2, seed=2)
    toks = ragged_batch([20, 5], seed=1)
    model.forward_varlen(toks, min_saving=None)
    eng = model._engine
    need = ctypes.c_size_t()
    lib = N.lib
    assert lib.esmk_packed_workspace_bytes(eng.handle, 2, 100, N.OUT_LOGITS, ctypes.byref(need)) != 0  # rows % 64
    assert lib.esmk_packed_workspace_bytes(eng.handle, 2, 128, N.OUT_ATTN, ctypes.byref(need)) != 0
    assert b"ESMK_OUT_LOGITS" in lib.esmk_last_error()
    N.check(lib.esmk_packed_workspace_bytes(eng.handle, 2, 128, N.OUT_LOGITS, ctypes.byref(need)))
    ws = torch.empty(need.value, dtype=torch.uint8, device="cuda")
    flat = torch.full((128,), PAD, dtype=torch.int64, device="cuda")
    logits = torch.empty((128, 33), device="cuda")

    def call(segs, flags=N.OUT_LOGITS):
        arr = (ctypes.c_int32 * len(segs))(*segs)
        return lib.esmk_forward_packed(eng.handle, N.ptr(eng.packed), N.ptr(flat), arr, len(segs) // 2, 128,
                                       None, 0, None, flags, N.ptr(logits), N.ptr(ws), ws.numel(), N.cur_stream())

    assert call([0, 20, 24, 5]) != 0 and b"multiples of 16" in lib.esmk_last_error()
    assert call([0, 20, 16, 5]) != 0 and b"disjoint" in lib.esmk_last_error()
    assert call([16, 20, 48, 5]) != 0
    assert call([0, 20, 112, 17]) != 0 and b"past the last row" in lib.esmk_last_error()
    assert call([0, 20, 32, 0]) != 0 and b"empty segment" in lib.esmk_last_error()
    assert call([0, 20, 32, 5], N.OUT_LOGITS | N.OUT_CONTACTS) != 0
    assert call([0, 20, 32, 5]) == 0
    torch.cuda.synchronize()


def test_packed_head_dim_128():
    """esm2_t48_15B geometry (head_dim 128): attention128.hip takes the same segment work list."""
    L, E, H = 2, 256, 2
    model, _ = build(L, E, H, seed=4)
    lengths = [2, 70, 129, 33, 200]
    toks = ragged_batch(lengths, seed=6, masks=[(1, 9)], interior_pad=[(4, 50)])
    with torch.no_grad():
        out = model.forward_varlen(toks, repr_layers=[L], min_saving=None)
        for b, n in enumerate(lengths):
            one = model(toks[b:b + 1, :n].cuda(), repr_layers=[L])
            assert torch.equal(out["representations"][L][b, :n], one["representations"][L][0]), (b, n)
            assert torch.equal(out["logits"][b, :n], one["logits"][0]), (b, n)


@pytest.mark.parametrize("ln_before", [True, False], ids=["esm1b", "esm1v_style"])
def test_packed_esm1b(ln_before):
    """Learned positions restart at every segment (esm/modules.py:240-257), padded rows are zeroed after the
    embedding LayerNorm (esm/model/esm1.py:133-139)."""
    import argparse

    from esm_amd.synth import synth_esm1b_state_dict

    L, E, H = 2, 128, 2
    args = argparse.Namespace(arch="roberta_large", layers=L, embed_dim=E, ffn_embed_dim=4 * E, attention_heads=H,
                              max_positions=1024, token_dropout=True, emb_layer_norm_before=ln_before)
    model = esm.ProteinBertModel(args, esm.Alphabet.from_architecture("roberta_large")).eval()
    model.load_state_dict(synth_esm1b_state_dict(L, E, H, seed=1, ln_before=ln_before), strict=True)
    model = model.cuda()
    lengths = [40, 2, 150, 65, 300]
    toks = ragged_batch(lengths, seed=8, masks=[(0, 3), (4, 77)], interior_pad=[(2, 30)])
    with torch.no_grad():
        out = model.forward_varlen(toks, repr_layers=[0, L], min_saving=None)
        for b, n in enumerate(lengths):
            one = model(toks[b:b + 1, :n].cuda(), repr_layers=[0, L])
            for layer in (0, L):
                assert torch.equal(out["representations"][layer][b, :n], one["representations"][layer][0]), (b, n, layer)
            assert torch.equal(out["logits"][b, :n], one["logits"][0]), (b, n)


def test_random_lengths_padded_packed_alone_bit_equal():
    """Random batches of random lengths (tools/fuzz_attention_lengths.py, 650M width): the padded forward, the
    token-packed forward and each sequence alone agree bit for bit on every real row.  The lazy softmax takes its
    exact / fast decision per wave, so this also pins the rule that padded query rows mirror the last real row of
    their wave (attention.hip) — a first version without it differed by 6e-4 on a (963, 713) batch."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_attention_lengths.py")
    spec = importlib.util.spec_from_file_location("fuzz_attention_lengths", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main(["--iters", "16", "--seed", "3"])
