"""`.half()` / `.bfloat16()` models: the engine writes representations and attention maps directly in the model
dtype (ESMK_OUT_REPR_LOWP / ESMK_OUT_ATTN_LOWP; SURVEY.md §8 f-2, the ESMFold language-model front end of reference
esm/esmfold/v1/esmfold.py:61-67,118-135).  The bits must be those of the fp32 outputs rounded once (what
``ESM_AMD_NATIVE_LOWP=0`` — fp32 outputs + a torch cast — produces), on every path that hands out representations:
intermediate layers, the final LayerNorm with and without logits, the padded-stride width E = 480, token-packed
batches, duplicates."""
import pytest
import torch

import esm
from esm_amd.synth import synth_esm2_state_dict, synth_tokens

pytestmark = pytest.mark.gpu


def build(L, E, H, dtype):
    m = esm.ESM2(L, E, H).eval()
    m.load_state_dict(synth_esm2_state_dict(L, E, H, seed=21))
    return m.cuda().to(dtype)


def ragged(B, Lmax, lengths, seed=3):
    toks = synth_tokens(B, Lmax, seed=seed)
    for b, n in enumerate(lengths):
        toks[b, n + 1] = 2
        toks[b, n + 2:] = 1
    return toks


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("L,E,H", [(3, 128, 2), (2, 480, 20), (2, 256, 2)], ids=["d64", "E480_padded_stride", "d128"])
def test_native_lowp_equals_cast_of_fp32(monkeypatch, dtype, L, E, H):
    model = build(L, E, H, dtype)
    toks = ragged(3, 70, [70, 41, 9]).cuda()
    layers = list(range(L + 1))
    with torch.no_grad():
        monkeypatch.setenv("ESM_AMD_NATIVE_LOWP", "0")
        ref = model(toks, repr_layers=layers, need_head_weights=True)
        refc = model(toks, repr_layers=[L], return_contacts=True)
        monkeypatch.setenv("ESM_AMD_NATIVE_LOWP", "1")
        got = model(toks, repr_layers=layers, need_head_weights=True)
        gotc = model(toks, repr_layers=[L], return_contacts=True)  # attention maps stay fp32 inside, cast outside
        pk = model.forward_varlen(toks, repr_layers=layers, min_saving=None)
    for l in layers:
        assert got["representations"][l].dtype == dtype
        assert torch.equal(got["representations"][l], ref["representations"][l]), l
    assert got["attentions"].dtype == dtype and torch.equal(got["attentions"], ref["attentions"])
    assert torch.equal(got["logits"], ref["logits"])
    assert torch.equal(gotc["contacts"], refc["contacts"]) and torch.equal(gotc["attentions"], refc["attentions"])
    assert torch.equal(gotc["representations"][L], ref["representations"][L])
    nonpad = toks.ne(1)
    for l in layers:  # packed rows carry the same bits as the padded forward (tests/test_varlen_gpu.py) -> same rounding
        assert pk["representations"][l].dtype == dtype
        assert torch.equal(pk["representations"][l][nonpad], ref["representations"][l][nonpad]), l


def test_last_layer_without_logits_and_duplicates():
    """C ABI: representation L alone (no logits: the final LayerNorm writes the output buffer itself) and a
    layer requested twice."""
    import ctypes

    from esm_amd import _native as N

    L, E, H = 2, 128, 2
    model = build(L, E, H, torch.float16)
    toks = ragged(2, 40, [40, 17]).cuda()
    with torch.no_grad():
        ref = model(toks, repr_layers=[1, L])
    eng = model._engine
    B, T = toks.shape
    flags = N.OUT_REPR_LOWP
    outs = [torch.empty((B, T, E), dtype=torch.float16, device="cuda") for _ in range(3)]
    layers_arr = (ctypes.c_int32 * 3)(L, 1, L)
    outs_arr = (ctypes.c_void_p * 3)(*[o.data_ptr() for o in outs])
    ws = eng.workspace_for(B, T, flags)
    N.check(N.lib.esmk_forward(eng.handle, N.ptr(eng.packed), N.ptr(toks), B, T, layers_arr, 3, outs_arr, flags,
                               None, None, None, N.ptr(ws), ws.numel(), N.cur_stream()))
    torch.cuda.synchronize()
    assert torch.equal(outs[0], ref["representations"][L]) and torch.equal(outs[2], outs[0])
    assert torch.equal(outs[1], ref["representations"][1])
    # the materialised contact path needs fp32 maps: the combination is refused, not silently mis-read
    attn = torch.empty((B, L, H, T, T), dtype=torch.float16, device="cuda")
    ct = torch.empty((B, T - 2, T - 2), device="cuda")
    bad = N.OUT_ATTN | N.OUT_CONTACTS | N.OUT_ATTN_LOWP
    ws = eng.workspace_for(B, T, bad)
    rc = N.lib.esmk_forward(eng.handle, N.ptr(eng.packed), N.ptr(toks), B, T, layers_arr, 0, outs_arr, bad,
                            None, N.ptr(attn), N.ptr(ct), N.ptr(ws), ws.numel(), N.cur_stream())
    assert rc != 0 and b"ESMK_OUT_ATTN_LOWP" in N.lib.esmk_last_error()
