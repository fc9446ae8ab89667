"""q, k and v projections as ONE launch (gemm9 EPI_QKV_ALL, kernels.h) against the two launches (EPI_QKV_ROPE + EPI_V_T):
the same tiles computed by the same instruction sequences — every output must be bit-equal, for aligned and ragged
sequence lengths, both tile heights, padded and token-packed batches.  Reference: the three projections + rotary + head
split of esm/multihead_attention.py:256-284,354-355."""
import ctypes
import math

import pytest
import torch

import esm
from esm_amd import _native as N
from esm_amd import ops
from esm_amd.synth import skip_param_init, synth_esm2_state_dict

pytestmark = pytest.mark.gpu


def knob(v):
    N.check(N.lib.esmk_debug_set(b"qkv_one_launch", ctypes.c_double(v)))


@pytest.fixture(autouse=True)
def _restore_knob():
    yield
    knob(-1)


@pytest.mark.parametrize("B,T,E,H", [(4, 1022, 1280, 20), (1, 1022, 1280, 20), (3, 160, 1280, 20), (2, 763, 1280, 20),
                                     (8, 256, 1280, 20), (16, 1022, 1280, 20), (5, 333, 640, 10), (2, 64, 2560, 40)])
def test_one_launch_equals_two_launches_bit_for_bit(B, T, E, H):
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + T)
    a = (torch.randn(B * T, E, device="cuda", generator=g)).half()
    w = (torch.randn(3 * E, E, device="cuda", generator=g) / math.sqrt(E)).half()
    bias = 0.1 * torch.randn(3 * E, device="cuda", generator=g)
    qkv = ops.QkvHandle(E, H)
    Tp = (T + 63) // 64 * 64
    outs = []
    for mode in (0, 1):
        knob(mode)
        q, k, vt = qkv(a, w, bias, B, T, log2_domain=True)
        outs.append((q.clone(), k.clone(), vt[..., :T].clone() if Tp != T else vt.clone()))
    for name, x, y in zip("q k vt".split(), outs[0], outs[1]):
        assert torch.isfinite(y.float()).all(), name
        assert torch.equal(x, y), (name, int((x != y).sum()))


def test_one_launch_is_taken_and_is_refused_where_it_does_not_apply():
    """The knob reaches the kernel choice: forced on, a shape the combined form does not support (E % 128 != 0) still
    runs (two launches); the C ABI rejects unknown knobs."""
    E, H, B, T = 320, 5, 2, 100
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn(B * T, E, device="cuda", generator=g).half()
    w = (torch.randn(3 * E, E, device="cuda", generator=g) / math.sqrt(E)).half()
    bias = 0.1 * torch.randn(3 * E, device="cuda", generator=g)
    qkv = ops.QkvHandle(E, H)
    knob(0)
    ref = [t.clone() for t in qkv(a, w, bias, B, T)]
    knob(1)
    got = qkv(a, w, bias, B, T)
    for x, y in zip(ref, got):
        assert torch.equal(x[..., :T] if x.shape[-1] != 64 else x, y[..., :T] if y.shape[-1] != 64 else y)
    assert N.lib.esmk_debug_set(b"no_such_knob", ctypes.c_double(1)) != 0


@pytest.mark.parametrize("lens", [[1022, 1022, 1022, 1022], [150, 33, 97, 128, 64], [763, 336]])
def test_forward_bits_do_not_depend_on_the_launch_form(lens):
    """Whole forward (650M dims, 3 layers): logits, representations and contacts with the combined launch forced on ==
    forced off == the library's choice, padded and token-packed."""
    L, E, H = 3, 1280, 20
    sd = synth_esm2_state_dict(L, E, H, seed=11)
    with skip_param_init():
        model = esm.ESM2(L, E, H).eval()
    model.load_state_dict(sd)
    model = model.cuda()
    gen = torch.Generator().manual_seed(3)
    toks = torch.full((len(lens), max(lens)), 1, dtype=torch.int64)  # <pad>
    for b, n in enumerate(lens):  # <cls> residues <eos> <pad>... as BatchConverter yields it
        toks[b, 0], toks[b, n - 1] = 0, 2
        toks[b, 1:n - 1] = torch.randint(4, 24, (n - 2,), generator=gen)
    toks = toks.cuda()
    res = {}
    with torch.no_grad():
        for mode in (0, 1, -1):
            knob(mode)
            out = model(toks, repr_layers=[0, 1, L], return_contacts=True)
            pk = model.forward_varlen(toks, repr_layers=[L], min_saving=None)
            res[mode] = (out["logits"].clone(), out["representations"][L].clone(), out["representations"][1].clone(),
                         out["contacts"].clone(), pk["logits"].clone(), pk["representations"][L].clone())
    for mode in (1, -1):
        for i, (x, y) in enumerate(zip(res[0], res[mode])):
            assert torch.equal(x, y), (mode, i, float((x - y).abs().max()))
