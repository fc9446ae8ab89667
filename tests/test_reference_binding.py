"""examples/reference_binding/_esmk.py — the ctypes stub INTEGRATION.md tells a reference maintainer to add — is real
code: on CPU its struct layout and the host-only C entry points are exercised against the built library; on the
MI355X its forward must give the same bits as this repo's own wrapper (both call esmk_forward on the same weights)."""
import ctypes
import importlib.util
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_stub():
    spec = importlib.util.spec_from_file_location("ref_esmk_stub", os.path.join(ROOT, "examples", "reference_binding", "_esmk.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from esm_amd import _native

    mod.lib(_native.LIB_PATH)
    return mod


def test_stub_struct_and_host_calls():
    import esm
    from esm_amd import _native

    stub = load_stub()
    assert ctypes.sizeof(stub.Config) == ctypes.sizeof(_native.EsmkConfig)
    assert [f[0] for f in stub.Config._fields_] == [f[0] for f in _native.EsmkConfig._fields_]
    m = esm.ESM2(6, 320, 20)
    cfg = stub.config_for(m)
    h = ctypes.c_void_p()
    L = stub.lib()
    assert L.esmk_create(ctypes.byref(cfg), ctypes.byref(h)) == 0
    n, w = ctypes.c_size_t(), ctypes.c_size_t()
    assert L.esmk_packed_bytes(h, ctypes.byref(n)) == 0 and n.value > 7_000_000 * 2  # 7.5 M parameters in fp16
    assert L.esmk_workspace_bytes(h, 2, 24, ctypes.c_uint32(1), ctypes.byref(w)) == 0 and w.value > 0
    L.esmk_destroy(h)
    bad = stub.config_for(m)
    bad.num_heads = 7  # 320 % 7 != 0
    assert L.esmk_create(ctypes.byref(bad), ctypes.byref(h)) != 0
    with pytest.raises(RuntimeError, match="divisible"):
        stub._chk(1)


@pytest.mark.gpu
def test_stub_forward_equals_package_forward(monkeypatch):
    import esm
    from esm_amd.synth import synth_esm2_state_dict, synth_tokens

    stub = load_stub()
    L, E, H = 3, 128, 2
    m = esm.ESM2(L, E, H).eval()
    m.load_state_dict(synth_esm2_state_dict(L, E, H, seed=9))
    m = m.cuda()
    toks = synth_tokens(3, 50, seed=2)
    toks[1, 30] = 2
    toks[1, 31:] = 1
    toks = toks.cuda()
    with torch.no_grad():
        want = m(toks, repr_layers=[0, 2, L], return_contacts=True)
        # the stub leaves esmk_config.ln_fold at 0 = "the library's default"; a suite run with ESM_AMD_LN_FOLD=0 (the package's
        # switch) puts the package in the plain mode, so the library's own switch gives the stub the same one
        monkeypatch.setenv("ESMK_LN_FOLD", "1" if m.ln_fold_active() else "0")
        got = stub.forward(m, toks, repr_layers=[0, 2, L], return_contacts=True)
    assert sorted(got) == sorted(want) and sorted(got["representations"]) == [0, 2, L]
    for k in ("logits", "attentions", "contacts"):
        assert torch.equal(got[k], want[k]), k
    for l in (0, 2, L):
        assert torch.equal(got["representations"][l], want["representations"][l]), l
    mh = m.half()
    with torch.no_grad():
        g16 = stub.forward(mh, toks, repr_layers=[L])
        w16 = mh(toks, repr_layers=[L])
    assert g16["logits"].dtype == torch.float16 and torch.equal(g16["representations"][L], w16["representations"][L])
