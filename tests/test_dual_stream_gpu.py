"""ESM2.forward splits medium batches into two half-batches on two HIP streams (esm_amd/esm2.py _dual_stream_window, round 6:
the second half's persistent kernels take the CUs the first half's partly filled rounds of tiles leave idle).  Sequences are
independent and the kernels batch-invariant, so every output must carry the bits of the one-stream forward — representations,
logits, attention maps, contacts, padded batches, odd batch sizes — and the caller's stream must see both halves."""
import pytest
import torch

import esm
from esm_amd.synth import synth_esm2_state_dict, synth_tokens

pytestmark = pytest.mark.gpu


def _model(L, E, H, seed):
    m = esm.ESM2(L, E, H).eval()
    m.load_state_dict(synth_esm2_state_dict(L, E, H, seed=seed))
    return m.cuda()


@pytest.mark.parametrize("B,n,dims", [(8, 1022, (2, 1280, 20)), (7, 300, (3, 320, 20)), (2, 40, (2, 128, 2))])
def test_two_half_batches_give_the_bits_of_one_stream(monkeypatch, B, n, dims):
    L, E, H = dims
    model = _model(L, E, H, seed=21)
    toks = synth_tokens(B, n, seed=5)
    toks[1, n // 2] = 2          # a padded sequence in the first half ...
    toks[1, n // 2 + 1:] = 1
    toks[B - 1, 17] = 32         # ... a <mask> in the second
    toks = toks.cuda()
    monkeypatch.setenv("ESM_AMD_DUAL_STREAM", "0")
    with torch.no_grad():
        one = model(toks, repr_layers=[0, 1, L], return_contacts=True)
        one_c = model.predict_contacts(toks)
    n_before = getattr(model._engine, "dual_calls", 0)
    monkeypatch.setenv("ESM_AMD_DUAL_STREAM", "1:100000000")  # every batch of >= 2 sequences
    with torch.no_grad():
        two = model(toks, repr_layers=[0, 1, L], return_contacts=True)
        two_c = model.predict_contacts(toks)
        # consumed right away on the caller's stream: it must wait for the side stream's half
        nonpad = toks.ne(1)
        s = two["representations"][L][nonpad].sum().item()
    assert model._engine.stream2 is not None and model._engine.dual_calls == n_before + 1, "the dual-stream path did not run (once: the fused contact map stays on one stream)"
    # non-pad positions (values at padded positions are unspecified: they depend on what the workspace held, README)
    assert s == one["representations"][L][nonpad].sum().item()
    for l in (0, 1, L):
        assert torch.equal(one["representations"][l][nonpad], two["representations"][l][nonpad]), l
    assert torch.equal(one["logits"][nonpad], two["logits"][nonpad])
    assert torch.equal(one["attentions"], two["attentions"])   # exact zeros on pad rows / columns, defined everywhere
    for b in range(B):
        nb = int(nonpad[b].sum().item()) - 2                    # contact map of sequence b: its own residues
        assert torch.equal(one["contacts"][b, :nb, :nb], two["contacts"][b, :nb, :nb]), b
    assert torch.equal(one_c, two_c)
    # nothing of a workspace may be read before it is written: both workspaces = 0xFF bytes (NaN in fp16 / fp32), padded batch again
    model._engine.workspace.fill_(255)
    model._engine.workspace2.fill_(255)
    with torch.no_grad():
        three = model(toks, repr_layers=[L])
    assert model._engine.dual_calls == n_before + 2
    assert torch.equal(one["representations"][L][nonpad], three["representations"][L][nonpad])
    assert torch.equal(one["logits"][nonpad], three["logits"][nonpad])


def test_default_window_and_switch(monkeypatch):
    from esm_amd.esm2 import _dual_stream_window

    monkeypatch.delenv("ESM_AMD_DUAL_STREAM", raising=False)
    from esm_amd.esm2 import _dual_stream_wanted

    assert all(_dual_stream_wanted(b * 1024) for b in (4, 8, 16, 32, 48))          # measured + 2 ... 9 %
    assert not any(_dual_stream_wanted(b * 1024) for b in (1, 2, 6, 64, 128))      # measured <= 0 / whole rounds of tiles
    monkeypatch.setenv("ESM_AMD_DUAL_STREAM", "0")
    assert _dual_stream_window() is None
    # the first forward of a new length always runs on one stream (its RoPE table must exist before a side stream reads it)
    monkeypatch.setenv("ESM_AMD_DUAL_STREAM", "1:100000000")
    model = _model(2, 128, 2, seed=3)
    toks = synth_tokens(4, 50, seed=2).cuda()
    with torch.no_grad():
        a = model(toks)
        assert model._engine.stream2 is None
        b = model(toks)
    assert model._engine.stream2 is not None and torch.equal(a["logits"], b["logits"])
