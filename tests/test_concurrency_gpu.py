"""Two engines driven from two host threads on two HIP streams at the same time (a serving process with one model per
worker thread): the library's process-wide state is read-only after first use (knobs behind call_once / atomics)
and every call works on the caller's stream and workspace, so concurrent forwards must give the bits of serial ones."""
import threading

import pytest
import torch

import esm
from esm_amd.synth import synth_esm2_state_dict, synth_tokens

pytestmark = pytest.mark.gpu


def test_two_threads_two_streams_same_bits_as_serial():
    L, E, H = 4, 256, 4
    sd = synth_esm2_state_dict(L, E, H, seed=12)
    models = []
    for _ in range(2):
        m = esm.ESM2(L, E, H).eval()
        m.load_state_dict(sd)
        models.append(m.cuda())
    toks = [synth_tokens(6, 700, seed=3).cuda(), synth_tokens(3, 333, seed=4).cuda()]
    toks[1][2, 200] = 2
    toks[1][2, 201:] = 1
    with torch.no_grad():
        serial = [models[i](toks[i], repr_layers=[L], return_contacts=True) for i in range(2)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(2)]
    results, errors = [None, None], []

    def worker(i):
        try:
            with torch.cuda.stream(streams[i]), torch.no_grad():
                for _ in range(8):  # long enough for the two streams to overlap
                    results[i] = models[i](toks[i], repr_layers=[L], return_contacts=True)
            streams[i].synchronize()
        except Exception as e:  # pragma: no cover
            errors.append(e)

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for i in range(2):
        nonpad = toks[i].ne(1)
        assert torch.equal(results[i]["representations"][L][nonpad], serial[i]["representations"][L][nonpad])
        assert torch.equal(results[i]["logits"][nonpad], serial[i]["logits"][nonpad])
        assert torch.equal(results[i]["contacts"], serial[i]["contacts"])
