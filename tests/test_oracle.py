"""The oracle (oracle/esm2_oracle.py) against fixtures produced by the reference implementation
itself (tests/golden/make_golden.py).  CPU only."""
import glob
import os

import pytest
import torch

from esm_amd.synth import synth_esm2_state_dict
from oracle.esm2_oracle import esm2_forward

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "esm2_*.pt")))


def test_fixtures_present():
    assert len(GOLDEN) >= 4


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_matches_reference_fixture(path):
    fix = torch.load(path, weights_only=False)
    d = fix["dims"]
    sd = synth_esm2_state_dict(d["L"], d["E"], d["H"], seed=d["seed"])
    chk = float(sum(v.double().sum() for k, v in sd.items() if k != "lm_head.weight"))
    assert abs(chk - fix["weights_checksum"]) < 1e-6 * max(1.0, abs(chk)), "synthetic weight generator drifted"
    out = esm2_forward(sd, fix["tokens"], d["L"], d["H"], repr_layers=range(d["L"] + 1), return_contacts=True)
    nonpad = fix["tokens"].ne(1)
    assert (out["logits"] - fix["logits"])[nonpad].abs().max() < 2e-5
    for layer, ref in fix["representations"].items():
        assert (out["representations"][layer] - ref)[nonpad].abs().max() < 2e-5, layer
    if fix["attentions"] is not None:
        assert (out["attentions"] - fix["attentions"]).abs().max() < 1e-6
    assert (out["contacts"] - fix["contacts"]).abs().max() < 2e-5


def test_oracle_shape_pin():
    # reference tests/test_load_all.py:42-47: logits (2,3,33) for [[0,1,2],[3,4,5]] (token 1 = <pad>)
    sd = synth_esm2_state_dict(2, 128, 2, seed=3)
    out = esm2_forward(sd, torch.tensor([[0, 1, 2], [3, 4, 5]]), 2, 2)
    assert out["logits"].shape == (2, 3, 33)
    assert torch.isfinite(out["logits"]).all()
