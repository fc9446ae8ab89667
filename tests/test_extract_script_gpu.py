"""The reference's bulk-extraction script on the REAL engine (Missing-6): checkpoint file ->
``pretrained.load_model_and_alphabet`` -> ``.eval().cuda()`` -> ``FastaBatchedDataset`` + DataLoader with
``alphabet.get_batch_converter(truncation_seq_length)`` -> ``toks.to("cuda", non_blocking=True)`` ->
``model(toks, repr_layers=.., return_contacts=..)`` -> per-label crops -> ``torch.save``
(reference scripts/extract.py:63-131).  Where /root/reference exists the UNMODIFIED script file runs (with this repo's
``esm`` package on the path); on the GPU box, which has no /root/reference, tests/_extract_replay.py issues the same
calls line by line.  The written files are checked against the CPU oracle."""
import os
import subprocess
import sys

import pytest
import torch

from esm_amd import Alphabet
from esm_amd.synth import synth_esm2_state_dict, write_esm2_checkpoint
from oracle.esm2_oracle import esm2_forward

import _contract as C

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SCRIPT = "/root/reference/scripts/extract.py"


@pytest.mark.parametrize("trunc,tpb", [(1022, 4096), (60, 300)], ids=["defaults", "truncated_small_batches"])
def test_reference_extract_call_sequence_on_the_engine(tmp_path, trunc, tpb):
    L, E, H = 6, 320, 20  # esm2_t6_8M_UR50D dimensions (BASELINE configs[0]) — here on the GPU, not --nogpu
    ckpt = write_esm2_checkpoint(str(tmp_path), "esm2_t6_8M_UR50D", L, E, H, seed=9)
    g = torch.Generator().manual_seed(4)
    aas = "LAGVSERTIDPKQNFYMHWC"
    lens = [57, 130, 33, 250, 91, 64, 1]
    seqs = {(f"fam/prot{i}" if i % 3 == 0 else f"prot{i}"): "".join(aas[j] for j in torch.randint(0, 20, (n,), generator=g).tolist())
            for i, n in enumerate(lens)}
    fasta = tmp_path / "in.fasta"
    fasta.write_text("".join(f">{k}\n{v[:40]}\n{v[40:]}\n" for k, v in seqs.items()))  # wrapped sequence lines
    out_dir = tmp_path / "out"
    script = REF_SCRIPT if os.path.exists(REF_SCRIPT) else os.path.join(ROOT, "tests", "_extract_replay.py")
    env = dict(os.environ, PYTHONPATH=ROOT, TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD="1")
    r = subprocess.run([sys.executable, script, ckpt, str(fasta), str(out_dir), "--repr_layers", "-1", "0", "3", "--include", "mean",
                        "per_tok", "bos", "contacts", "--toks_per_batch", str(tpb), "--truncation_seq_length", str(trunc)],
                       capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    print("ran", script)
    sd = synth_esm2_state_dict(L, E, H, seed=9)
    conv = Alphabet.from_architecture("ESM-1b").get_batch_converter(trunc)
    worst = 0.0
    for label, s in seqs.items():
        got = torch.load(out_dir / f"{label}.pt", weights_only=False)
        n = min(trunc, len(s))
        _, _, toks = conv([(label, s)])
        ref = esm2_forward(sd, toks, L, H, repr_layers=[0, 3, 6], return_contacts=True)
        floor = C.floor_forward(sd, toks, L, H, fold=C.default_fold(E, H), repr_layers=[3, 6])
        assert got["label"] == label and sorted(got["representations"]) == [0, 3, 6]
        for l in (0, 3, 6):
            full = ref["representations"][l][0]
            rep = got["representations"][l]
            assert rep.shape == (n, E) and rep.dtype == torch.float32 and rep.is_contiguous()
            scale = full.abs().max().item()
            e = (rep - full[1 : n + 1]).abs().max().item() / scale
            worst = max(worst, e)
            if l == 0:
                assert e < 1e-6, (label, e)
                bound = 1e-6
            else:  # the parity contract (tests/_contract.py): a 6-layer toy model, floor-referenced in both norms
                _, mx = C.check_tensors(f"extract {label} repr[{l}]", rep, full[1 : n + 1], floor["representations"][l][0, 1 : n + 1])
                bound = max(C.CONTRACT, C.SLACK * C.errors(floor["representations"][l][0], full)[1])
            assert (got["mean_representations"][l] - full[1 : n + 1].mean(0)).abs().max().item() <= bound * scale
            assert (got["bos_representations"][l] - full[0]).abs().max().item() <= bound * scale
        assert got["contacts"].shape == (n, n)
        assert (got["contacts"] - ref["contacts"][0, :n, :n]).abs().max().item() < 8e-3, label
    print(f"reference extract call sequence on the engine ({os.path.basename(script)}): worst representation error {worst:.2e}")
