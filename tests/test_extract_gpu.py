"""The sharded extraction driver (python -m esm_amd.extract, the repo's mirror of the reference's
scripts/extract.py) end to end on one MI355X: synthetic checkpoint + FASTA -> per-sequence .pt files and the
gathered mean-embedding matrix, checked against the oracle."""
import os
import subprocess
import sys

import pytest
import torch

import _contract as C
from esm_amd.synth import synth_esm2_state_dict, write_esm2_checkpoint
from oracle.esm2_oracle import esm2_forward

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# toks_per_batch 300: small batches with little padding (padded path); 1300: one batch of all five sequences,
# run token-packed (default) and padded (--no_varlen)
@pytest.mark.parametrize("tpb,extra", [("300", []), ("1300", []), ("1300", ["--no_varlen"])],
                         ids=["small_batches", "one_batch_packed", "one_batch_padded"])
def test_extract_cli_matches_oracle(tmp_path, tpb, extra):
    L, E, H = 3, 320, 20  # esm2_t6_8M width (head_dim 16), 3 layers
    ckpt = write_esm2_checkpoint(str(tmp_path), "esm2_synth_8M", L, E, H, seed=5)
    g = torch.Generator().manual_seed(3)
    aas = "LAGVSERTIDPKQNFYMHWC"
    seqs = {f"prot{i}": "".join(aas[j] for j in torch.randint(0, 20, (n,), generator=g).tolist())
            for i, n in enumerate([57, 130, 33, 250, 91])}
    fasta = tmp_path / "in.fasta"
    fasta.write_text("".join(f">{k}\n{v}\n" for k, v in seqs.items()))
    out_dir = tmp_path / "out"
    env = dict(os.environ, PYTHONPATH=ROOT, TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD="1")
    subprocess.run([sys.executable, "-m", "esm_amd.extract", ckpt, str(fasta), str(out_dir), "--repr_layers", "-1", "0",
                    "--include", "mean", "per_tok", "bos", "--toks_per_batch", tpb, "--mean_matrix",
                    str(tmp_path / "means.pt")] + extra, check=True, env=env, cwd=ROOT, timeout=600)
    sd = synth_esm2_state_dict(L, E, H, seed=5)
    from esm_amd import Alphabet

    alphabet = Alphabet.from_architecture("ESM-1b")
    means = torch.load(tmp_path / "means.pt", weights_only=False)
    assert means["labels"] == list(seqs)
    for i, (label, s) in enumerate(seqs.items()):
        toks = torch.tensor([[alphabet.cls_idx] + alphabet.encode(s) + [alphabet.eos_idx]])
        ref = esm2_forward(sd, toks, L, H, repr_layers=[0, L])
        # the ONE parity contract (tests/_contract.py); the driver ran in a subprocess: the floor in this environment's mode
        floor = C.floor_forward(sd, toks, L, H, fold=C.default_fold(E, H), repr_layers=[L])["representations"][L][0]
        r = torch.load(out_dir / f"{label}.pt", weights_only=False)
        assert r["label"] == label and sorted(r["representations"]) == [0, L]
        for l in (0, L):
            full = ref["representations"][l][0]
            want = full[1:len(s) + 1]
            got = r["representations"][l]
            assert got.shape == want.shape
            scale = full.abs().max().item()
            if l == 0:
                assert (got - want).abs().max().item() < 1e-6 * scale + 1e-7
                bound = 1e-6 * scale + 1e-7
            else:
                C.check_tensors(f"extract[{tpb}{' '.join(extra)}] {label} repr[{l}]", got, want, floor[1:len(s) + 1])
                bound = max(C.CONTRACT, C.SLACK * C.errors(floor, full)[1]) * scale
            # pooled outputs: means / single rows of the same rows — never above the per-token bound
            assert (r["mean_representations"][l] - want.mean(0)).abs().max().item() <= bound
            assert (r["bos_representations"][l] - full[0]).abs().max().item() <= bound
        assert (means["mean_representations"][L][i] - ref["representations"][L][0, 1:len(s) + 1].mean(0)).abs().max().item() <= bound


def test_extract_contacts_without_attention_maps(tmp_path):
    """--include contacts: the driver asks for ``contacts_only`` (csrc/contacts.hip), padded batches; every saved
    map is the [n,n] crop of reference scripts/extract.py:123-124 and matches the oracle on the sequence alone."""
    L, E, H = 3, 128, 2
    ckpt = write_esm2_checkpoint(str(tmp_path), "esm2_synth_ct", L, E, H, seed=6)
    g = torch.Generator().manual_seed(4)
    aas = "LAGVSERTIDPKQNFYMHWC"
    seqs = {f"p{i}": "".join(aas[j] for j in torch.randint(0, 20, (n,), generator=g).tolist())
            for i, n in enumerate([40, 131, 77])}
    fasta = tmp_path / "in.fasta"
    fasta.write_text("".join(f">{k}\n{v}\n" for k, v in seqs.items()))
    out_dir = tmp_path / "out"
    env = dict(os.environ, PYTHONPATH=ROOT, TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD="1")
    subprocess.run([sys.executable, "-m", "esm_amd.extract", ckpt, str(fasta), str(out_dir), "--repr_layers", "-1",
                    "--include", "mean", "contacts", "--toks_per_batch", "600"], check=True, env=env, cwd=ROOT,
                   timeout=600)
    sd = synth_esm2_state_dict(L, E, H, seed=6)
    from esm_amd import Alphabet

    alphabet = Alphabet.from_architecture("ESM-1b")
    for label, s in seqs.items():
        toks = torch.tensor([[alphabet.cls_idx] + alphabet.encode(s) + [alphabet.eos_idx]])
        ref = esm2_forward(sd, toks, L, H, repr_layers=[L], return_contacts=True)
        r = torch.load(out_dir / f"{label}.pt", weights_only=False)
        assert r["contacts"].shape == (len(s), len(s))
        assert (r["contacts"] - ref["contacts"][0]).abs().max().item() < 5e-3
        full = ref["representations"][L][0]
        floor = C.floor_forward(sd, toks, L, H, fold=C.default_fold(E, H), repr_layers=[L])["representations"][L][0]
        bound = max(C.CONTRACT, C.SLACK * C.errors(floor, full)[1]) * full.abs().max().item()
        assert (r["mean_representations"][L] - full[1:len(s) + 1].mean(0)).abs().max().item() <= bound
