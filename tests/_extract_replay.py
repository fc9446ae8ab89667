"""Test helper (never part of the product): the call sequence of the reference's bulk-extraction script, step by step,
against whatever ``esm`` package is importable — on the GPU box /root/reference does not exist, so the unmodified script
cannot be the thing that runs there; this replay issues the same calls in the same order with the same arguments
(reference scripts/extract.py:63-131; each step cites its line).  tests/test_extract_script_gpu.py runs the real script
instead whenever it is present.

    python tests/_extract_replay.py <checkpoint> <fasta> <out_dir> --repr_layers .. --include .. [--toks_per_batch N]
                                    [--truncation_seq_length N]
"""
import argparse
import pathlib
import sys

import torch


def replay(a):
    from esm import FastaBatchedDataset, MSATransformer, pretrained

    model, alphabet = pretrained.load_model_and_alphabet(a.model_location)  # extract.py:64
    model.eval()  # :65
    assert not isinstance(model, MSATransformer)  # :66-69
    assert torch.cuda.is_available()
    model = model.cuda()  # :70-72
    dataset = FastaBatchedDataset.from_file(a.fasta_file)  # :74
    batches = dataset.get_batch_indices(a.toks_per_batch, extra_toks_per_seq=1)  # :75
    loader = torch.utils.data.DataLoader(  # :76-78
        dataset, collate_fn=alphabet.get_batch_converter(a.truncation_seq_length), batch_sampler=batches)
    a.output_dir.mkdir(parents=True, exist_ok=True)  # :81
    want_contacts = "contacts" in a.include  # :82
    nl = model.num_layers
    assert all(-(nl + 1) <= i <= nl for i in a.repr_layers)  # :84
    layers = [(i + nl + 1) % (nl + 1) for i in a.repr_layers]  # :85
    with torch.no_grad():  # :87
        for labels, strs, toks in loader:  # :88
            toks = toks.to(device="cuda", non_blocking=True)  # :92-93
            out = model(toks, repr_layers=layers, return_contacts=want_contacts)  # :95
            out["logits"].to(device="cpu")  # :97
            reps = {l: t.to(device="cpu") for l, t in out["representations"].items()}  # :98-100
            contacts = out["contacts"].to(device="cpu") if want_contacts else None  # :101-102
            for i, label in enumerate(labels):  # :104
                path = a.output_dir / f"{label}.pt"  # :105
                path.parent.mkdir(parents=True, exist_ok=True)  # :106
                res = {"label": label}  # :107
                n = min(a.truncation_seq_length, len(strs[i]))  # :108
                if "per_tok" in a.include:  # :111-115
                    res["representations"] = {l: t[i, 1 : n + 1].clone() for l, t in reps.items()}
                if "mean" in a.include:  # :116-120
                    res["mean_representations"] = {l: t[i, 1 : n + 1].mean(0).clone() for l, t in reps.items()}
                if "bos" in a.include:  # :121-124
                    res["bos_representations"] = {l: t[i, 0].clone() for l, t in reps.items()}
                if want_contacts:  # :125-126
                    res["contacts"] = contacts[i, :n, :n].clone()
                torch.save(res, path)  # :128-131


def main():
    p = argparse.ArgumentParser()  # the script's own arguments (extract.py:21-60)
    p.add_argument("model_location")
    p.add_argument("fasta_file", type=pathlib.Path)
    p.add_argument("output_dir", type=pathlib.Path)
    p.add_argument("--toks_per_batch", type=int, default=4096)
    p.add_argument("--repr_layers", type=int, default=[-1], nargs="+")
    p.add_argument("--include", type=str, nargs="+", choices=["mean", "per_tok", "bos", "contacts"], required=True)
    p.add_argument("--truncation_seq_length", type=int, default=1022)
    p.add_argument("--nogpu", action="store_true")
    replay(p.parse_args())


if __name__ == "__main__":
    sys.exit(main())
