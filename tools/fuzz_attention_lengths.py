"""Soak / fuzz of the attention path over sequence lengths: random batches of random lengths (1 … 1022 residues,
optionally with interior <pad> / <mask> tokens) through a 2-layer model of 650M width, checking that
  * the token-packed forward equals the padded forward bit for bit on every non-pad position,
  * a sequence alone equals its rows inside a batch,
  * nothing is NaN / inf,
so that tile-tail handling (keys past the end of a row come out of range of the K descriptor in the flash kernel),
masked-tile flags and the lazy softmax offset see every alignment.   python tools/fuzz_attention_lengths.py [--iters 40]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import esm
from esm_amd.synth import synth_esm2_state_dict


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args(argv)
    L, E, H = 2, 1280, 20
    model = esm.ESM2(L, E, H).eval().requires_grad_(False)
    model.load_state_dict(synth_esm2_state_dict(L, E, H, seed=21))
    model = model.cuda()
    g = torch.Generator().manual_seed(a.seed)
    worst = 0
    for it in range(a.iters):
        B = int(torch.randint(1, 9, (1,), generator=g))
        lens = torch.randint(1, 1023, (B,), generator=g).tolist()
        if it % 5 == 0:
            lens[0] = [1, 62, 63, 64, 65, 126, 127, 128, 129, 1022][(it // 5) % 10]
        T = max(lens) + 2
        toks = torch.full((B, T), 1, dtype=torch.int64)
        for b, n in enumerate(lens):
            toks[b, 0] = 0
            toks[b, 1:n + 1] = torch.randint(4, 24, (n,), generator=g)
            toks[b, n + 1] = 2
            if n > 10 and it % 3 == 0:
                toks[b, 1 + int(torch.randint(0, n, (1,), generator=g))] = 32  # <mask>
            if n > 10 and it % 4 == 0:
                toks[b, 1 + int(torch.randint(0, n, (1,), generator=g))] = 1   # interior <pad>
        dev = toks.cuda()
        pad = model(dev, repr_layers=[L])
        pk = model.forward_varlen(toks, repr_layers=[L], min_saving=None)
        nonpad = dev.ne(1)
        r_pad, r_pk = pad["representations"][L], pk["representations"][L]
        assert torch.isfinite(r_pad[nonpad]).all() and torch.isfinite(pad["logits"][nonpad]).all(), (it, lens)
        assert torch.equal(r_pad[nonpad], r_pk[nonpad]), (it, lens, (r_pad - r_pk)[nonpad].abs().max().item())
        assert torch.equal(pad["logits"][nonpad], pk["logits"][nonpad]), (it, lens)
        b = int(torch.randint(0, B, (1,), generator=g))
        n = lens[b]
        alone = model(dev[b:b + 1, :n + 2], repr_layers=[L])["representations"][L]
        keep = dev[b, :n + 2].ne(1)
        assert torch.equal(alone[0][keep], r_pad[b, :n + 2][keep]), (it, lens, b)
        worst = max(worst, T)
    print(f"fuzz ok: {a.iters} random batches (up to {worst} tokens wide): packed == padded == alone, all finite")


if __name__ == "__main__":
    main()
