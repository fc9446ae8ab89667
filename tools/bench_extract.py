"""End-to-end timing of the extraction driver (esm_amd.extract.extract) on one MI355X, 650M dimensions:
FASTA strings -> tokens -> forward -> device->host copy -> per-sequence results, next to the bare forward rate.
    python tools/bench_extract.py [--seqs 256] [--include mean] [--write]"""
import argparse, json, os, sys, tempfile, time, pathlib
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import esm
from esm_amd.extract import extract, make_embed_fn
from esm_amd.synth import ESM2_DIMS, synth_esm2_state_dict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seqs", type=int, default=256)
    ap.add_argument("--length", type=int, default=1022)
    ap.add_argument("--include", nargs="+", default=["mean"])
    ap.add_argument("--write", action="store_true", help="also torch.save the per-sequence files")
    ap.add_argument("--mixed", action="store_true", help="log-normal lengths (median ~270, clipped to [30, length])")
    ap.add_argument("--no-varlen", action="store_true", help="padded batches only")
    a = ap.parse_args()
    L, E, H = ESM2_DIMS["esm2_t33_650M_UR50D"]
    model = esm.ESM2(L, E, H).eval()
    model.load_state_dict(synth_esm2_state_dict(L, E, H, seed=0))
    dev = torch.device("cuda", 0)
    model = model.to(dev)
    g = torch.Generator().manual_seed(1)
    aas = "LAGVSERTIDPKQNFYMHWC"
    lens = [a.length] * a.seqs
    if a.mixed:
        lens = torch.exp(torch.randn(a.seqs, generator=g) * 0.7 + 5.6).clamp(30, a.length).long().tolist()
    seqs = ["".join(aas[i] for i in torch.randint(0, 20, (n,), generator=g).tolist()) for n in lens]
    ds = esm.FastaBatchedDataset([f"s{i}" for i in range(a.seqs)], seqs)
    alphabet = esm.Alphabet.from_architecture("ESM-1b")
    fwd = make_embed_fn(model, varlen=not a.no_varlen)
    with tempfile.TemporaryDirectory() as tmp:
        out_dir = pathlib.Path(tmp) if a.write else None
        for rep in range(2):  # first pass warms up the pinned pool and the engine
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            extract(ds, alphabet, fwd, L, E, [L], a.include, output_dir=out_dir, toks_per_batch=64 * 1024, device=dev,
                    gather_mean=True, log=lambda s: None)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
    what = f"{a.seqs} x {'mixed (median 270) up to ' if a.mixed else ''}{a.length} residues"
    print(json.dumps({"workload": f"{what}, include={a.include}, write={a.write}, varlen={not a.no_varlen}",
                      "end_to_end_residues_per_s": round(sum(lens) / dt, 1), "seconds": round(dt, 3)}))


if __name__ == "__main__":
    main()
