"""Timing of BASELINE config 5: esm_msa1b_t12_100M dimensions (12 x 768, 12 heads, FFN 3072), one
128 x 513 MSA per forward, on one MI355X.   python tools/bench_msa.py [--rows 128] [--cols 513] [--steps 10]

Algorithmic cost (SURVEY.md §8 a): 32 D^2 + 4 D (C + R) FLOP per token-layer = 251.4 MFLOP/token,
16.5 TFLOP per 128 x 513 MSA."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import esm
from esm_amd.synth import MSA_DIMS, synth_msa_state_dict, synth_msa_tokens


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=128)
    ap.add_argument("--cols", type=int, default=513)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    L, E, H, F = MSA_DIMS["esm_msa1b_t12_100M_UR50S"]
    ns = argparse.Namespace(layers=L, embed_dim=E, ffn_embed_dim=F, attention_heads=H, dropout=0.1, attention_dropout=0.1,
                            activation_dropout=0.1, max_positions=1024, embed_positions_msa=True,
                            embed_positions_msa_dim=E, max_tokens=2 ** 14, max_tokens_per_msa=2 ** 14)
    model = esm.MSATransformer(ns, esm.Alphabet.from_architecture("msa_transformer")).eval()
    model.load_state_dict(synth_msa_state_dict(L, E, H, F, seed=0))
    model = model.cuda()
    toks = synth_msa_tokens(a.batch, a.rows, a.cols, seed=1).cuda()
    with torch.no_grad():
        for _ in range(a.warmup):
            model(toks, repr_layers=[L])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            model(toks, repr_layers=[L])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        for _ in range(2):
            model(toks, repr_layers=[L], return_contacts=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            model(toks, repr_layers=[L], return_contacts=True)
        torch.cuda.synchronize()
        dtc = (time.perf_counter() - t0) / a.steps
    ntok = a.batch * a.rows * a.cols
    flop = ntok * L * (32.0 * E * E + 4.0 * E * (a.cols + a.rows))
    print(json.dumps({"workload": f"esm_msa1b_t12_100M dims, {a.batch} x {a.rows} x {a.cols} MSA, fp16 operands",
                      "ms_per_forward": round(dt * 1e3, 3), "msa_tokens_per_s": round(ntok / dt, 1),
                      "tflops_algorithmic": round(flop / dt / 1e12, 1), "mfma_roof_frac": round(flop / dt / 2.5e15, 4),
                      "ms_per_forward_with_contacts": round(dtc * 1e3, 3)}))


if __name__ == "__main__":
    main()
