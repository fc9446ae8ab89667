"""ESM2.forward with and without its dual-stream split (esm_amd/esm2.py _dual_stream_window), same process, same model: ms per forward,
residues/s, and bit-equality of the outputs, over a list of (B, L) shapes.  650M dims.

    python tools/bench_dual_forward.py [--shapes 4x1022,8x1022,...] [--steps 20]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import esm  # noqa: E402
from esm_amd.synth import ESM2_DIMS, skip_param_init, synth_esm2_state_dict, synth_tokens  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="4x1022,6x1022,8x1022,12x1022,16x1022,24x1022,32x1022,40x1022,48x1022,64x1022,64x126,32x254,128x126")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--contacts", action="store_true")
    a = ap.parse_args()
    L, E, H = ESM2_DIMS["esm2_t33_650M_UR50D"]
    with skip_param_init():
        model = esm.ESM2(L, E, H).eval()
    model.load_state_dict(synth_esm2_state_dict(L, E, H, seed=0))
    model = model.cuda()

    def run(toks, steps):
        with torch.no_grad():
            for _ in range(3):
                out = model.predict_contacts(toks) if a.contacts else model(toks, repr_layers=[L])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                out = model.predict_contacts(toks) if a.contacts else model(toks, repr_layers=[L])
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps, out

    for shape in a.shapes.split(","):
        B, n = (int(v) for v in shape.split("x"))
        toks = synth_tokens(B, n, seed=1).cuda()
        os.environ["ESM_AMD_DUAL_STREAM"] = "0"
        t1, o1 = run(toks, a.steps)
        os.environ["ESM_AMD_DUAL_STREAM"] = "1:100000000"  # always
        t2, o2 = run(toks, a.steps)
        os.environ.pop("ESM_AMD_DUAL_STREAM")
        same = torch.equal(o1, o2) if a.contacts else (torch.equal(o1["representations"][L], o2["representations"][L]) and torch.equal(o1["logits"], o2["logits"]))
        rs = B * n
        print(f"B = {B:3d} L = {n:4d} ({B * (n + 2):6d} rows): one stream {t1 * 1e3:8.3f} ms = {rs / t1 / 1e3:7.1f} k res/s;  two half-batches "
              f"{t2 * 1e3:8.3f} ms = {rs / t2 / 1e3:7.1f} k res/s  ({t1 / t2:.3f} x);  bits equal: {same}", flush=True)


if __name__ == "__main__":
    main()
