"""ESMFold-front-end-like call (reference esm/esmfold/v1/esmfold.py:118-135: a `.half()` language model, ALL layer
representations) with the engine writing fp16 outputs itself (ESMK_OUT_REPR_LOWP) vs fp32 outputs + torch cast
(ESM_AMD_NATIVE_LOWP=0).     python tools/bench_lowp_outputs.py [--batch 16] [--len 1022]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import esm  # noqa: E402
from esm_amd.synth import ESM2_DIMS, synth_esm2_state_dict, synth_tokens  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--len", type=int, default=1022)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    L, E, H = ESM2_DIMS["esm2_t33_650M_UR50D"]
    model = esm.ESM2(L, E, H).eval()
    model.load_state_dict(synth_esm2_state_dict(L, E, H, seed=0))
    model = model.cuda().half()
    toks = synth_tokens(a.batch, a.len, seed=1).cuda()
    layers = list(range(L + 1))
    for mode in ("0", "1", "0", "1"):
        os.environ["ESM_AMD_NATIVE_LOWP"] = mode
        with torch.no_grad():
            model(toks, repr_layers=layers)
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            t0 = time.perf_counter()
            for _ in range(a.iters):
                out = model(toks, repr_layers=layers)
                s = torch.stack([out["representations"][l] for l in layers], dim=2)  # esmfold.py:135
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.iters
        print(f"ESM_AMD_NATIVE_LOWP={mode}: {dt*1e3:8.2f} ms per call ({a.batch} x {a.len}, {len(layers)} fp16 representations "
              f"+ stack), {a.batch*a.len/dt:9.0f} residues/s, peak {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
        del out, s


if __name__ == "__main__":
    main()
