"""Small-batch probe: one forward of B sequences against the same B sequences as N_S sub-batches on N_S HIP streams / host threads /
engine instances running concurrently.  At B = 4 the persistent GEMMs leave 38 % of the CUs idle (fc2 / out-proj: 160 half-height tiles
on 256 CUs); workgroups without a tile exit at once, so a second stream's kernels can take the idle CUs.  650M dims, T = 1024.

    python tools/bench_dual_stream.py [--batches 1,2,4,8] [--streams 2] [--steps 20]
"""
import argparse
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import esm  # noqa: E402
from esm_amd.synth import ESM2_DIMS, skip_param_init, synth_esm2_state_dict, synth_tokens  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="2,4,8,16")
    ap.add_argument("--streams", type=int, default=2)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--seq-len", type=int, default=1022)
    ap.add_argument("--one-thread", action="store_true", help="ONE host thread issues the sub-batches one after the other on their "
                    "streams (what a forward() that splits its batch internally would do) instead of one thread per stream")
    a = ap.parse_args()
    L, E, H = ESM2_DIMS["esm2_t33_650M_UR50D"]
    sd = synth_esm2_state_dict(L, E, H, seed=0)
    models = []
    for _ in range(a.streams):
        with skip_param_init():
            m = esm.ESM2(L, E, H).eval()
        m.load_state_dict(sd)
        models.append(m.cuda())
    streams = [torch.cuda.Stream() for _ in range(a.streams)]
    for B in (int(b) for b in a.batches.split(",")):
        toks = synth_tokens(B, a.seq_len, seed=1).cuda()
        with torch.no_grad():
            for _ in range(3):
                ref = models[0](toks, repr_layers=[L])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                models[0](toks, repr_layers=[L])
            torch.cuda.synchronize()
            t_single = (time.perf_counter() - t0) / a.steps
        ns = min(a.streams, B)
        parts = [toks[i * B // ns:(i + 1) * B // ns].contiguous() for i in range(ns)]
        if a.one_thread:
            outs = [None] * ns
            with torch.no_grad():
                def step():
                    for i in range(ns):
                        with torch.cuda.stream(streams[i]):
                            outs[i] = models[i](parts[i], repr_layers=[L])
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    step()
                torch.cuda.synchronize()
                t_multi = (time.perf_counter() - t0) / a.steps
            same = all(torch.equal(outs[i]["representations"][L], ref["representations"][L][i * B // ns:(i + 1) * B // ns]) for i in range(ns))
            rs = B * a.seq_len
            print(f"B = {B}: single stream {t_single * 1e3:7.3f} ms = {rs / t_single / 1e3:7.1f} k residues/s;  ONE thread, {ns} streams x B = {B // ns}: "
                  f"{t_multi * 1e3:7.3f} ms = {rs / t_multi / 1e3:7.1f} k residues/s ({t_single / t_multi:.3f} x);  bits equal: {same}", flush=True)
            continue
        outs = [None] * ns
        barrier = threading.Barrier(ns + 1)

        def worker(i, steps):
            with torch.cuda.stream(streams[i]), torch.no_grad():
                for _ in range(3):
                    models[i](parts[i], repr_layers=[L])
                streams[i].synchronize()
                barrier.wait()
                for _ in range(steps):
                    outs[i] = models[i](parts[i], repr_layers=[L])
                streams[i].synchronize()
                barrier.wait()

        ts = [threading.Thread(target=worker, args=(i, a.steps)) for i in range(ns)]
        for t in ts:
            t.start()
        barrier.wait()
        t0 = time.perf_counter()
        barrier.wait()
        t_multi = (time.perf_counter() - t0) / a.steps
        for t in ts:
            t.join()
        torch.cuda.synchronize()
        same = all(torch.equal(outs[i]["representations"][L], ref["representations"][L][i * B // ns:(i + 1) * B // ns]) for i in range(ns))
        rs = B * a.seq_len
        print(f"B = {B}: single stream {t_single * 1e3:7.3f} ms = {rs / t_single / 1e3:7.1f} k residues/s;  {ns} streams x B = {B // ns}: "
              f"{t_multi * 1e3:7.3f} ms = {rs / t_multi / 1e3:7.1f} k residues/s ({t_single / t_multi:.3f} x);  bits equal: {same}", flush=True)


if __name__ == "__main__":
    main()
