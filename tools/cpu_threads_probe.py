"""How many host threads should the CPU baseline use?  Times the oracle on one 650M sequence."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esm_amd.synth import synth_esm2_state_dict, synth_tokens
from oracle.esm2_oracle import esm2_forward
sd = synth_esm2_state_dict(33, 1280, 20, seed=0)
toks = synth_tokens(1, 1022, seed=1)
for n in [int(a) for a in sys.argv[1:]] or [32, 64, 128]:
    torch.set_num_threads(n)
    ts = []
    for _ in range(2):
        t0 = time.perf_counter(); esm2_forward(sd, toks, 33, 20, repr_layers=[33]); ts.append(time.perf_counter() - t0)
    print(f"threads={n}: {ts[0]:.2f}s, {ts[1]:.2f}s -> {1022/ts[1]:.1f} residues/s", flush=True)
