"""How fast is the REFERENCE's own CPU path next to this repo's CPU oracle (bench.py's `cpu_baseline`, kind "port")?

`/root/reference` only exists in the build container, so the GPU box times the oracle (a functional restatement of the
reference, pinned to it by the golden fixtures).  This script times both HERE, in one process each, on the same threads,
on the headline shape (ESM-2 650M dimensions, random-init weights, L = 1022): the ratio lets a reader translate the
bench line's oracle number into "what the imported reference would have done on that host".

    python tools/cpu_reference_vs_oracle.py            # runs both children, prints one line each + the ratio
    python tools/cpu_reference_vs_oracle.py --child reference|oracle   (internal)

The reference side follows tests/_reference_probe.py: sys.path -> /root/reference, `import esm` IS the reference
(esm/model/esm2.py:77-147 forward), weights from esm_amd/synth.py loaded by file path.
"""
import argparse
import importlib.util
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
L, E, H, T = 33, 1280, 20, 1022


def load_synth():
    spec = importlib.util.spec_from_file_location("esm_amd_synth", os.path.join(ROOT, "esm_amd", "synth.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def child(kind, batch, threads):
    import torch

    torch.set_num_threads(threads)
    synth = load_synth()
    sd = synth.synth_esm2_state_dict(L, E, H, seed=0)
    toks = synth.synth_tokens(batch, T, seed=1)
    if kind == "reference":
        sys.path.insert(0, REFERENCE)
        import esm

        assert esm.__file__.startswith(REFERENCE), esm.__file__
        model = esm.model.esm2.ESM2(L, E, H).eval()
        model.load_state_dict(sd)
        fwd = lambda: model(toks, repr_layers=[L])["representations"][L]
    else:
        sys.path.insert(0, ROOT)
        from oracle.esm2_oracle import esm2_forward

        fwd = lambda: esm2_forward(sd, toks, L, H, repr_layers=[L])["representations"][L]
    times = []
    with torch.no_grad():
        for _ in range(3):
            t0 = time.perf_counter()
            out = fwd()
            times.append(time.perf_counter() - t0)
    best = sorted(times[1:])[0]
    print(json.dumps({"kind": kind, "batch": batch, "threads": threads, "seconds": [round(t, 3) for t in times],
                      "residues_per_s": round(batch * T / best, 1), "checksum": float(out.double().abs().mean())}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", default=None)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    args = ap.parse_args()
    if args.child:
        child(args.child, args.batch, args.threads)
        return
    res = {}
    for kind in ("reference", "oracle"):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", kind, "--batch", str(args.batch),
                            "--threads", str(args.threads)], capture_output=True, text=True)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not line:
            print(kind, "failed:", p.stderr[-400:])
            return
        res[kind] = json.loads(line[-1])
        print(line[-1], flush=True)
    r, o = res["reference"], res["oracle"]
    print(f"B = {args.batch}, L = {T}, {args.threads} threads: reference {r['residues_per_s']} residues/s, oracle {o['residues_per_s']} "
          f"residues/s (oracle / reference = {o['residues_per_s'] / r['residues_per_s']:.2f}); mean |repr| {r['checksum']:.6f} vs {o['checksum']:.6f}")


if __name__ == "__main__":
    main()
