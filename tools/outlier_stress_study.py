"""Outlier channels in the residual stream (SURVEY.md §7.4's "stress" weight set; ADVICE r5: "a parity run on weights with
outlier channels").  GPU: the engine in its modes on esm_amd.synth.add_outlier_channels weights of growing magnitude, each
against the fp32 oracle (test infrastructure) and the fp16-operand floor in the engine's own form on the same inputs.

    python tools/outlier_stress_study.py [--magnitudes 0,200,2000,20000] > profiles/r6_outlier_stress_study.log

What it answers: does the LayerNorm fold (operand rows fp16(x - mean), un-normalised; gains folded into row-centred weight
images) lose anything against the plain mode (operand rows = the normalised LayerNorm output) when a few channels are 100 ...
10^4 x the ordinary stream and the LayerNorm gains silence them?  It does, in proportion to the outliers' size, through two
terms (separated on the CPU with variants of the oracle's fold form, not kept in the tree):
  (1) column j of a folded image holds gamma_j w_ij - c_i (c_i: the row's centring constant), so a channel with a tiny gain holds
      -c_i alone, and the fp16 rounding of x_j and of c_i is multiplied by the large x_j.  With two outliers up and two down
      (--balanced) this is the whole effect: an UN-CENTRED image + the mean correction - rstd (mean - centre) sum_j w'_ij in the
      consumers' epilogues brings the fold's floor back onto the plain one (7.8e-4 / 1.05e-3 at 2000 x, 7.8e-4 / 1.13e-3 at
      20000 x) — one more FMA per element and two more operands in the q/k, v and fc1 epilogues; not built;
  (2) same-signed outliers (the default set: three up, one down) also move the row mean, i.e. give every ordinary channel the
      same offset, and the LayerNorm bias takes it back: in the plain mode that happens BEFORE the rounding and the product with
      the rounded weights, in the fold the bias enters exactly (fp32 W . beta) while its counterpart goes through the rounded
      image — with (1) removed the fold's floor is still 2.8e-3 / 3.9e-3 at 2000 x; a robust centre (median of the slab means)
      does not change that; taking the bias through the same rounded image (sum_j w'_ij beta_j / gamma_j, pack time) does:
      9.0e-4 / 1.4e-3 at 2000 x, 3.9e-3 / 5.1e-3 at 20000 x (what remains is the rounding of the offset rows themselves).
The plain mode rounds gamma_j (x_j - mean) rstd + beta_j and sees neither.  The engine sits on its form's floor in every line,
and the package's default checks the gains and leaves the fold off for such checkpoints (esm_amd/esm2.py ln_fold_hazard)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import _contract as C  # noqa: E402
import esm  # noqa: E402
from esm_amd.synth import add_outlier_channels, skip_param_init, synth_esm2_state_dict, synth_tokens  # noqa: E402
from oracle.esm2_oracle import esm2_forward  # noqa: E402

MODES = [("default", {}), ("fold (forced)", {"ESM_AMD_LN_FOLD": "1"}), ("plain", {"ESM_AMD_LN_FOLD": "0"}),
         ("f16x2a", {"ESM_AMD_OPERAND": "f16x2a"}), ("f16x3", {"ESM_AMD_OPERAND": "f16x3"})]


def errs(a, b, mask):
    l2, mx = C.errors(a, b, mask)
    return f"{l2:.2e} / {mx:.2e}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--magnitudes", default="0,200,2000,20000")
    ap.add_argument("--dims", default="33,1280,20")
    ap.add_argument("--balanced", action="store_true", help="two outliers up, two down: no common offset of the ordinary channels")
    ap.add_argument("--floors-only", action="store_true", help="CPU: the oracle's floors, no engine")
    a = ap.parse_args()
    L, E, H = (int(x) for x in a.dims.split(","))
    toks = synth_tokens(2, 254, seed=1)
    toks[1, 100] = 2
    toks[1, 101:] = 1
    nonpad = toks.ne(1)
    print(__doc__.split("    python")[0].strip())
    print(f"\n{L} x {E} x {H} heads, tokens {tuple(toks.shape)} (one sequence padded from 101); errors are L2 / max norm, relative")
    for mag in (float(x) for x in a.magnitudes.split(",")):
        sd = synth_esm2_state_dict(L, E, H, seed=0)
        idx = add_outlier_channels(sd, L, E, magnitude=mag, balanced=a.balanced) if mag > 0 else torch.tensor([], dtype=torch.long)
        ref = esm2_forward(sd, toks, L, H, repr_layers=[16, L], return_contacts=True)
        r16 = ref["representations"][16][nonpad]
        ordinary = torch.ones(E, dtype=torch.bool)
        ordinary[idx] = False
        print(f"\n== outlier magnitude {mag:g}: channels {idx.tolist()}, |x| at layer 16 "
              f"{[round(v, 1) for v in r16[:, idx].abs().mean(0).tolist()]}, ordinary std {r16[:, ordinary].std().item():.2f}, "
              f"largest |x| {r16.abs().max().item():.1f}")
        floors = {f: C.floor_forward(sd, toks, L, H, fold=f, repr_layers=[16, L], return_contacts=True) for f in (False, True)}
        for f in (False, True):
            fl = floors[f]
            print(f"   floor, {'fold' if f else 'plain'} form: repr[{L}] {errs(fl['representations'][L], ref['representations'][L], nonpad)}   "
                  f"repr[16] ordinary {errs(fl['representations'][16][..., ordinary], ref['representations'][16][..., ordinary], nonpad)}   "
                  f"logits {errs(fl['logits'], ref['logits'], nonpad)}   "
                  f"contact logits {C.contact_logit_errors(fl['contacts'][0], ref['contacts'][0])[1]:.2e}   "
                  f"argmax {C.raw_argmax_agreement(fl['logits'], ref['logits'], nonpad):.4f}")
        for name, env in ([] if a.floors_only else MODES):
            for k, v in env.items():
                os.environ[k] = v
            try:
                with skip_param_init():
                    model = esm.ESM2(L, E, H).eval()
                model.load_state_dict(sd)
                model = model.cuda()
                with torch.no_grad():
                    o = model(toks.cuda(), repr_layers=[16, L], return_contacts=True)
                o = {"representations": {k: v.float().cpu() for k, v in o["representations"].items()}, "logits": o["logits"].float().cpu(),
                     "contacts": o["contacts"].float().cpu()}
                finite = all(torch.isfinite(t).all().item() for t in (o["logits"], o["representations"][L]))
                print(f"   {name:15s}: repr[{L}] {errs(o['representations'][L], ref['representations'][L], nonpad)}   "
                      f"repr[16] ordinary {errs(o['representations'][16][..., ordinary], ref['representations'][16][..., ordinary], nonpad)}   "
                      f"logits {errs(o['logits'], ref['logits'], nonpad)}   "
                      f"contact logits {C.contact_logit_errors(o['contacts'][0], ref['contacts'][0])[1]:.2e}   "
                      f"argmax {C.raw_argmax_agreement(o['logits'], ref['logits'], nonpad):.4f}   finite {finite}"
                      + (f"   [gain hazard h = {model._fold_hazard:.2f}: runs {'with' if model.ln_fold_active() else 'WITHOUT'} the fold]"
                         if name == "default" else ""))
                del model
                torch.cuda.empty_cache()
            finally:
                for k in env:
                    os.environ.pop(k, None)


if __name__ == "__main__":
    main()
