"""Round-2 additions on the MI355X: the masked row-mean kernel (mean pooling of the extraction driver), weight
re-packing when parameters are replaced, the forward-only warning, duplicate / edge-case handling of the extraction
driver, and bench.py's self-launch path with RCCL at N = 1."""
import json
import os
import subprocess
import sys
import warnings

import pytest
import torch

import esm
from esm_amd.synth import synth_esm2_state_dict, synth_tokens

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("dt", [torch.float32, torch.float16, torch.bfloat16], ids=["f32", "f16", "bf16"])
@pytest.mark.parametrize("B,T,E", [(5, 37, 320), (3, 1024, 1280), (2, 9, 484)])
def test_masked_row_mean_matches_torch(dt, B, T, E):
    """reference scripts/extract.py:113-116: t[i, 1:n+1].mean(0); slices cut at T, empty slice -> NaN."""
    from esm_amd import ops

    g = torch.Generator().manual_seed(B * T + E)
    x = torch.randn((B, T, E), generator=g).to(dt).cuda()
    counts = torch.randint(1, T - 1, (B,), generator=g, dtype=torch.int32)
    counts[0] = T + 5      # longer than the tensor: the slice stops at the last row
    if B > 2:
        counts[2] = 0      # empty sequence
    out = ops.masked_row_mean(x, counts.cuda(), first_row=1)
    assert out.dtype == torch.float32 and out.shape == (B, E)
    for b in range(B):
        want = x[b, 1:int(counts[b]) + 1].float().mean(0)
        if counts[b] == 0:
            assert torch.isnan(out[b]).all() and torch.isnan(want).all()
        else:
            assert (out[b] - want).abs().max().item() < 2e-6 * max(1.0, T / 64), (b, int(counts[b]))
    again = ops.masked_row_mean(x, counts.cuda(), first_row=1)
    assert torch.equal(torch.nan_to_num(out), torch.nan_to_num(again))  # deterministic


def test_replaced_parameters_are_repacked():
    """sync_weights must follow the LIVE parameters.  A replaced Parameter object, load_state_dict
    (plain and assign=True) and tracked in-place edits are seen automatically; a write through `.data` bypasses the
    version counter and needs refresh_engine() (documented in ESM2.forward)."""
    L, E, H = 2, 128, 2
    sd_a, sd_b = synth_esm2_state_dict(L, E, H, seed=1), synth_esm2_state_dict(L, E, H, seed=2)
    toks = synth_tokens(2, 30, seed=3).cuda()
    mk = lambda sd: esm.ESM2(L, E, H).eval().requires_grad_(False)

    ref_b = mk(sd_b)
    ref_b.load_state_dict(sd_b)
    want_b = ref_b.cuda()(toks, repr_layers=[L])["representations"][L]

    m = mk(sd_a)
    m.load_state_dict(sd_a)
    m = m.cuda()
    out_a = m(toks, repr_layers=[L])["representations"][L]
    assert not torch.equal(out_a, want_b)
    m.load_state_dict({k: v.cuda() for k, v in sd_b.items()}, assign=True)  # Parameter objects replaced
    assert torch.equal(m(toks, repr_layers=[L])["representations"][L], want_b)
    m.load_state_dict(sd_a)                                                  # copied in place (version bump)
    assert torch.equal(m(toks, repr_layers=[L])["representations"][L], out_a)
    m.layers[0].fc1.weight = torch.nn.Parameter(sd_b["layers.0.fc1.weight"].cuda(), requires_grad=False)
    mixed = m(toks, repr_layers=[L])["representations"][L]
    assert not torch.equal(mixed, out_a)
    with torch.no_grad():
        m.layers[0].fc1.weight.copy_(sd_a["layers.0.fc1.weight"])            # tracked in-place edit
    assert torch.equal(m(toks, repr_layers=[L])["representations"][L], out_a)
    m.layers[0].fc1.weight.data.copy_(sd_b["layers.0.fc1.weight"])           # untracked: stale until refreshed
    m.refresh_engine()
    assert torch.equal(m(toks, repr_layers=[L])["representations"][L], mixed)


def test_forward_only_warning_once():
    L, E, H = 1, 128, 2
    m = esm.ESM2(L, E, H).eval()
    m.load_state_dict(synth_esm2_state_dict(L, E, H, seed=4))
    m = m.cuda()
    toks = synth_tokens(1, 10, seed=1).cuda()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = m(toks)
        m(toks)
    assert sum("forward-only" in str(x.message) for x in w) == 1
    assert out["logits"].grad_fn is None
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m2 = esm.ESM2(L, E, H).eval().cuda()
        with torch.no_grad():
            m2(toks)
    assert not any("forward-only" in str(x.message) for x in w)


def test_extract_duplicate_layers_and_model_dtype_means(tmp_path):
    """--repr_layers -1 L names the same layer twice: handled like the reference's dict comprehension;
    a .half() model saves fp16 mean_representations as the reference does."""
    from esm_amd.extract import extract, make_embed_fn

    L, E, H = 2, 128, 2
    model = esm.ESM2(L, E, H).eval()
    model.load_state_dict(synth_esm2_state_dict(L, E, H, seed=7))
    model = model.half().cuda()
    seqs = ["MKTVRQERLK", "KALTARQQEVFDLIRD", "A"]
    ds = esm.FastaBatchedDataset(["a", "b", "c"], seqs)
    alphabet = esm.Alphabet.from_architecture("ESM-1b")
    means = extract(ds, alphabet, make_embed_fn(model), L, E, [-1, L], ["mean", "per_tok"], output_dir=tmp_path,
                    toks_per_batch=64, device=torch.device("cuda", 0), log=lambda s: None)
    assert sorted(means) == [L] and means[L].shape == (3, E)
    for label, s in zip("abc", seqs):
        r = torch.load(tmp_path / f"{label}.pt", weights_only=False)
        assert sorted(r["representations"]) == [L] and r["representations"][L].shape == (len(s), E)
        assert r["mean_representations"][L].dtype == torch.float16 and r["representations"][L].dtype == torch.float16
        want = r["representations"][L].float().mean(0)
        assert (r["mean_representations"][L].float() - want).abs().max().item() < 2e-3


def test_bench_self_spawn_initialises_rccl(tmp_path):
    """`python bench.py --spawn` = the N = 1 run through the same self-launch + init_process_group("nccl") +
    all-reduce path that `python bench.py --gpus N` takes: proves RCCL comes up on the box."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["PYTHONPATH"] = ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--spawn", "--steps", "2", "--warmup", "1",
                          "--batch", "4", "--no-cpu-baseline"], capture_output=True, text=True, env=env,
                         cwd=str(tmp_path), timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == 1 and r["collective_backend"] == "nccl" and r["value"] > 0
    assert r["roofline"]["bound"] == "mfma" and r["roofline_hbm"]["bound"] == "hbm"
    assert r["host_cores"] == os.cpu_count() and r["e2e_with_d2h"]["value"] > 0


def test_extract_cli_refuses_nogpu_and_msa(tmp_path):
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-m", "esm_amd.extract", "x.pt", "y.fasta", str(tmp_path), "--include", "mean",
                          "--nogpu"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert out.returncode != 0 and "--nogpu is not available" in (out.stderr + out.stdout)
