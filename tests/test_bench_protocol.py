"""bench.py's launch + timing protocol at world_size 2 on CPU (gloo): rank / world handling from the
torch.distributed.run environment, barrier-bracketed timed region, MAX over ranks, ONE JSON line from rank 0.
The step is a stub (5 ms x (rank + 1)); the real step needs an MI355X (tests -m gpu, bench.py itself)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_protocol(tmp_path):
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4",
           "--warmup", "1", "--protocol-test"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout  # rank 0 only
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 4 and r["warmup"] == 1 and r["scaling"] == "weak"
    # rank 1's steps take 10 ms: the MAX over ranks is reported, not rank 0's own 5 ms
    assert 9.5 <= r["ms_per_step"] < 1000, r  # lower bound is the point; the upper one only guards nonsense
    assert abs(r["value"] - 2 * 64 * 1022 * 4 / (r["ms_per_step"] * 4e-3)) / r["value"] < 1e-3
    # every rank's OWN time is in the line as well: rank 0 ~5 ms per step, rank 1 ~10 ms (the straggler is visible)
    own = r["per_rank_ms_per_step"]
    assert len(own) == 2 and 4.5 <= own[0] < own[1] and 9.5 <= own[1] <= r["ms_per_step"] + 1e-3, r


def test_self_spawned_two_rank_protocol(tmp_path):
    """`python bench.py --gpus 2` from a plain shell (no external launcher): the script re-executes itself as two
    ranks (esm_amd/launch.py), which is the form the driver's N > 1 runs may take."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["PYTHONPATH"] = ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--protocol-test"], capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["backend"] == "gloo" and r["launched_by"] == "self-spawned"
    assert 9.5 <= r["ms_per_step"] < 1000, r


def test_self_spawned_single_rank_goes_through_the_process_group(tmp_path):
    """--spawn: the N = 1 run takes the same launcher + init_process_group + all-reduce path as N > 1."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["PYTHONPATH"] = ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--spawn", "--steps", "2", "--warmup", "0",
                          "--protocol-test"], capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert r["n_gpus"] == 1 and r["backend"] == "gloo" and r["launched_by"] == "self-spawned"


def test_world_size_mismatch_is_refused(tmp_path):
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29543", os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1",
           "--warmup", "0", "--protocol-test"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=300)
    assert out.returncode != 0 and "does not match --gpus" in (out.stderr + out.stdout)


def test_single_process_protocol():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "0",
                          "--protocol-test"], capture_output=True, text=True, env=dict(os.environ, PYTHONPATH=ROOT),
                         timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert r["n_gpus"] == 1 and 4.5 <= r["ms_per_step"] < 1000
    assert len(r["per_rank_ms_per_step"]) == 1 and r["per_rank_ms_per_step"][0] <= r["ms_per_step"] + 1e-3


def test_argmax_report():
    import importlib.util

    import torch

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ref = torch.tensor([[[5.0, 1.0, 0.0], [2.0, 2.0005, 0.0], [0.0, 3.0, 2.9]]])
    got = ref.clone()
    got[0, 1] = torch.tensor([2.001, 2.0, 0.0])   # near-tie flips: not a decided position
    got[0, 0, 0] += 0.002
    r = bench.argmax_report(got, ref)
    assert abs(r["logits_argmax_agreement"] - 2 / 3) < 1e-6
    assert r["decided_positions"] == 2 and r["argmax_agreement_where_decided"] == 1.0
    assert abs(r["logits_max_abs_diff"] - 0.002) < 1e-6


def test_operand_floor_report(monkeypatch):
    """The `parity.operand_floor_same_inputs` block of the bench line: the oracle with operand rounding injected, on
    the CPU sample; fp16 by default, bf16 when the bench runs bf16 operands; an error never costs the line."""
    import importlib.util
    import sys

    import torch

    sys.path.insert(0, ROOT)
    from esm_amd.synth import synth_esm2_state_dict, synth_tokens
    from oracle.esm2_oracle import esm2_forward

    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    L, E, H = 3, 128, 2
    sd = synth_esm2_state_dict(L, E, H, seed=1)
    toks = synth_tokens(2, 48, seed=2)
    ref = esm2_forward(sd, toks, L, H, repr_layers=[L])["representations"][L].double()
    monkeypatch.delenv("ESM_AMD_OPERAND", raising=False)
    f16 = bench.operand_floor_report(sd, toks, L, H, ref)
    assert 2e-5 < f16["rel_l2_repr_diff_vs_cpu"] < 2e-3 and f16["rel_repr_diff_vs_cpu"] > 0 and "f16" in f16["what"]
    monkeypatch.setenv("ESM_AMD_OPERAND", "bf16")
    b16 = bench.operand_floor_report(sd, toks, L, H, ref)
    assert b16["rel_l2_repr_diff_vs_cpu"] > 4 * f16["rel_l2_repr_diff_vs_cpu"] and "bf16" in b16["what"]
    assert "error" in bench.operand_floor_report(sd, toks, L + 1, H, ref)  # a missing layer: reported, not raised


def test_secondary_workloads_glue(monkeypatch):
    """`secondary_workloads` (the other BASELINE configs appended to the default bench line): child runs of this same
    script, one JSON line parsed per workload, time limits and failures reported instead of raised.  The children
    run as --protocol-test here (CPU stubs); launcher variables of the parent must not leak into them."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod3", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setenv("PYTHONPATH", ROOT)
    monkeypatch.setenv("RANK", "0")  # as if the parent had been started by a launcher with one rank
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("MASTER_PORT", "1")
    import time as _t

    out = bench.secondary_workloads(extra=["--protocol-test"], budget_end=_t.perf_counter() + 1000)
    assert list(out) == ["msa1b", "extract_650m", "esm2_3b_contacts", "esm2_650m_b4", "esm2_650m_f16x2a", "esm2_3b_contacts_f16x3", "esm2_650m_sharp",
                         "esm2_650m_plain"]
    # the plain lines switch the LayerNorm fold off, the others leave the library default (on since round 5)
    assert out["esm2_650m_plain"]["config"]["ln_fold"] == "0"
    assert out["esm2_650m_b4"]["config"]["ln_fold"] != "0"
    for name, r in out.items():
        assert "error" not in r, (name, r)
        assert r["metric"].startswith("protocol-test") and r["ms_per_step"] >= 4.5 and r["wall_s"] > 0
    assert out["esm2_3b_contacts"]["steps"] == 4 and out["esm2_3b_contacts_f16x3"]["steps"] == 2 and out["extract_650m"]["steps"] == 24 and out["extract_650m"]["warmup"] == 2
    # a child that fails or overruns its limit is an entry with "error", not an exception
    monkeypatch.setattr(bench, "SECONDARY", [("bad_flag", ["--no-such-flag"], 60), ("too_slow", ["--steps", "400"], 1)])
    monkeypatch.setattr(bench, "SECONDARY_MIN_S", 0.5)
    import time

    out = bench.secondary_workloads(extra=["--protocol-test"], budget_end=time.perf_counter() + 1000)
    assert "error" in out["bad_flag"] and "exit code" in out["bad_flag"]["error"]
    assert "error" in out["too_slow"] and "Timeout" in out["too_slow"]["error"]
    # and once the default run's time budget is spent the remaining children are skipped, not started
    out = bench.secondary_workloads(extra=["--protocol-test"], budget_end=time.perf_counter() + 0.2)
    assert all("skipped" in r for r in out.values())


def test_skip_param_init_then_strict_load_gives_the_state_dict():
    """bench.py constructs its models without the constructors' random fill and then loads strictly: every parameter
    and buffer must come out equal to the synthetic state dict (ESM-2 and MSA Transformer), and the patch must be
    gone afterwards."""
    import argparse
    import importlib.util
    import sys

    import torch

    sys.path.insert(0, ROOT)
    import esm
    from esm_amd.synth import skip_param_init, synth_esm2_state_dict, synth_msa_state_dict

    before = torch.nn.Linear.reset_parameters
    L, E, H = 2, 128, 2
    sd = synth_esm2_state_dict(L, E, H, seed=3)
    with skip_param_init():
        m = esm.ESM2(L, E, H).eval()
    m.load_state_dict(sd)
    got = m.state_dict()
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    ns = argparse.Namespace(layers=2, embed_dim=64, ffn_embed_dim=128, attention_heads=2, dropout=0.1,
                            attention_dropout=0.1, activation_dropout=0.1, max_positions=1024, embed_positions_msa=True,
                            embed_positions_msa_dim=64, max_tokens=2 ** 14, max_tokens_per_msa=2 ** 14)
    sdm = synth_msa_state_dict(2, 64, 2, 128, seed=4)
    with skip_param_init():
        mm = esm.MSATransformer(ns, esm.Alphabet.from_architecture("msa_transformer")).eval()
    mm.load_state_dict(sdm)
    gotm = mm.state_dict()
    assert set(gotm) == set(sdm) and all(torch.equal(gotm[k], sdm[k]) for k in sdm)
    assert torch.nn.Linear.reset_parameters is before
    lin = torch.nn.Linear(8, 8)
    assert lin.weight.abs().sum() > 0  # constructors fill again outside the context


def test_pmc_traffic_is_gated_by_the_library_hash(tmp_path, monkeypatch):
    """roofline.traffic comes from a committed PMC summary only when that summary was taken on exactly this (workload,
    per-GPU batch, LayerNorm-fold mode) AND on the library that is running; anything else is null with the reason
    (VERDICT r4 Weak-6: the B = 4 lines used to report the B = 64 launch's bytes)."""
    sys.path.insert(0, ROOT)
    import bench

    monkeypatch.setattr(bench, "PMC_DIR", str(tmp_path))

    def write(workload, batch, fold, h, val):
        with open(bench.pmc_summary_path(workload, batch, fold), "w") as f:
            json.dump({"library_src_hash": h, "git_sha": "deadbeef", "workload": workload, "batch": batch, "ln_fold": fold,
                       "kernels": {"gemm_fc1_gelu": {"hbm_bytes_corrected": val}}}, f)

    write("esm2_650m", 64, True, "aaaa", 123.0)
    v, src = bench.pmc_traffic("gemm_fc1_gelu", "aaaa", "esm2_650m", 64, True)
    assert v == 123.0 and "aaaa" in src and "batch 64" in src and "deadbeef" in src
    assert bench.pmc_traffic("gemm_fc1_gelu", "aaaa", "esm2_650m", None, True)[0] == 123.0   # None = the workload's default batch
    # another library
    v, src = bench.pmc_traffic("gemm_fc1_gelu", "cccc", "esm2_650m", 64, True)
    assert v is None and "aaaa" in src and "cccc" in src
    # another batch / another fold mode: their own files, which do not exist -> null, never the B = 64 fold figure
    v, src = bench.pmc_traffic("gemm_fc1_gelu", "aaaa", "esm2_650m", 4, True)
    assert v is None and "missing" in src and "_b4" in src
    v, src = bench.pmc_traffic("gemm_fc1_gelu", "aaaa", "esm2_650m", 64, False)
    assert v is None and "_plain" in src
    v, src = bench.pmc_traffic("gemm_fc1_gelu", "aaaa", "esm2_650m", 4, False)
    assert v is None and "_b4_plain" in src
    # their own summaries are picked up, each for its own key only
    write("esm2_650m", 4, True, "aaaa", 7.0)
    write("esm2_650m", 4, False, "aaaa", 9.0)
    assert bench.pmc_traffic("gemm_fc1_gelu", "aaaa", "esm2_650m", 4, True)[0] == 7.0
    assert bench.pmc_traffic("gemm_fc1_gelu", "aaaa", "esm2_650m", 4, False)[0] == 9.0
    assert bench.pmc_traffic("gemm_fc1_gelu", "aaaa", "esm2_650m", 64, True)[0] == 123.0
    # a summary whose recorded key contradicts its file name is refused
    with open(bench.pmc_summary_path("esm2_650m", 8, True), "w") as f:
        json.dump({"library_src_hash": "aaaa", "workload": "esm2_650m", "batch": 64, "ln_fold": True,
                   "kernels": {"gemm_fc1_gelu": {"hbm_bytes_corrected": 1.0}}}, f)
    assert bench.pmc_traffic("gemm_fc1_gelu", "aaaa", "esm2_650m", 8, True)[0] is None
    # a class the summary does not hold
    assert bench.pmc_traffic("attention", "aaaa", "esm2_650m", 64, True)[0] is None
    # the committed summaries of this round describe themselves consistently
    import glob
    for path in glob.glob(os.path.join(ROOT, "profiles", f"{bench.PMC_ROUND}_pmc_summary_*.json")):
        with open(path) as f:
            sm = json.load(f)
        assert path == os.path.join(ROOT, "profiles", os.path.basename(bench.pmc_summary_path(sm["workload"], sm["batch"], sm["ln_fold"]))), path
        assert sm.get("library_src_hash") and sm.get("kernels")
