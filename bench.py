#!/usr/bin/env python
"""Benchmark of the hot path on MI355X.  Default: ESM-2 650M bulk embedding extraction, synthetic L=1022 batches.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--workload NAME]

``--gpus N`` from a plain shell re-executes this script as N ranks (one per GPU) under torch.distributed.run
(esm_amd/launch.py); started by a launcher already (RANK / WORLD_SIZE set) it joins it.  Ranks create the process
group with backend "nccl" (= RCCL over xGMI), check world size / backend and run one all-reduce before anything is
timed.  Sequences are independent, so ranks shard the batch with no data-path collective (weak scaling); the only
collectives are the timing barrier and the MAX over ranks.

Workloads (one "step" each; the default is BASELINE.json's headline configuration, the others make the remaining
BASELINE configs driver-measurable):
  esm2_650m         model(tokens [B,1024], repr_layers=[33])            — the call scripts/extract.py:95 makes
  esm2_3b_contacts  model.predict_contacts(tokens [B,1024]) at 3B dims   — config 3
  msa1b             model(tokens [1,128,513], repr_layers=[12])          — config 5 (MSA Transformer axial path)
  extract_650m      the extraction driver end to end: FASTA strings -> tokens -> forward -> device->host ->
                    per-sequence .pt files written (--include mean per_tok), SURVEY §8 f-3

Prints ONE JSON line (rank 0): metric residues/sec (whole job) plus
  roofline            dominant kernel class vs the fp16/bf16 MFMA roof (2.5 PFLOP/s dense), HIP events on the launch
                      stream in extra profiled steps after the timed region; `traffic` = HBM bytes per launch from the
                      committed rocprofv3 PMC pass — only when that pass was taken on THIS build of the library
                      (source hash in esmk_version()), else null;
  roofline_attention  the attention kernel against the same roof;  roofline_hbm: LayerNorm against 8 TB/s HBM;
  e2e_with_d2h        SURVEY §8 d figure (ii): forward + device->host copy of representations[33], overlapped;
  cpu_baseline        the oracle (CPU restatement of the reference, oracle/esm2_oracle.py) timed on the host cores
                      at B=1 (1 warm-up + 3 timed, median) and B=4, and the parity of the GPU outputs against it
                      (parity.operand_floor_same_inputs: the same sample through the oracle with fp16 rounding at
                      every operand point — the floor of any 16-bit-operand engine);
  per_rank_ms_per_step  every rank's own time before the closing barrier;
  secondary_workloads   default run (esm2_650m, N = 1, CPU baseline on) only: the msa1b, extract_650m and
                      esm2_3b_contacts lines of this same script, run as child processes inside a 3-minute budget
                      (--no-secondary skips them), so that one driver run records every BASELINE configuration.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0    # dense fp16/bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0        # spec; 6.29 TB/s is the measured copy ceiling
HBM_ACHIEVABLE_GBS = 6300.0
DEFAULT_BATCH = {"esm2_650m": 64, "esm2_3b_contacts": 32, "msa1b": 1, "extract_650m": 64}
PMC_DIR = os.path.join(ROOT, "profiles")
PMC_ROUND = "r6"


def pmc_summary_path(workload, batch, ln_fold):
    """profiles/r6_pmc_summary_<workload>[_b<B>][_plain].json: one file per (workload, per-GPU batch, LayerNorm-fold mode)
    — the default batch and the default (fold) mode carry no suffix.  tools/profile_bench.sh writes them."""
    tag = workload
    if batch not in (None, 0, DEFAULT_BATCH.get(workload)):
        tag += f"_b{batch}"
    if ln_fold is False:
        tag += "_plain"
    return os.path.join(PMC_DIR, f"{PMC_ROUND}_pmc_summary_{tag}.json")


def argmax_report(logits, ref_logits):
    """Token-argmax agreement with the CPU path.  Random-init weights give near-tied logits, so the raw agreement
    counts coin flips; `decided` restricts it to positions whose top-2 margin in the reference exceeds twice the
    largest logit difference — there the argmax must be identical (north star: token argmax bit-exact)."""
    err = (logits - ref_logits).abs().max().item()
    top2 = ref_logits.topk(2, dim=-1).values
    decided = (top2[..., 0] - top2[..., 1]) > 2 * err
    same = logits.argmax(-1) == ref_logits.argmax(-1)
    n = logits[..., 0].numel()
    return {"logits_argmax_agreement": same.float().mean().item(),
            "logits_max_abs_diff": err,
            "logits_rel_diff": err / max(ref_logits.abs().max().item(), 1e-30),
            **logits_l2_report(logits, ref_logits),
            "positions": n,
            "decided_positions": int(decided.sum().item()),
            "undecided_fraction": round(1.0 - decided.sum().item() / n, 5),
            "argmax_agreement_where_decided": same[decided].float().mean().item() if bool(decided.any()) else None}


def logits_l2_report(logits, ref_logits):
    """L2 error of the logits, and its split into the row-independent part (the mean error vector over all positions: ONE
    draw of the weight-rounding bias per model and operand form) and the token-dependent rest."""
    d = (logits.double() - ref_logits.double()).reshape(-1, ref_logits.shape[-1])
    c = d.mean(0, keepdim=True)
    n = max(ref_logits.double().norm().item(), 1e-30)
    return {"logits_rel_l2_diff": d.norm().item() / n, "logits_rel_l2_row_common": c.norm().item() * d.shape[0] ** 0.5 / n,
            "logits_rel_l2_token_dependent": (d - c).norm().item() / n}


PER_RANK_MS = []  # each rank's own ms per step of the last timed region (before it waits for the others)


def timed_steps(step, steps, warmup, sync_all, dist, dev):
    """The contract's timed region: `warmup` untimed steps, then exactly `steps` steps bracketed by
    synchronise + barrier on both sides; returns the MAX elapsed seconds over ranks.  Every rank's own time (its K
    steps done, before the closing barrier) is gathered into PER_RANK_MS: a straggler shows up there, not only in
    the maximum."""
    for _ in range(warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)  # sync_all() starts with the same call: no extra wait inside the timed region
    own = time.perf_counter() - t0
    sync_all()
    elapsed = time.perf_counter() - t0
    own_all = [own]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([own], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(every, mine)
        own_all = [float(v.item()) for v in every]
    PER_RANK_MS[:] = [round(1e3 * o / steps, 3) for o in own_all]
    return elapsed


def finish(dist):
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def protocol_test(args):
    """Same launch (self-spawn or external launcher), rank / world handling, barriers, MAX-over-ranks and single JSON
    line as the real run, on CPU with the gloo backend and a stub step that takes 5 ms x (rank + 1): there is no GPU
    in the build container and the 8-GPU run belongs to the driver, so this is what keeps the N > 1 path honest."""
    from esm_amd.launch import init_ranks

    dist, rank, world, _ = init_ranks(args.gpus, "gloo")
    dev = torch.device("cpu")

    def sync_all():
        if dist is not None:
            dist.barrier()

    elapsed = timed_steps(lambda: time.sleep(0.005 * (rank + 1)), args.steps, args.warmup, sync_all, dist, dev)
    if rank == 0:
        print(json.dumps({"metric": "protocol-test (not a measurement)", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
                          "value": round(world * args.batch * args.seq_len * args.steps / elapsed, 1),
                          "scaling": "weak", "per_rank_ms_per_step": list(PER_RANK_MS),
                          "config": {"workload": "protocol-test", "ln_fold": os.environ.get("ESM_AMD_LN_FOLD", "library default (on)")},
                          "backend": dist.get_backend() if dist is not None else None,
                          "launched_by": os.environ.get("ESM_AMD_BENCH_LAUNCH", "external-or-single")}), flush=True)
    finish(dist)


# ---------------------------------------------------------------------------------------------------------------
def operand_name():
    env = os.environ.get("ESM_AMD_OPERAND", "").lower()
    return ("bf16" if env in ("bf16", "bfloat16") else "f16x2" if env in ("f16x2", "fp16x2")
            else "f16x2a" if env in ("f16x2a", "fp16x2a") else "f16x2v" if env in ("f16x2v", "fp16x2v")
            else "f16x3" if env in ("f16x3", "fp16x3") else "f16")


def library_build():
    from esm_amd import _native
    from esm_amd.build import library_hash

    return {"version": _native.lib.esmk_version().decode(), "src_hash": library_hash(_native.LIB_PATH)}


def class_table(prof, prof_steps):
    return {
        e["name"]: {
            "ms_per_step": round(e["ms"] / prof_steps, 4),
            "launches_per_step": e["launches"] // prof_steps,
            "tflops": round(e["flops"] / (e["ms"] * 1e-3) / 1e12, 1) if e["flops"] and e["ms"] > 0 else None,
            "gbs": round(e["bytes"] / (e["ms"] * 1e-3) / 1e9, 1) if e["ms"] > 0 else None,
        }
        for e in prof
    }


# bench.py's class names -> the classes tools/pmc_summary.py can tell apart by kernel name (the q/k and the v
# projection are two launches of one class here; the q/k launch is the larger one)
PMC_CLASS = {"gemm_qkv_rope": "gemm_qkv_rope(qk)"}


def pmc_traffic(kernel_class, src_hash, workload="esm2_650m", batch=None, ln_fold=None):
    """HBM bytes per launch of `kernel_class` from the committed rocprofv3 PMC passes (tools/profile_bench.sh ->
    profiles/r6_pmc_summary_*.json; FETCH_SIZE doubled as the microarch guide prescribes for gfx950).  A figure is
    reported ONLY when a summary exists for exactly this (workload, per-GPU batch, LayerNorm-fold mode) AND records the
    source hash of the library that is running — anything else is null with the reason (never a stale number, never
    another launch shape's number: VERDICT r4 Weak-6)."""
    batch = batch or DEFAULT_BATCH.get(workload)
    path = pmc_summary_path(workload, batch, ln_fold)
    rel = os.path.relpath(path, ROOT)
    try:
        with open(path) as f:
            s = json.load(f)
    except OSError:
        return None, f"no PMC summary for workload {workload}, batch {batch}, ln_fold {ln_fold} ({rel} missing)"
    try:
        if s.get("library_src_hash") != src_hash:
            return None, f"{rel} is from build {s.get('library_src_hash')}, this is {src_hash}"
        if s.get("workload") != workload or s.get("batch") != batch or bool(s.get("ln_fold")) != bool(ln_fold):
            return None, (f"{rel} was taken at workload {s.get('workload')}, batch {s.get('batch')}, ln_fold {s.get('ln_fold')}; "
                          f"this run is {workload}, batch {batch}, ln_fold {ln_fold}")
        return (s["kernels"][kernel_class]["hbm_bytes_corrected"],
                f"{rel} (rocprofv3 PMC passes of library {s['library_src_hash']}, batch {batch}, ln_fold {bool(ln_fold)}"
                + (f", commit {s['git_sha']}" if s.get("git_sha") else "") + ")")
    except Exception as e:  # missing class / malformed file
        return None, f"{rel}: {type(e).__name__}: {e}"


def mfma_roofline(prof, name=None):
    cand = [e for e in prof if e["flops"] > 0 and e["launches"] > 0]
    dom = max(cand, key=lambda e: e["ms"]) if name is None else next(e for e in cand if e["name"] == name)
    ms = dom["ms"] / dom["launches"]
    achieved = dom["flops"] / dom["launches"] / (ms * 1e-3) / 1e12
    return dom, {
        "kernel": dom["name"], "bound": "mfma", "achieved": round(achieved, 1), "peak": MFMA_PEAK_TFLOPS,
        "unit": "TFLOP/s", "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": None,
        "avg_launch_ms": round(ms, 4), "algorithmic_flops_per_launch": dom["flops"] / dom["launches"],
    }


def hbm_roofline(prof, name="layernorm"):
    e = next((e for e in prof if e["name"] == name and e["launches"] > 0), None)
    if e is None:
        return None
    ms = e["ms"] / e["launches"]
    gbs = e["bytes"] / e["launches"] / (ms * 1e-3) / 1e9
    return {"kernel": name, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4), "frac_of_copy_ceiling": round(gbs / HBM_ACHIEVABLE_GBS, 4),
            "avg_launch_ms": round(ms, 4), "algorithmic_bytes_per_launch": e["bytes"] / e["launches"], "traffic": None}


def profile_steps(model, step, prof_steps=2):
    model.profile_begin()
    for _ in range(prof_steps):
        step()
    return model.profile_end(), prof_steps


def fold_state(model):
    """'on' / 'off' as the ENGINE reports it (esmk_ln_fold_enabled), plus who decided."""
    act = getattr(model, "ln_fold_active", lambda: None)()
    env = os.environ.get("ESM_AMD_LN_FOLD")
    return {"active": act, "label": ("on" if act else "off" if act is not None else "n/a (no fold in this engine)")
                                    + (f" (ESM_AMD_LN_FOLD={env})" if env is not None else " (library default)")}


def base_result(args, world, metric, value, elapsed, workload, extra_cfg, model=None):
    fs = fold_state(model) if model is not None else {"active": None, "label": "n/a"}
    extra_cfg = dict(extra_cfg, ln_fold=fs["label"])
    return {
        "metric": metric, "value": round(value, 1), "unit": "residues/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
        "scaling": getattr(args, "scaling", "weak"), "vs_baseline": None, "dtype": operand_name(), "data": "synthetic",
        "config": {"workload": workload, "sharding": f"dp{world} (no data-path collective)", **extra_cfg},
        "host_cores": os.cpu_count(), "per_rank_ms_per_step": list(PER_RANK_MS),
    }


def cpu_threads():
    # MKL on all 256 hardware threads of the GPU host is 30x SLOWER than on 32 (measured: 11 vs 354 residues/s,
    # tools/cpu_threads_probe.py); ESM_AMD_CPU_THREADS overrides
    n = min(os.cpu_count() or 1, int(os.environ.get("ESM_AMD_CPU_THREADS", "32")))
    torch.set_num_threads(n)
    return n


# ---------------------------------------------------------------------------------------------------------------
def run_esm2_650m(args, dist, rank, world, dev):
    import esm
    from esm_amd.synth import ESM2_DIMS, skip_param_init, synth_esm2_state_dict, synth_tokens

    MODEL = "esm2_t33_650M_UR50D"
    FLOP_PER_RESIDUE = 1.4769e9  # SURVEY.md §8 d: 1.509 TFLOP per 1024-token sequence / 1022 residues
    L, E, H = ESM2_DIMS[MODEL]
    sd = synth_esm2_state_dict(L, E, H, seed=0, qk_gain=args.qk_gain, ln_gamma_std=args.ln_gamma_std)  # identical replica on every rank
    with skip_param_init():
        model = esm.ESM2(L, E, H).eval()
    model.load_state_dict(sd)
    model = model.to(dev)
    if args.scaling == "strong":  # the node's work per step is fixed: 512 sequences (64 per GPU at N = 8)
        total = args.batch or 512
        assert total % world == 0, f"--scaling strong: --batch {total} must be a multiple of --gpus {world}"
        batch = total // world
    else:
        batch = args.batch or 64
    toks = synth_tokens(batch, args.seq_len, seed=1 + rank).to(dev)  # each rank its own shard
    residues_per_step = batch * args.seq_len

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    step = lambda: model(toks, repr_layers=[L])
    with torch.no_grad():
        elapsed = timed_steps(step, args.steps, args.warmup, sync_all, dist, dev)
        prof, prof_steps = profile_steps(model, step)  # separate steps: the timed region is unperturbed

        # figure (ii): every step's representations[33] also goes to pinned host memory on a side stream while the
        # next forward runs (what the extraction driver does, scripts/extract.py:97-100)
        d2h_steps = max(2, min(args.steps, 10))
        copy_stream = torch.cuda.Stream(dev)
        host = [torch.empty((batch, args.seq_len + 2, E), dtype=torch.float32, pin_memory=True) for _ in range(2)]
        done = [torch.cuda.Event() for _ in range(2)]
        sync_all()
        t0 = time.perf_counter()
        for i in range(d2h_steps):
            out = model(toks, repr_layers=[L])["representations"][L]
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(dev))
            if i >= 2:
                done[i % 2].synchronize()  # the host buffer is free again
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(ready)
                host[i % 2].copy_(out, non_blocking=True)
                out.record_stream(copy_stream)
                done[i % 2].record(copy_stream)
        copy_stream.synchronize()
        sync_all()
        d2h_elapsed = time.perf_counter() - t0

    value = world * residues_per_step * args.steps / elapsed
    if rank != 0:
        return None
    result = base_result(
        args, world, "residues/sec (whole node) ESM-2 650M L=1022 bulk extract", value, elapsed,
        f"{MODEL} forward (repr_layers=[33] + logits), synthetic tokens [B,{args.seq_len + 2}], random-init weights "
        "of the 650M architecture", {"batch_per_gpu": batch, "seq_len": args.seq_len,
                                     "synthetic_weights": {"qk_gain": args.qk_gain, "ln_gamma_std": args.ln_gamma_std}}, model)
    build = library_build()
    fold_on = model.ln_fold_active()
    traffic = lambda cls: pmc_traffic(cls, build["src_hash"], "esm2_650m", batch, fold_on)
    dom, roof = mfma_roofline(prof)
    roof["traffic"], roof["traffic_source"] = traffic(dom["name"])
    result["e2e_mfma_frac_per_gpu"] = round(value / world * FLOP_PER_RESIDUE / (MFMA_PEAK_TFLOPS * 1e12), 4)
    result["roofline"] = roof
    _, ra = mfma_roofline(prof, "attention")
    ra["traffic"], _ = traffic("attention")
    result["roofline_attention"] = ra
    # the HBM-bound kernel of the step: the standalone LayerNorm passes (plain mode: 66 per step; with the fold only the
    # row-statistics entry, the per-layer finalize launches and the final LayerNorm are left in this class)
    rh = hbm_roofline(prof)
    if rh is not None:
        rh["traffic"], _ = traffic("layernorm")
        rh["note"] = ("LayerNorm fold on: this class holds rowstats + ln_finalize + the final LayerNorm only" if fold_on
                      else "standalone LayerNorm passes, 2 per layer")
    result["roofline_hbm"] = rh
    result["kernel_classes"] = class_table(prof, prof_steps)
    result["profiled_ms_per_step"] = round(sum(e["ms"] for e in prof) / prof_steps, 3)
    result["e2e_with_d2h"] = {
        "value": round(residues_per_step * d2h_steps / d2h_elapsed, 1), "unit": "residues/s (this rank)",
        "steps": d2h_steps, "what": "forward + device->host copy of representations[33] (fp32, "
                                    f"{batch * (args.seq_len + 2) * E * 4 / 1e6:.0f} MB per step) into pinned memory on a "
                                    "side stream, overlapped with the next forward"}
    result["library"] = build
    if world == 1 and args.also_qk_gain > 0:
        # data sensitivity, second point: the same run once more on weights with this q / k gain (timing only — random
        # weights with maps this sharp make a chaotic network: the operand floor itself is O(1) there, parity is undefined)
        try:
            model.load_state_dict(synth_esm2_state_dict(L, E, H, seed=0, qk_gain=args.also_qk_gain, ln_gamma_std=args.ln_gamma_std))
            with torch.no_grad():
                el2 = timed_steps(step, max(4, args.steps // 2), 2, sync_all, dist, dev)
            n2 = max(4, args.steps // 2)
            result["also_qk_gain"] = {"qk_gain": args.also_qk_gain, "ln_gamma_std": args.ln_gamma_std, "steps": n2,
                                      "value": round(residues_per_step * n2 / el2, 1), "ms_per_step": round(1e3 * el2 / n2, 3),
                                      "parity": "not taken: at this gain the fp16-operand FLOOR of the synthetic network is 0.47 (chaotic), DESIGN.md I.5"}
            model.load_state_dict(sd)  # back to the line's own weights for the parity sample below
        except Exception as e:
            result["also_qk_gain"] = {"error": f"{type(e).__name__}: {e}"[:200]}

    if world == 1 and args.parity_ref and os.path.exists(args.parity_ref):
        # a secondary line of the default run: parity of THIS engine configuration on the parent's 4 sample sequences,
        # against the oracle outputs the parent computed once (bench.py secondary_workloads)
        try:
            pr = torch.load(args.parity_ref)
            with torch.no_grad():
                got = model(pr["tokens"].to(dev), repr_layers=[L])
            r_gpu, r_ref = got["representations"][L].cpu().double(), pr["repr"].double()
            max_abs = (r_gpu - r_ref).abs().max().item()
            result["parity"] = {"max_abs_repr_diff_vs_cpu": max_abs, "rel_repr_diff_vs_cpu": max_abs / r_ref.abs().max().item(),
                                "rel_l2_repr_diff_vs_cpu": ((r_gpu - r_ref).norm() / r_ref.norm()).item(),
                                **argmax_report(got["logits"].float().cpu(), pr["logits"].float()), "sample_sequences": 4,
                                "reference": "fp32 oracle outputs of the parent run on the same 4 sequences"}
        except Exception as e:
            result["parity"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    if world == 1 and not args.no_cpu_baseline and args.quick_baseline:
        # a secondary line's own bounded sample: ONE sequence of the timed batch through the fp32 oracle (+ the operand floor)
        from oracle.esm2_oracle import esm2_forward

        ncores = cpu_threads()
        s1 = toks[:1].cpu()
        c0 = time.perf_counter()
        ref = esm2_forward(sd, s1, L, H, repr_layers=[L])
        t1 = time.perf_counter() - c0
        with torch.no_grad():
            got = model(toks[:1], repr_layers=[L])
        r_gpu, r_ref = got["representations"][L].cpu().double(), ref["representations"][L].double()
        max_abs = (r_gpu - r_ref).abs().max().item()
        result["cpu_baseline"] = {"value": round(args.seq_len / t1, 1), "unit": "residues/s", "cores": torch.get_num_threads(),
                                  "host_cores": os.cpu_count(), "kind": "port",
                                  "sample": f"fp32 oracle on {ncores} threads: 1 sequence of the timed batch (L={args.seq_len}), 1 forward, cold"}
        result["parity"] = {"max_abs_repr_diff_vs_cpu": max_abs, "rel_repr_diff_vs_cpu": max_abs / r_ref.abs().max().item(),
                            "rel_l2_repr_diff_vs_cpu": ((r_gpu - r_ref).norm() / r_ref.norm()).item(),
                            **argmax_report(got["logits"].float().cpu(), ref["logits"].float()), "sample_sequences": 1}
        result["parity"]["operand_floor_same_inputs"] = operand_floor_report(sd, s1, L, H, r_ref, ref["logits"], fold=bool(model.ln_fold_active()))
    elif world == 1 and not args.no_cpu_baseline:
        from oracle.esm2_oracle import esm2_forward

        ncores = cpu_threads()
        # SURVEY §8 d: B = 1 and B = 4, 1 warm-up + 3 timed forwards, median.  B = 4 costs ~12 s per forward on the
        # GPU host, so it is timed once after the warm caches of B = 1 (the sample stays inside ~30 s of CPU work).
        s1, s4 = toks[:1].cpu(), toks[:4].cpu()
        t1 = []
        for _ in range(4):
            c0 = time.perf_counter()
            esm2_forward(sd, s1, L, H, repr_layers=[L])
            t1.append(time.perf_counter() - c0)
        med1 = sorted(t1[1:])[1]
        c0 = time.perf_counter()
        ref = esm2_forward(sd, s4, L, H, repr_layers=[L])
        t4 = time.perf_counter() - c0
        with torch.no_grad():
            got = model(toks[:4], repr_layers=[L])
        r_gpu, r_ref = got["representations"][L].cpu().double(), ref["representations"][L].double()
        max_abs = (r_gpu - r_ref).abs().max().item()
        try:
            amax = argmax_report(got["logits"].float().cpu(), ref["logits"].float())
        except Exception as e:  # never lose the JSON line over a report detail
            amax = {"logits_argmax_agreement": None, "error": str(e)}
        result["cpu_baseline"] = {
            "value": round(4 * args.seq_len / t4, 1), "unit": "residues/s", "cores": torch.get_num_threads(),
            "host_cores": os.cpu_count(), "kind": "port",
            "sample": f"fp32 oracle on {ncores} threads: B=4 sequences of the timed batch (L={args.seq_len}), 1 forward "
                      f"after the B=1 runs; B=1: 1 warm-up + 3 timed, median",
            "b1_residues_per_s": round(args.seq_len / med1, 1), "b1_seconds": [round(t, 3) for t in t1],
            "b4_seconds": round(t4, 3),
        }
        result["parity"] = {
            "max_abs_repr_diff_vs_cpu": max_abs,
            "rel_repr_diff_vs_cpu": max_abs / r_ref.abs().max().item(),
            "rel_l2_repr_diff_vs_cpu": ((r_gpu - r_ref).norm() / r_ref.norm()).item(),
            **amax, "sample_sequences": 4,
        }
        result["parity"]["operand_floor_same_inputs"] = operand_floor_report(sd, s4, L, H, r_ref, ref["logits"], fold=bool(model.ln_fold_active()))
        if args.save_parity_ref:  # for the secondary 650M lines (other batch / fold settings): same sequences, same reference
            torch.save({"tokens": s4, "repr": ref["representations"][L], "logits": ref["logits"]}, args.save_parity_ref)
    return result


def operand_floor_report(sd, toks_cpu, L, H, r_ref, logits_ref=None, fold=False):
    """The CPU sample once more through the oracle with every MFMA operand (weights, GEMM inputs, q, k, v, P) rounded
    to the operand dtype: the accuracy floor of ANY 16-bit-operand engine on these inputs, to read the engine's own
    `parity` numbers against — in the FORM the engine ran in (`fold`: the LayerNorm-fold form of the
    LayerNorm -> Linear pairs, oracle "FOLD" injection; one model's two forms differ by a draw of the weight-rounding bias).
    Test infrastructure on the CPU leg only; never costs the JSON line."""
    try:
        from oracle.esm2_oracle import ALL_OPERANDS, esm2_forward

        odt = torch.bfloat16 if operand_name() == "bf16" else torch.float16
        # f16x2 (split weights): the floor of that mode keeps the weights exact; f16x2a: those of the attention projections;
        # both run the LM head on the fp32 MFMA path (no rounding there)
        kinds = [k for k in ALL_OPERANDS if not (operand_name() == "f16x2" and k == "W")] + (["FOLD"] if fold else [])
        if operand_name() in ("f16x2a", "f16x2v"):
            kinds += ["W!v", "W!o"] + (["W!qk"] if operand_name() == "f16x2a" else [])
        if operand_name() == "f16x3":  # weights and GEMM inputs both split: q / k, v and P are what is still rounded
            kinds = ["QK", "V", "P"]
        head = {"inject_head": None} if operand_name() in ("f16x2", "f16x2a", "f16x2v", "f16x3") else {}
        fl = esm2_forward(sd, toks_cpu, L, H, repr_layers=[L], inject=(frozenset(kinds), odt), **head)
        lg = fl["logits"].double()
        fl = fl["representations"][L].double()
        rep = {"rel_repr_diff_vs_cpu": ((fl - r_ref).abs().max() / r_ref.abs().max()).item(),
               "rel_l2_repr_diff_vs_cpu": ((fl - r_ref).norm() / r_ref.norm()).item(),
               "form": "ln_fold" if fold else "plain",
               "what": f"fp32 oracle with {operand_name()} rounding injected at every operand point, same sequences"}
        if logits_ref is not None:  # the floor of the logits as well: parity.logits_rel_diff is to be read against it
            lr = logits_ref.double()
            rep["logits_rel_diff"] = ((lg - lr).abs().max() / lr.abs().max()).item()
            rep["logits_argmax_agreement"] = (lg.argmax(-1) == lr.argmax(-1)).double().mean().item()
            rep.update(logits_l2_report(lg, lr))
        return rep
    except Exception as e:
        return {"error": str(e)}


# The other BASELINE configurations, so that the driver's default run records them too ("driver-visible numbers
# for everything but config 2").  Each is the same `--workload` run a user would start, as a child process with a
# time limit; nothing in here can cost the flagship line.
SECONDARY = [  # --quick-baseline: the children's own CPU-oracle sample (parity + cpu_baseline), sized for a few seconds
    ("msa1b", ["--workload", "msa1b", "--quick-baseline"], 120),
    # 24 batches: the timed region ends when the writer threads have closed the LAST file, i.e. it contains one un-overlapped
    # drain (device->host + 64 files of 5.2 MB, ~80 ms) whatever its length — at 8 batches that was 10 % of the figure
    ("extract_650m", ["--workload", "extract_650m", "--steps", "24", "--warmup", "2", "--quick-baseline"], 150),
    ("esm2_3b_contacts", ["--workload", "esm2_3b_contacts", "--steps", "4", "--quick-baseline"], 200),
    # the reference script's default token budget (scripts/extract.py:36: 4096 tokens = 4 sequences of L = 1022), default mode
    ("esm2_650m_b4", ["--workload", "esm2_650m", "--batch", "4", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-secondary",
                      "--parity-ref", "{PARITY_REF}"], 60),
    # precision mode f16x2a (round 6): split weights on the attention projections only — representations AND logits inside
    # 1e-3 (representations in both norms, logits in L2) at 1.21 x the plain step; same 4 sequences, same fp32 reference as the
    # headline line
    ("esm2_650m_f16x2a", ["--workload", "esm2_650m", "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--no-secondary",
                          "--operand", "f16x2a", "--parity-ref", "{PARITY_REF}"], 90),
    # precision mode f16x3 (round 6): weights AND GEMM inputs split — the mode in which EVERY output, contact logits included, is
    # inside 1e-3 of the reference; config 3 with its T = 258 parity sample (representations, logits, contact logits)
    ("esm2_3b_contacts_f16x3", ["--workload", "esm2_3b_contacts", "--steps", "2", "--warmup", "1", "--quick-baseline",
                                "--operand", "f16x3"], 120),
    # data sensitivity: under the power cap the rates depend on the operand statistics — the headline
    # configuration once more on synthetic weights with sharper attention and wider LayerNorm gains (std 0.1).  qk_gain 2.5
    # (mean attention-row maximum 0.18 instead of 0.05) still is a non-chaotic network: value + its own parity sample;
    # qk_gain 4 (row maximum 0.68; what VERDICT r5 named) is timed in the same child, without parity (its floor is 0.47)
    ("esm2_650m_sharp", ["--workload", "esm2_650m", "--steps", "10", "--warmup", "3", "--no-secondary", "--quick-baseline",
                         "--qk-gain", "2.5", "--ln-gamma-std", "0.1", "--also-qk-gain", "4"], 100),
    # the headline configuration WITHOUT the LayerNorm fold (round 4's default path), same run, same box
    # (the B = 4 plain-mode line of round 5 left the default run in round 6: eight children, ~4 min on a slow host)
    ("esm2_650m_plain", ["--workload", "esm2_650m", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-secondary",
                         "--ln-fold", "0", "--parity-ref", "{PARITY_REF}"], 90),
]
T_PROCESS_START = time.perf_counter()
SECONDARY_BUDGET_S = 270.0  # the default run, children included, ends within ~4.5 minutes of its start (eight children: ~165 s)
SECONDARY_MIN_S = 30.0      # a child is not started with less than this left


def secondary_workloads(extra=(), budget_end=None, parity_ref=""):
    import subprocess

    if budget_end is None:
        budget_end = T_PROCESS_START + SECONDARY_BUDGET_S

    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_PORT",
                        "MASTER_ADDR", "TORCHELASTIC_RUN_ID", "ESM_AMD_BENCH_LAUNCH")}
    keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "e2e_mfma_frac_per_gpu",
            "tflops_algorithmic", "config", "parity", "cpu_baseline", "also_qk_gain")
    out = {}
    for name, argv, limit in SECONDARY:
        t0 = time.perf_counter()
        limit = min(limit, budget_end - t0)
        if limit < SECONDARY_MIN_S:
            out[name] = {"skipped": "time budget of the default run spent (run it with --workload " + name + ")"}
            continue
        try:
            argv = [a.replace("{PARITY_REF}", parity_ref) for a in argv]
            p = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv + list(extra), capture_output=True,
                               text=True, timeout=limit, env=env)
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not lines:
                raise RuntimeError(f"exit code {p.returncode}: {p.stderr[-200:]}")
            r = json.loads(lines[-1])
            out[name] = {k: r[k] for k in keep if k in r}
            if isinstance(r.get("roofline"), dict):
                out[name]["roofline"] = {k: r["roofline"].get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac",
                                                                             "traffic", "traffic_source")}
        except Exception as e:  # a time limit, a crash, a malformed line: reported, never raised
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        out[name]["wall_s"] = round(time.perf_counter() - t0, 1)
    return out


def run_esm2_3b_contacts(args, dist, rank, world, dev):
    import esm
    from esm_amd.synth import ESM2_DIMS, skip_param_init, synth_esm2_state_dict, synth_tokens

    MODEL = "esm2_t36_3B_UR50D"
    L, E, H = ESM2_DIMS[MODEL]
    T = args.seq_len + 2
    flop_per_seq = L * (24.0 * E * E + 4.0 * T * E) * T  # the transformer stack (SURVEY §8 a); the contact pass
    sd = synth_esm2_state_dict(L, E, H, seed=2)          # recomputes QK^T once more (+2TE per token-layer)
    with skip_param_init():
        model = esm.ESM2(L, E, H).eval()
    model.load_state_dict(sd)
    model = model.to(dev)
    # B = 32: the smallest batch at which every GEMM of the 3B layer (N = 2560 / 5120 / 10240) has a tile count that is
    # a multiple of the 256 CUs (B = 16 leaves out-proj, v and fc2 at 2.5 rounds, paid as 3)
    batch = args.batch or 32
    toks = synth_tokens(batch, args.seq_len, seed=1 + rank).to(dev)

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    step = lambda: model.predict_contacts(toks)
    with torch.no_grad():
        elapsed = timed_steps(step, args.steps, args.warmup, sync_all, dist, dev)
        prof, prof_steps = profile_steps(model, step)
    value = world * batch * args.seq_len * args.steps / elapsed
    if rank != 0:
        return None
    result = base_result(
        args, world, "residues/sec ESM-2 3B L=1022 contact prediction (predict_contacts)", value, elapsed,
        f"{MODEL} predict_contacts(tokens [B,{T}]): 36-layer forward + contact head without the [B,L,H,T,T] tensor, "
        "random-init weights of the 3B architecture", {"batch_per_gpu": batch, "seq_len": args.seq_len}, model)
    result["e2e_mfma_frac_per_gpu"] = round(value / world / args.seq_len * flop_per_seq / (MFMA_PEAK_TFLOPS * 1e12), 4)
    dom, result["roofline"] = mfma_roofline(prof)
    result["roofline_hbm"] = hbm_roofline(prof)
    result["kernel_classes"] = class_table(prof, prof_steps)
    result["library"] = library_build()
    result["roofline"]["traffic"], result["roofline"]["traffic_source"] = pmc_traffic(
        PMC_CLASS.get(dom["name"], dom["name"]), result["library"]["src_hash"], "esm2_3b_contacts", batch, model.ln_fold_active())
    if world == 1 and not args.no_cpu_baseline:
        from oracle.esm2_oracle import ALL_OPERANDS, esm2_forward

        ncores = cpu_threads()
        # T = 258 — the length of the committed config-3 fixture (tests/golden/large_esm2_3b_T258.pt), also under
        # --quick-baseline: a 128-residue sample read ~15 % friendlier than the fixture.  The oracle
        # materialises [1440,T,T] four times: ~1.5 GB, seconds on 32 threads.
        n_small = 256
        small = synth_tokens(1, n_small, seed=5)
        c0 = time.perf_counter()
        ref = esm2_forward(sd, small, L, H, repr_layers=[L], return_contacts=True)
        t_cpu = time.perf_counter() - c0
        with torch.no_grad():
            out = model(small.to(dev), repr_layers=[L])
            got = model.predict_contacts(small.to(dev)).cpu()
        lg = lambda t: torch.logit(t.double().clamp(1e-12, 1 - 1e-12))
        z, zr = lg(got), lg(ref["contacts"])
        r_gpu, r_ref = out["representations"][L].cpu().double(), ref["representations"][L].double()
        result["cpu_baseline"] = {"value": round(n_small / t_cpu, 1), "unit": "residues/s", "cores": ncores,
                                  "host_cores": os.cpu_count(), "kind": "port",
                                  "sample": f"one sequence of {n_small} residues (T = {n_small + 2}) through the fp32 oracle with contacts, 1 run"}
        result["parity"] = {"contacts_max_abs_prob_diff": (got - ref["contacts"]).abs().max().item(),
                            "contacts_max_abs_logit_diff": (z - zr).abs().max().item(),
                            "contacts_logit_diff_rel_to_range": ((z - zr).abs().max() / zr.abs().max()).item(),
                            "rel_repr_diff_vs_cpu": ((r_gpu - r_ref).abs().max() / r_ref.abs().max()).item(),
                            "rel_l2_repr_diff_vs_cpu": ((r_gpu - r_ref).norm() / r_ref.norm()).item(),
                            **argmax_report(out["logits"].float().cpu(), ref["logits"].float()), "sample_residues": n_small}
        try:  # the fp16-operand floor on the same sample: what any 16-bit-operand engine gets at best
            fold = bool(model.ln_fold_active())  # the floor in the form the engine ran in
            fl = esm2_forward(sd, small, L, H, repr_layers=[L], return_contacts=True,
                              inject=(frozenset(ALL_OPERANDS + (("FOLD",) if fold else ())), torch.float16))
            zf = lg(fl["contacts"])
            f_rep = fl["representations"][L].double()
            result["parity"]["operand_floor_same_inputs"] = {
                "contacts_logit_diff_rel_to_range": ((zf - zr).abs().max() / zr.abs().max()).item(),
                "rel_repr_diff_vs_cpu": ((f_rep - r_ref).abs().max() / r_ref.abs().max()).item(),
                "rel_l2_repr_diff_vs_cpu": ((f_rep - r_ref).norm() / r_ref.norm()).item(),
                "logits_rel_diff": ((fl["logits"] - ref["logits"]).abs().max() / ref["logits"].abs().max()).item(),
                "form": "ln_fold" if fold else "plain", **logits_l2_report(fl["logits"], ref["logits"])}
        except Exception as e:
            result["parity"]["operand_floor_same_inputs"] = {"error": str(e)[:200]}
    return result


def run_msa1b(args, dist, rank, world, dev):
    import esm
    from esm_amd.synth import MSA_DIMS, skip_param_init, synth_msa_state_dict, synth_msa_tokens

    L, E, H, F = MSA_DIMS["esm_msa1b_t12_100M_UR50S"]
    R, C = 128, 513
    ns = argparse.Namespace(layers=L, embed_dim=E, ffn_embed_dim=F, attention_heads=H, dropout=0.1, attention_dropout=0.1,
                            activation_dropout=0.1, max_positions=1024, embed_positions_msa=True,
                            embed_positions_msa_dim=E, max_tokens=2 ** 14, max_tokens_per_msa=2 ** 14)
    sd = synth_msa_state_dict(L, E, H, F, seed=0)
    with skip_param_init():
        model = esm.MSATransformer(ns, esm.Alphabet.from_architecture("msa_transformer")).eval()
    model.load_state_dict(sd)
    model = model.to(dev)
    batch = args.batch or 1
    toks = synth_msa_tokens(batch, R, C, seed=1 + rank).to(dev)
    ntok = batch * R * C
    flop = ntok * L * (32.0 * E * E + 4.0 * E * (C + R))  # SURVEY §8 a: 16.5 TFLOP per 128 x 513 MSA

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    step = lambda: model(toks, repr_layers=[L])
    with torch.no_grad():
        elapsed = timed_steps(step, args.steps, args.warmup, sync_all, dist, dev)
        prof, prof_steps = profile_steps(model, step)
    residues = batch * R * (C - 1)
    value = world * residues * args.steps / elapsed
    if rank != 0:
        return None
    result = base_result(
        args, world, "MSA residues/sec esm_msa1b_t12_100M 128-seq MSA L=512", value, elapsed,
        f"esm_msa1b_t12_100M dims (12 x 768, 12 heads): model(tokens [{batch},{R},{C}], repr_layers=[12]) — tied row "
        "attention + column attention + FFN, random-init weights", {"msas_per_gpu": batch, "rows": R, "cols": C}, model)
    result["e2e_mfma_frac_per_gpu"] = round(flop * args.steps / elapsed / (MFMA_PEAK_TFLOPS * 1e12), 4)
    result["tflops_algorithmic"] = round(flop * args.steps / elapsed / 1e12, 1)
    dom, result["roofline"] = mfma_roofline(prof)
    result["roofline_hbm"] = hbm_roofline(prof)
    result["kernel_classes"] = class_table(prof, prof_steps)
    result["library"] = library_build()
    result["roofline"]["traffic"], result["roofline"]["traffic_source"] = pmc_traffic(
        PMC_CLASS.get(dom["name"], dom["name"]), result["library"]["src_hash"], "msa1b", batch, None)
    if world == 1 and not args.no_cpu_baseline:
        from oracle.msa_oracle import msa_forward

        ncores = cpu_threads()
        n_rows = 32  # also under --quick-baseline: an 8-row sample reads ~35 % lower than the 32-row one (Weak-2)
        small = toks[:1, :n_rows].cpu()  # bounded sample: a slice of the same MSA (depth changes the tied scale)
        c0 = time.perf_counter()
        ref = msa_forward(sd, small, L, H, repr_layers=[L])
        t_cpu = time.perf_counter() - c0
        with torch.no_grad():
            got = model(small.to(dev), repr_layers=[L])
        r_gpu, r_ref = got["representations"][L].cpu().double(), ref["representations"][L].double()
        result["cpu_baseline"] = {"value": round(n_rows * (C - 1) / t_cpu, 1), "unit": "residues/s", "cores": ncores,
                                  "host_cores": os.cpu_count(), "kind": "port",
                                  "sample": f"the first {n_rows} rows of the MSA ({n_rows} x 513) through the fp32 oracle, 1 run"}
        result["parity"] = {"rel_repr_diff_vs_cpu": ((r_gpu - r_ref).abs().max() / r_ref.abs().max()).item(),
                            "rel_l2_repr_diff_vs_cpu": ((r_gpu - r_ref).norm() / r_ref.norm()).item(),
                            **argmax_report(got["logits"].float().cpu(), ref["logits"].float())}
        try:  # the floor of any 16-bit-operand engine on exactly this sample (as the 650M / 3B lines report it)
            from oracle.msa_oracle import msa_operand_floor

            c0 = time.perf_counter()
            fl = msa_operand_floor(sd, small, L, H, repr_layers=[L])
            f_r = fl["representations"][L].double()
            lg_f, lg_r = fl["logits"].double(), ref["logits"].double()
            result["parity"]["operand_floor_same_inputs"] = {
                "rel_repr_diff_vs_cpu": ((f_r - r_ref).abs().max() / r_ref.abs().max()).item(),
                "rel_l2_repr_diff_vs_cpu": ((f_r - r_ref).norm() / r_ref.norm()).item(),
                "logits_rel_diff": ((lg_f - lg_r).abs().max() / lg_r.abs().max()).item(),
                "seconds": round(time.perf_counter() - c0, 1),
                "what": "fp32 oracle with f16 rounding injected at every operand point (weights, GEMM inputs, q / k / v, P), same rows"}
        except Exception as e:  # the floor is a report, never a reason to lose the line
            result["parity"]["operand_floor_same_inputs"] = {"error": str(e)[:200]}
    return result


def run_extract_650m(args, dist, rank, world, dev):
    """The extraction driver end to end with result files written.  One step = one 64-sequence
    batch through tokenise -> forward -> device->host -> per-sequence torch.save (reference file format,
    scripts/extract.py:104-131).  --out-dir chooses the file system (default: a tmpfs directory under /dev/shm)."""
    import pathlib
    import shutil
    import tempfile

    import esm
    from esm_amd.extract import default_writer_threads, extract, make_embed_fn
    from esm_amd.synth import ESM2_DIMS, skip_param_init, synth_esm2_state_dict

    L, E, H = ESM2_DIMS["esm2_t33_650M_UR50D"]
    with skip_param_init():
        model = esm.ESM2(L, E, H).eval()
    model.load_state_dict(synth_esm2_state_dict(L, E, H, seed=0))
    model = model.to(dev)
    batch = args.batch or 64
    g = torch.Generator().manual_seed(1 + rank)
    aas = "LAGVSERTIDPKQNFYMHWC"

    def dataset(n_batches):
        n = n_batches * batch
        seqs = ["".join(aas[i] for i in torch.randint(0, 20, (args.seq_len,), generator=g).tolist()) for _ in range(n)]
        return esm.FastaBatchedDataset([f"rank{rank}/s{i}" for i in range(n)], seqs)

    alphabet = esm.Alphabet.from_architecture("ESM-1b")
    fwd = make_embed_fn(model, varlen=True)
    base = args.out_dir or ("/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir())
    # every result file of the warm-up and the timed run exists at the end (n_files / file bytes are reported): 5.3 MB per sequence
    need = (args.steps + max(args.warmup, 1)) * batch * (args.seq_len * E * 4 + (1 << 16)) * 1.05
    if not args.out_dir and shutil.disk_usage(base).free < need:
        alt = tempfile.gettempdir()
        if shutil.disk_usage(alt).free >= need:
            base = alt
        else:  # a small box: time what fits (never fewer than 4 batches) rather than fail the line
            free = max(shutil.disk_usage(base).free, shutil.disk_usage(alt).free)
            base = base if shutil.disk_usage(base).free >= shutil.disk_usage(alt).free else alt
            args.steps = max(4, min(args.steps, int(free / (need / (args.steps + max(args.warmup, 1)))) - max(args.warmup, 1) - 1))
    out_dir = pathlib.Path(tempfile.mkdtemp(prefix="esm_amd_bench_", dir=base))
    include = ["mean", "per_tok"]

    def run(ds):
        extract(ds, alphabet, fwd, L, E, [L], include, output_dir=out_dir, toks_per_batch=batch * (args.seq_len + 2),
                device=dev, gather_mean=False, log=lambda s: None, writer_threads=args.writer_threads)

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    try:
        warm, timed = dataset(max(args.warmup, 1)), dataset(args.steps)
        run(warm)
        sync_all()
        t0 = time.perf_counter()
        run(timed)  # returns after the writer threads have closed every file
        sync_all()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        n_files = sum(1 for _ in out_dir.rglob("*.pt"))
        nbytes = sum(p.stat().st_size for p in out_dir.rglob("*.pt"))
        check = None
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            # parity of what was WRITTEN: one result file of the timed run read back and compared with the CPU oracle on
            # that sequence (a bounded sample: one L = 1022 forward, ~2 s on 32 threads); its timing is the CPU baseline
            try:
                from oracle.esm2_oracle import esm2_forward

                ncores = cpu_threads()
                label, seq = timed[0]
                rec = torch.load(out_dir / f"{label}.pt")
                _, _, toks1 = alphabet.get_batch_converter()([(label, seq)])
                sd = {k: v.float().cpu() for k, v in model.state_dict().items()}
                esm2_forward(sd, toks1[:, :64], L, H, repr_layers=[L])  # warm-up of the thread pool
                c0 = time.perf_counter()
                ref = esm2_forward(sd, toks1, L, H, repr_layers=[L])["representations"][L][0, 1:len(seq) + 1].double()
                dt = time.perf_counter() - c0
                got = rec["representations"][L].double()
                gm = rec["mean_representations"][L].double()
                check = ({"rel_repr_diff_vs_cpu": ((got - ref).abs().max() / ref.abs().max()).item(),
                          "rel_l2_repr_diff_vs_cpu": ((got - ref).norm() / ref.norm()).item(),
                          "mean_repr_rel_diff_vs_cpu": ((gm - ref.mean(0)).abs().max() / ref.mean(0).abs().max()).item(),
                          "what": f"result file {label}.pt of the timed run read back (representations[{L}] [{len(seq)}, {E}] and "
                                  "its mean) against the fp32 CPU oracle on the same sequence"},
                         {"value": round(len(seq) / dt, 1), "unit": "residues/s", "cores": ncores, "host_cores": os.cpu_count(),
                          "kind": "port", "sample": f"fp32 oracle on {ncores} threads, one sequence of the timed FASTA "
                                                    f"(L = {len(seq)}), one forward after a short warm-up"})
            except Exception as e:  # reported, never raised
                check = ({"error": f"{type(e).__name__}: {e}"[:200]}, None)
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)
    value = world * batch * args.seq_len * args.steps / elapsed
    if rank != 0:
        return None
    result = base_result(
        args, world, "residues/sec ESM-2 650M L=1022 extraction driver, result files written", value, elapsed,
        "esm_amd.extract.extract on synthetic FASTA strings: tokenise -> esm2_t33_650M forward -> device->host -> "
        "one .pt per sequence (--include mean per_tok, reference file format)",
        {"batch_per_gpu": batch, "seq_len": args.seq_len, "output_fs": str(base),
         "writer_threads": args.writer_threads or default_writer_threads()}, model)
    result["files_written_this_rank"] = n_files
    result["file_gb_per_s_this_rank"] = round(nbytes / elapsed / 1e9 * args.steps / (args.steps + max(args.warmup, 1)), 3)
    result["e2e_mfma_frac_per_gpu"] = round(value / world * 1.4769e9 / (MFMA_PEAK_TFLOPS * 1e12), 4)
    result["roofline"] = None   # a host + PCIe + file-system pipeline: the forward's roofline is the esm2_650m line
    result["library"] = library_build()
    if check is not None:
        result["parity"], cb = check
        if cb is not None:
            result["cpu_baseline"] = cb
    return result


WORKLOADS = {"esm2_650m": run_esm2_650m, "esm2_3b_contacts": run_esm2_3b_contacts, "msa1b": run_msa1b,
             "extract_650m": run_extract_650m}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="esm2_650m")
    ap.add_argument("--batch", type=int, default=0, help="units per GPU per step (0 = the workload's default: 64 "
                                                         "sequences for esm2_650m, 32 for esm2_3b_contacts, 1 MSA)")
    ap.add_argument("--seq-len", type=int, default=1022)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick-baseline", action="store_true",
                    help="esm2_3b_contacts / msa1b: a smaller CPU-oracle sample (128 residues / 8 MSA rows), used by the "
                         "default run's child processes to stay inside its time budget")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="esm2_650m with --gpus N: weak = --batch sequences per GPU (default), strong = --batch sequences "
                         "in total per step (default 512), split over the GPUs")
    ap.add_argument("--no-secondary", action="store_true",
                    help="esm2_650m at N = 1: do not append the other workloads (msa1b, esm2_3b_contacts, extract_650m; "
                         "child runs, ~2 min) as `secondary_workloads`; --no-cpu-baseline implies it")
    ap.add_argument("--out-dir", default=None, help="extract_650m: directory (file system) the result files go to")
    ap.add_argument("--writer-threads", type=int, default=0, help="extract_650m: writer threads (0 = from host cores)")
    ap.add_argument("--ln-fold", type=int, choices=[0, 1], default=None,
                    help="LayerNorm fold of the engine (esmk_config.ln_fold, DESIGN.md 4.8): 1 = the per-layer LayerNorm passes "
                         "become GEMM epilogue work (the library default since round 5: faster at every batch size), 0 = off")
    ap.add_argument("--operand", choices=["f16", "bf16", "f16x2", "f16x2a", "f16x2v", "f16x3"], default=None,
                    help="MFMA operand type (default f16; bf16 is ~4 %% faster at ~7e-3 relative error; f16x2 = fp16 with "
                         "split weights W = W_hi + W_lo: 2x GEMM time, ~40 %% lower error — the precision mode with "
                         "margin under the 1e-3 contract).  Sets ESM_AMD_OPERAND for this run.")
    ap.add_argument("--qk-gain", type=float, default=2.0, help="esm2_650m: gain of the synthetic q / k projection weights (default 2)")
    ap.add_argument("--also-qk-gain", type=float, default=0.0, help="esm2_650m: after the run, time the same batch on weights with "
                    "this q / k gain as well (data-sensitivity line; timing only)")
    ap.add_argument("--ln-gamma-std", type=float, default=0.02, help="esm2_650m: spread of the synthetic LayerNorm gains (default 0.02)")
    ap.add_argument("--parity-ref", default="", help="esm2_650m: file with {tokens, repr, logits} of the fp32 oracle on 4 sample "
                                                       "sequences (written by the default run for its secondary lines): report "
                                                       "`parity` of this configuration against it")
    ap.add_argument("--save-parity-ref", default="", help=argparse.SUPPRESS)
    ap.add_argument("--spawn", action="store_true",
                    help="go through the self-launch path even for --gpus 1 (tests: the N = 1 run then initialises RCCL "
                         "exactly as an N > 1 run does)")
    ap.add_argument("--protocol-test", action="store_true",
                    help="CPU/gloo dry run of the launch + timing protocol with a stub step (tests/test_bench_protocol.py); "
                         "never a measurement")
    args = ap.parse_args()

    from esm_amd.launch import init_ranks, relaunch, under_launcher

    if args.operand:
        os.environ["ESM_AMD_OPERAND"] = args.operand
    if args.ln_fold is not None:
        os.environ["ESM_AMD_LN_FOLD"] = str(args.ln_fold)
    if (args.gpus > 1 or args.spawn) and not under_launcher():
        os.environ["ESM_AMD_BENCH_LAUNCH"] = "self-spawned"
        raise SystemExit(relaunch(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    if args.protocol_test:
        if args.batch == 0:
            args.batch = 64
        return protocol_test(args)

    dist, rank, world, local_rank = init_ranks(args.gpus, "nccl")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    with_secondary = world == 1 and args.workload == "esm2_650m" and not (args.no_cpu_baseline or args.no_secondary)
    ref_file = None
    if with_secondary and not args.save_parity_ref:
        import tempfile

        fd, ref_file = tempfile.mkstemp(prefix="esm_amd_parity_ref_", suffix=".pt")
        os.close(fd)
        args.save_parity_ref = ref_file
    try:
        result = WORKLOADS[args.workload](args, dist, rank, world, dev)
        if rank == 0:
            result["collective_backend"] = dist.get_backend() if dist is not None else None
            if with_secondary:
                try:
                    torch.cuda.empty_cache()
                    result["secondary_workloads"] = secondary_workloads(parity_ref=ref_file or "")
                except Exception as e:  # belt and braces: the flagship line is printed whatever happens here
                    result["secondary_workloads"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            print(json.dumps(result), flush=True)
    finally:
        if ref_file and os.path.exists(ref_file):
            os.remove(ref_file)
    finish(dist)


if __name__ == "__main__":
    main()
