"""rocprofv3 PMC passes of `bench.py` -> profiles/rN_pmc_summary.json, keyed by bench.py's kernel classes.

    python tools/pmc_summary.py <out.json> <pmc_*/..._results.db> ...

FETCH_SIZE / WRITE_SIZE are KiB per launch; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream
(MI355X_MICROARCH.md, HBM section), so hbm_bytes_corrected = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  The summary
records the source hash of the profiled libesmk.so, the git SHA, the per-GPU batch and the LayerNorm-fold mode of the
profiled run; bench.py reports `roofline.traffic` only when all of them match the run it is describing."""
import json
import os
import re
import sqlite3
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

EPI_CLASS = {"2": "gemm_fc1_gelu", "5": "gemm_qkv_rope(qk)", "6": "gemm_qkv_rope(v)", "3": "lm_head_dense", "1": "lm_head_logits"}


RESID_KERNELS = set()  # residual-GEMM kernel symbols of the profiled run (gemm8 and / or gemm9, epilogue code 4)


def classify(name, nth_epi4):
    m = re.search(r"gemm([89])_kernelI(?:DF16_|DF16b)Li(\d+)E", name)
    if m:
        epi = m.group(2)
        if epi == "4":
            fams = {re.search(r"gemm([89])_kernel", k).group(1) for k in RESID_KERNELS}
            if len(fams) == 2:  # round 3a: fc2 on gemm9, the out projection on gemm8
                return "gemm_fc2" if m.group(1) == "9" else "gemm_out_proj"
            # one kernel family for both: out_proj and fc2 alternate in launch order inside every layer
            return "gemm_out_proj" if nth_epi4 % 2 == 0 else "gemm_fc2"
        return EPI_CLASS.get(epi, "gemm_epi" + epi)
    for key, cls in (("attn_fwd", "attention"), ("layernorm_kernel", "layernorm"), ("ln_finalize", "ln_finalize"),
                     ("rowstats_kernel", "rowstats"), ("attn_probs", "attention_probs"),
                     ("msa_row_softmax", "msa_row_softmax"), ("contact_", "contacts"), ("embed_kernel", "embed")):
        if key in name:
            return cls
    return None


def main(out_path, dbs, workload="esm2_650m", batch=None, ln_fold=None):
    acc = defaultdict(lambda: defaultdict(list))
    for path in dbs:
        c = sqlite3.connect(path)
        cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        order = next((x for x in ("dispatch_id", "start", "id") if x in cols), None)
        q = "select kernel_name, counter_name, value, duration" + (f", {order}" if order else "") + " from counters_collection"
        rows = c.execute(q + (f" order by {order}" if order else "")).fetchall()
        RESID_KERNELS.clear()
        RESID_KERNELS.update(r[0] for r in rows if re.search(r"gemm[89]_kernelI(?:DF16_|DF16b)Li4E", r[0]))
        one_family = len({re.search(r"gemm([89])_kernel", k).group(1) for k in RESID_KERNELS}) == 1
        seen = {}  # (dispatch key) -> class: every counter of one dispatch gets the same class
        n4 = defaultdict(int)
        for r in rows:
            name, ctr, val, dur = r[:4]
            key = (r[4] if order else None, name)
            if key not in seen or order is None:
                is4 = re.search(r"gemm[89]_kernelI(?:DF16_|DF16b)Li4E", name) is not None and one_family
                seen[key] = classify(name, n4[ctr] if order is None else n4["_"])
                if is4:
                    n4[ctr if order is None else "_"] += 1
            cls = seen[key]
            if cls is None:
                continue
            acc[cls][ctr].append(val)
            acc[cls]["_dur_us"].append(dur / 1e3)
    from esm_amd.build import library_hash

    try:
        sha = subprocess.check_output(["git", "rev-parse", "HEAD"], cwd=ROOT, text=True).strip()
    except Exception:
        sha = os.environ.get("GRAFT_GIT_SHA")
    kernels = {}
    for cls, d in acc.items():
        avg = lambda k: (sum(d[k]) / len(d[k])) if d.get(k) else None
        e = {"launches_profiled": len(d["_dur_us"]), "avg_us_profiled": round(avg("_dur_us"), 2)}
        f, w = avg("FETCH_SIZE"), avg("WRITE_SIZE")
        if f is not None and w is not None:
            e.update(fetch_kib=f, write_kib=w, hbm_bytes_corrected=(2 * f + w) * 1024)
        wc, busy = avg("SQ_WAVE_CYCLES"), avg("SQ_VALU_MFMA_BUSY_CYCLES")
        gui = avg("GRBM_GUI_ACTIVE")
        if gui:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs: per-XCD active cycles = gui / 8
            e["effective_clock_ghz"] = round(gui / 8 / (avg("_dur_us") * 1e3), 3)
            if busy is not None:  # MFMA-busy cycles summed over the 1024 SIMDs / (cycles x 1024 SIMDs)
                e["mfma_busy_frac"] = busy / (gui / 8 * 1024)
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
            if avg(k) is not None and wc:
                e[k.lower() + "_per_wave_cycle"] = avg(k) / wc
        kernels[cls] = e
    json.dump({"source": "tools/profile_bench.sh on MI355X (bench.py --steps 2 --warmup 1 --no-cpu-baseline; one rocprofv3 "
                         "--pmc pass per counter group)",
               "note": "hbm_bytes_corrected = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch (gfx950 FETCH_SIZE correction)",
               "workload": workload, "batch": batch, "ln_fold": ln_fold, "library_src_hash": library_hash(), "git_sha": sha,
               "kernels": kernels},
              open(out_path, "w"), indent=1)
    print(json.dumps(kernels, indent=1))


if __name__ == "__main__":
    argv = sys.argv[1:]

    def opt(name, default=None):
        if name in argv:
            i = argv.index(name)
            v = argv[i + 1]
            del argv[i:i + 2]
            return v
        return default

    wl = opt("--workload", "esm2_650m")
    batch = opt("--batch")        # per-GPU batch of the profiled run and its LayerNorm-fold mode: bench.py reports a traffic
    fold = opt("--ln-fold")       # figure only for exactly the (workload, batch, mode, library) a summary was taken on
    main(argv[0], argv[1:], wl, int(batch) if batch is not None else None, None if fold is None else bool(int(fold)))
