"""Per-kernel averages of rocprofv3 --pmc counters from rocpd sqlite outputs.
    python tools/rocpd_pmc.py gpurun_out/pmc_*/*/*_results.db"""
import re, sqlite3, sys
from collections import defaultdict

def short(n):
    m = re.search(r"(gemm9_kernel)I(DF16_|DF16b)((?:L[ib]\d+E)+)E", n)  # every template argument: EPI, VAR, HM, LNF
    if m:
        return "gemm9_kernel<" + ("f16" if m.group(2) == "DF16_" else "bf16") + "," + ",".join(re.findall(r"L[ib](\d+)E", m.group(3))) + ">"
    m = re.search(r"(gemm8_kernel|gemm9_kernel|gemm32_kernel|gemm256_kernel|gemm64_kernel|attn_fwd_kernel|attn_probs_kernel|layernorm_kernel)I([A-Za-z0-9_]*?)E", n)
    if m:
        return m.group(1) + "<" + m.group(2).replace("DF16_", "f16,").replace("DF16b", "bf16,").replace("Li", "") + ">"
    return re.sub(r"\(.*", "", n)[:60]

acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    c = sqlite3.connect(path)
    for name, grid, ctr, val, dur in c.execute(
            "select kernel_name, grid_size, counter_name, value, duration from counters_collection"):
        acc[(short(name), grid)][ctr].append(val)
        acc[(short(name), grid)]["_dur_us"].append(dur / 1e3)
for key in sorted(acc, key=lambda k: -sum(acc[k]["_dur_us"])):
    if "convert" in key[0] or "copy" in key[0].lower() or "elementwise" in key[0]:
        continue
    d = acc[key]
    print(key[0], "grid", key[1], " ".join(f"{k}={sum(v)/len(v):.4g}" for k, v in sorted(d.items())))
