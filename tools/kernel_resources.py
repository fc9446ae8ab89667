"""Compile one HIP source for gfx950 and print per-kernel VGPR / SGPR / scratch / LDS usage.
    python tools/kernel_resources.py esm_amd/csrc/gemm8.hip [filter]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "--cuda-device-only",
                      "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null", src], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    body = m.group(1).strip()
    if body.startswith("Function Name:"):
        cur = body.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in body:
        k, v = body.rsplit(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    if flt and flt not in name:
        continue
    print(f"{name[:70]:70s} vgpr {r.get('VGPRs','?'):>4s} agpr {r.get('AGPRs','?'):>3s} sgpr {r.get('TotalSGPRs','?'):>4s} "
          f"scratch {r.get('ScratchSize [bytes/lane]','?'):>4s} occ {r.get('Occupancy [waves/SIMD]','?')}")
