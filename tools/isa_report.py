"""Per-kernel ISA report of one HIP source of libesmk.so, and a function-by-function comparison against another git
revision — the check behind statements like "the default kernels compile to the same instructions as before"
(DESIGN.md §4.1b, §4.8).  CPU only (hipcc cross-compiles gfx950).

    python tools/isa_report.py gemm9.hip                       # registers, scratch, K-loop blocks of every kernel
    python tools/isa_report.py gemm9.hip --against HEAD~3      # which kernels changed since that revision

Labels are normalised (their numbers shift when functions are added); comment and directive lines are ignored.
hipcc is not deterministic for every kernel: the bf16 full-height V^T instantiation of gemm9 differs between two runs on
the same source — `--twice` compiles the current source twice and lists such kernels.
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from esm_amd import build  # noqa: E402


def compile_asm(csrc_dir, name, out):
    flags = [f for f in build.FLAGS if f not in ("-fPIC",)]
    cmd = ["/opt/rocm/bin/hipcc", *flags, "--cuda-device-only", "-S", "-o", out, os.path.join(csrc_dir, name)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-3000:])


def functions(path):
    fn, name = collections.OrderedDict(), None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1)
            fn[name] = []
        elif name and not line.lstrip().startswith((";", ".")) and line.strip():
            fn[name].append(re.sub(r"\.LBB\d+_", ".LBB_", line))
    return fn


def meta(path):
    out, text = {}, open(path).read()
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        g = lambda k: int(re.search(k + r" (\d+)", m.group(2)).group(1))
        out[m.group(1)] = (g(r"\.amdhsa_next_free_vgpr"), g(r"\.amdhsa_accum_offset"), g(r"\.amdhsa_private_segment_fixed_size"))
    return out


def loops(path, kernel):
    """Basic blocks of a kernel with >= 64 MFMAs (the K-loop bodies): instructions, scratch ops, AGPR moves."""
    text = open(path).read()
    i = text.index("\n" + kernel + ":")
    body = text[i:text.index("s_endpgm", i)].split("\n")
    blocks, cur = collections.OrderedDict(), None
    for line in body:
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            cur = m.group(1)
            blocks[cur] = []
        elif cur and line.strip() and not line.strip().startswith(";"):
            blocks[cur].append(line.split()[0])
    res = []
    for b, ops in blocks.items():
        n = sum(o.startswith("v_mfma") for o in ops)
        if n >= 64:
            res.append((len(ops), n, sum(o.startswith("scratch_") for o in ops), sum(o.startswith("v_accvgpr") for o in ops)))
    return res


def tied_mfma_hazards(path, kernel):
    """Inline-asm MFMAs (Op<T>::mma16_tied, common.h) are invisible to the compiler's hazard recogniser: between the first of
    them and the `s_nop 7; s_nop 7` pair that closes the first K tile, nothing but MFMAs may touch an AGPR (a v_accvgpr_read,
    an LDS / global store of an `a` register would read a result before the matrix pipe has written it).
    Returns (number of tied MFMAs, has the s_nop pair, [offending lines])."""
    text = open(path).read()
    i = text.index("\n" + kernel + ":")
    body = text[i:text.index("s_endpgm", i)].split("\n")
    in_asm, tied, nops, first, last_nop, bad = False, 0, 0, None, None, []
    ins = []
    for line in body:
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        elif t and not t.startswith((";", ".")) and not re.match(r"^\.?\w+:", t):
            ins.append((t, in_asm))
    for n, (t, a) in enumerate(ins):
        if a and t.startswith("v_mfma"):
            tied += 1
            first = n if first is None else first
        if a and t.startswith("s_nop 7") and first is not None and n + 1 < len(ins) and ins[n + 1][1] and ins[n + 1][0].startswith("s_nop 7"):
            nops += 1
            # everything between the first tied MFMA of this tile and the pair
            for t2, _ in ins[first:n]:
                if not t2.startswith("v_mfma") and re.search(r"(?<![\w.])a(\[\d+:\d+\]|\d+)\b", t2):
                    bad.append(t2)
            first = None
    return tied, nops, bad


def short(k):
    r = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    return re.sub(r"\(.*", "", r)[:110]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source", help="file name under esm_amd/csrc, e.g. gemm9.hip")
    ap.add_argument("--against", help="git revision to compare with")
    ap.add_argument("--twice", action="store_true", help="compile the current source twice: which kernels are not reproducible")
    ap.add_argument("--filter", default="", help="only kernels whose (mangled or demangled) name contains this, e.g. Li8E = EPI 8")
    a = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="isa_")
    cur = os.path.join(tmp, "cur.s")
    compile_asm(os.path.join(ROOT, "esm_amd", "csrc"), a.source, cur)
    f_cur, m_cur = functions(cur), meta(cur)
    if a.against or a.twice:
        other = os.path.join(tmp, "other.s")
        if a.against:
            old = os.path.join(tmp, "old")
            os.makedirs(old)
            tar = subprocess.run(["git", "-C", ROOT, "archive", a.against, "esm_amd/csrc"], capture_output=True, check=True).stdout
            subprocess.run(["tar", "-x", "-C", old], input=tar, check=True)
            compile_asm(os.path.join(old, "esm_amd", "csrc"), a.source, other)
        else:
            compile_asm(os.path.join(ROOT, "esm_amd", "csrc"), a.source, other)
        f_old = functions(other)
        changed = [k for k in f_old if k in f_cur and f_old[k] != f_cur[k]]
        print(f"{len(f_old)} kernels in {a.against or 'the second compile'}, {len(f_cur)} now; identical: "
              f"{sum(1 for k in f_old if k in f_cur) - len(changed)}; changed: {len(changed)}; "
              f"new: {len([k for k in f_cur if k not in f_old])}; gone: {len([k for k in f_old if k not in f_cur])}")
        for k in changed:
            print("  changed:", short(k), len(f_old[k]), "->", len(f_cur[k]), "instructions")
        for k in f_cur:
            if k not in f_old:
                print("  new:    ", short(k), m_cur.get(k))
        return
    print("kernel | VGPRs (accum offset) | scratch bytes | K-loop blocks (instructions, MFMAs, scratch ops, AGPR moves)")
    for k in f_cur:
        name = short(k)
        if (a.filter in name or a.filter in k) and k in m_cur:
            v, acc, scr = m_cur[k]
            print(f"{name} | {v} ({acc}) | {scr} | {loops(cur, k)}")


if __name__ == "__main__":
    main()
