"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls, total / average duration.

    python tools/rocpd_summary.py gpurun_out/prof_r1/*/*_results.db > profiles/r1_kernel_stats.md
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void esmk::", "").replace("esmk::", "")
    return name[:90]


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    ncol = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {ncol}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     f"from kernels group by {ncol} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"source: {path}")
    print(f"total kernel time: {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for n, cnt, tot, avg, mn, mx in rows:
        print(f"| `{short(n)}` | {cnt} | {tot/1e6:.3f} | {avg/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*tot/total:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])
