"""List the kernels of a rocprofv3 (rocpd sqlite) trace with their launch geometry and register / LDS budgets.
    python tools/rocpd_kernels.py <results.db>"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
q = ("select name, count(*), avg(end-start)/1e3, workgroup_x, grid_x, grid_y, lds_size, vgpr_count, accum_vgpr_count, "
     "sgpr_count from kernels group by name, grid_x order by 3 desc")
print("avg_us calls wg grid_x grid_y lds vgpr agpr sgpr name")
for n, cnt, avg, wg, gx, gy, lds, vg, ag, sg in c.execute(q).fetchall():
    print(f"{avg:9.1f} {cnt:4d} {wg:4d} {gx:8d} {gy:4d} {lds:6d} {vg:4d} {ag:4d} {sg:4d} {n[:200]}")
