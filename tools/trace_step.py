"""timeline of the last RK2 step of a short run from a rocprofv3 sqlite trace (rocprofv3 --kernel-trace -d DIR -o NAME -- python3 ...):
start offset, duration, gap to the previous kernel, kernel, grid.  usage: python tools/trace_step.py DB [marker-kernel-substring]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
mark = sys.argv[2] if len(sys.argv) > 2 else "k_rhs_lpw"
rows = list(db.execute("select d.start, d.end, s.display_name, d.grid_size_x, d.grid_size_y, d.workgroup_size_x from rocpd_kernel_dispatch d "
                       "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
idx = [i for i, r in enumerate(rows) if mark in r[2]]
i0, i1 = idx[-3], idx[-1]
t0, prev = rows[i0][0], rows[i0][0]
busy = 0
for r in rows[i0:i1 + 1]:
    print(f"{(r[0] - t0) / 1e3:9.1f} {(r[1] - r[0]) / 1e3:8.1f} gap {(r[0] - prev) / 1e3:6.1f}  {r[2][:64]:64s} {r[3]}x{r[4]} wg {r[5]}")
    prev = r[1]; busy += r[1] - r[0]
print(f"wall {(rows[i1][0] - t0) / 1e3:.1f} us, kernels {(busy - (rows[i1][1] - rows[i1][0])) / 1e3:.1f} us")
