"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a text table."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
title = sys.argv[2] if len(sys.argv) > 2 else ""
nsamp = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
print("#", title)
rows = list(cur.execute("select name, count(*), sum(duration)/1e6, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3 from kernels "
                        "where name not like '%at::native%' group by name order by sum(duration) desc"))
tot = sum(r[2] for r in rows)
print(f"# total kernel time {tot:.3f} ms over {nsamp:g} test images -> {tot/nsamp:.3f} ms/image (our kernels + rocclr copies/fills; torch setup kernels excluded)")
print("%-64s %7s %11s %10s %9s %9s %9s %6s" % ("kernel", "calls", "total_ms", "ms/image", "avg_us", "min_us", "max_us", "pct"))
for r in rows:
    print("%-64s %7d %11.3f %10.3f %9.1f %9.1f %9.1f %6.2f" % (r[0][:64], r[1], r[2], r[2] / nsamp, r[3], r[4], r[5], 100 * r[2] / tot))
print("\n# GEMM launches by grid")
for r in cur.execute("select name, grid_x/workgroup_x, count(*), avg(duration)/1e3, sum(duration)/1e6 from kernels where name like '%gemm%' "
                     "group by name, grid_x order by sum(duration) desc"):
    print("%-48s blocks=%6d calls=%5d avg_us=%9.1f total_ms=%9.3f" % (r[0][:48], r[1], r[2], r[3], r[4]))
