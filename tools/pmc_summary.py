"""Per-kernel sums of the PMC counters in a rocprofv3 (sqlite) result: python tools/pmc_summary.py <results.db> [kernel-substring]."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
def tab(prefix):
    return next(t for t in tabs if t.startswith(prefix))
pmc, info, disp, sym = tab("rocpd_pmc_event"), tab("rocpd_info_pmc"), tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol")
q = f"""select s.kernel_name, d.grid_size_x / d.workgroup_size_x, p.name, count(*), sum(e.value)
        from {pmc} e join {info} p on e.pmc_id = p.id join {disp} d on e.event_id = d.event_id join {sym} s on d.kernel_id = s.id
        group by s.kernel_name, d.grid_size_x / d.workgroup_size_x, p.name"""
for name, blocks, ctr, n, tot in cur.execute(q):
    if sub in name:
        print(f"{name[:48]:48s} blocks={blocks:6d} {ctr:12s} launches={n:4d} sum={tot:.6g} per_launch={tot / n:.6g}")
