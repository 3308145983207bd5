"""Dependent-launch gaps of a rocprofv3 --kernel-trace result: for every kernel, the idle time on the device since the previous kernel
ended (single stream), summed by (previous kernel -> kernel).  python tools/gap_analysis.py <results.db> [top_n]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = list(cur.execute("select name, start, end from kernels order by start"))
if "--ours" in sys.argv:             # only the stretch after the last torch kernel (input generation): the engine's own launches
    last = max(i for i, r in enumerate(rows) if "at::native" in r[0])
    rows = rows[last + 1:]
    print(f"window: {len(rows)} kernels, wall {(rows[-1][2] - rows[0][1]) / 1e6:.2f} ms")
gaps = collections.defaultdict(lambda: [0, 0.0])
busy = idle = 0.0
for (pn, ps, pe), (n, s, e) in zip(rows, rows[1:]):
    g = max(0, s - pe) / 1e3
    if g > 2000: continue            # host-side pauses between bench legs
    key = (pn.split("(")[0][:40], n.split("(")[0][:40])
    gaps[key][0] += 1; gaps[key][1] += g
    idle += g; busy += (e - s) / 1e3
print(f"kernels {len(rows)}  busy {busy/1e3:.1f} ms  idle-between-kernels {idle/1e3:.1f} ms  mean gap {idle/len(rows):.1f} us")
for (a, b), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{t/1e3:8.2f} ms  n={c:5d}  mean {t/c:6.1f} us   {a}  ->  {b}")

if "--ours" in sys.argv:
    tot = collections.defaultdict(lambda: [0, 0.0])
    for n, st_, en in rows:
        k = n.split("(")[0][:56]; tot[k][0] += 1; tot[k][1] += (en - st_) / 1e6
    print("\nper-kernel busy time in the window:")
    for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:18]:
        print(f"{t:8.2f} ms  n={c:5d}  avg {1e3 * t / c:7.1f} us  {k}")
