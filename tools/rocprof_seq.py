"""Per-dispatch kernel durations in launch order from a rocprofv3 results DB: python tools/rocprof_seq.py <db> [substring] [--tail N]
(the gap column is the idle time between the end of the previous dispatch and the start of this one)"""
import sqlite3
import sys

args = list(sys.argv[1:])
tail = 0
if "--tail" in args:
    i = args.index("--tail")
    tail = int(args[i + 1])
    del args[i:i + 2]
db = args[0]
sub = args[1] if len(args) > 1 else ""
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in con.execute("pragma table_info(%s)" % ks)]
name_col = "display_name" if "display_name" in cols else "kernel_name"
q = "select s.%s, d.start, d.end, d.grid_size_x, d.workgroup_size_x from %s d join %s s on d.kernel_id = s.id order by d.start" % (name_col, kd, ks)
rows = [r for r in con.execute(q) if sub in r[0]]
if tail:
    rows = rows[-tail:]
prev = None
for name, t0, t1, gx, wx in rows:
    gap = 0.0 if prev is None else (t0 - prev) * 1e-3
    prev = t1
    print("%-60s grid %8d wg %4d  %10.3f us  gap %9.3f us" % (name[:60], gx, wx, (t1 - t0) * 1e-3, gap))
