"""Per-dispatch kernel durations in launch order from a rocprofv3 results DB: python tools/rocprof_seq.py <db> [substring]"""
import sqlite3
import sys

db = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in con.execute("pragma table_info(%s)" % ks)]
name_col = "display_name" if "display_name" in cols else "kernel_name"
q = "select s.%s, d.start, d.end, d.grid_size_x, d.workgroup_size_x from %s d join %s s on d.kernel_id = s.id order by d.start" % (name_col, kd, ks)
for name, t0, t1, gx, wx in con.execute(q):
    if sub in name:
        print("%-60s grid %8d wg %4d  %10.3f us" % (name[:60], gx, wx, (t1 - t0) * 1e-3))
