"""Per-kernel totals of a rocprofv3 run kept as a rocpd database (what --stats prints when no CSV is asked for):
rocpd_kernels.py RESULTS.db ..."""
import sqlite3
import sys

for db in sys.argv[1:]:
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch_")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol_")][0]
    print(db)
    q = ("select s.kernel_name, count(*), sum(d.end - d.start) / 1e6, avg(d.end - d.start) / 1e6 from %s d join %s s on d.kernel_id = s.id "
         "group by s.kernel_name order by 3 desc" % (kd, ks))
    for name, n, total, avg in c.execute(q):
        print("  %-70s %6d calls %10.2f ms  avg %8.3f ms" % (name[:70], n, total, avg))
