"""Per-kernel duration summary of a rocprofv3 rocpd database (the default output of
`rocprofv3 --kernel-trace -d DIR -o NAME -- python3 ...` on ROCm 7): python tools/rocpd_stats.py DB [csv_out]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                   "from kernels group by name order by sum(end-start) desc").fetchall()
lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs"]
for name, calls, tot, avg, mn, mx in rows:
    lines.append('"%s",%d,%d,%.1f,%d,%d' % (name, calls, tot, avg, mn, mx))
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write("\n".join(l for l in lines if l.startswith("Name") or '"ms_' in l or "ms_" in l.split(",")[0]) + "\n")
for l in lines[:14]:
    print(l[:160])
