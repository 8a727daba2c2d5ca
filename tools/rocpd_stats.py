"""Per-kernel statistics (the columns of rocprofv3's kernel_stats.csv) from a rocpd SQLite database,
which is what `rocprofv3 --kernel-trace --stats` writes by default on ROCm 7.2.
usage: rocpd_stats.py results.db > kernel_stats.csv"""
import sqlite3, sys, statistics
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
rows = c.execute(f"select {name_col}, start, end from kernels").fetchall()
by = {}
for n, s, e in rows:
    by.setdefault(n, []).append(e - s)
total = sum(sum(v) for v in by.values())
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"')
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    sd = statistics.pstdev(v) if len(v) > 1 else 0.0
    print(f'"{n}",{len(v)},{sum(v)},{sum(v)/len(v):.3f},{100.0*sum(v)/total:.2f},{min(v)},{max(v)},{sd:.3f}')
