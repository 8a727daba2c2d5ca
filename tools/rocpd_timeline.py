"""Kernel time line of the LAST n launches in a rocpd database (rocprofv3 --kernel-trace): name, duration and the
gap to the previous kernel's end - where a multi-launch step spends its time between kernels.
usage: rocpd_timeline.py results.db [n [skip]]     (skip: leave out the last `skip` launches)"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = c.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
rows = rows[:len(rows) - skip][-n:]
prev = None
for name, s, e in rows:
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    print(f"{(e - s) / 1e3:9.2f} us  gap {gap:7.2f} us  {name[:110]}")
    prev = e
