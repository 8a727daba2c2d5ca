"""rocpd_stats.py with kernel names cut to their first 70 characters (template argument lists run to hundreds):
usage: rocpd_stats_short.py results.db"""
import subprocess, sys, os
out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "rocpd_stats.py"), sys.argv[1]],
                     capture_output=True, text=True).stdout.splitlines()
print(out[0])
for line in out[1:]:
    name, rest = line.rsplit('",', 1) if '",' in line else (line, "")
    print(f'{name[:71]}",{rest}')
