import cProfile, pstats, sys, os, io
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
from mac_bench_util import cfg1_bench
print({k: v for k, v in cfg1_bench().items() if 'phase' in k or 'per_s' in k})
pr = cProfile.Profile(); pr.enable(); out = cfg1_bench(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22); print(s.getvalue()[:6000])
