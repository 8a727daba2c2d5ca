"""CPU oracle for the cslam loop-closure hot path — TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under cslam_amd/ imports this package.
"""
