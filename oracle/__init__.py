"""TEST INFRASTRUCTURE ONLY — CPU checker for the HIP ABEA path (see oracle/abea_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
