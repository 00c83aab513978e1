"""TEST INFRASTRUCTURE: CPU oracle for the HighwayEnv hot path (see hwy_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
