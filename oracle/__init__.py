"""CPU oracle (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
the product (nanorq_amd/) never does.  See oracle/rq_oracle.c for what it restates and how
parity is pinned.
"""
from .orc import *  # noqa: F401,F403
