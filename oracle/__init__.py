"""TEST INFRASTRUCTURE: CPU oracle for the msufsort hot path (see msufsort_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from .oracle import *  # noqa: F401,F403
