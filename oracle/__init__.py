"""CPU oracle for the LSQ encoding hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
package.  The product (local-search-quantization_amd/) never does.
See oracle/lsq_oracle.c for the reference citations and the "parity unpinned" statement.
"""
from .oracle import *  # noqa: F401,F403
