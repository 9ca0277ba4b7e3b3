"""CPU oracle (test infrastructure only) -- see oracle/tetsim_oracle.c.

Importable ONLY from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from .orc import OracleNH, OraclePJ, OrcParams, build_oracle, max_threads, set_threads  # noqa: F401
