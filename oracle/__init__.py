"""CPU parity oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package.  See oracle/saige_oracle.c for scope and citations.
"""
from .oracle import (GrmOracle, Oracle, OracleTrace, build_oracle, pchisq1_upper,  # noqa: F401
                     pnorm, qnorm, saddle_prob_fast)
