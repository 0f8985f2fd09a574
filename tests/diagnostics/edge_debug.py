"""Rows of the constructed edge-case inputs (tests/edge_cases.py) where GPU and oracle differ."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
from edge_cases import build, KINDS
from conftest import table_errors
from oracle import Oracle
from saigegds_amd._lib import Scanner
np.set_printoptions(linewidth=200, precision=12)
cases, found = build()
for sm, packed, census in cases:
    ref, rv = Oracle(sm).scan_2bit(packed)
    with Scanner(sm, device=0) as sc:
        for exact in (0, 1):
            sc.set_option("spa_exact", exact)
            out, v = sc.scan_2bit(packed)
            st = sc.stats()
            print("N", sm.n, "exact", exact, census, {k: st[k] for k in ("n_spa", "n_spa_dense", "n_spa_slow")})
            m = rv.astype(bool)
            errs = table_errors(out[m], ref[m])
            idx = np.flatnonzero(m)
            bad = set()
            for k, e in errs.items():
                bad |= set(idx[np.flatnonzero(~(e <= 1))])
            bad |= set(idx[out[m][:, 7] != ref[m][:, 7]])
            for j in sorted(bad):
                o = Oracle(sm); o.scan_2bit(packed[j:j + 1]); tr = o.trace.as_dict()
                print(" row", j, "gpu", out[j], "\n        ref", ref[j], "\n        trace", {k: tr[k] for k in KINDS + ("newton_iters", "spa_done", "cutoff_exit")})
