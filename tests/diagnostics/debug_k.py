#!/usr/bin/env python3
"""GPU vs oracle for one covariate count: python tests/diagnostics/debug_k.py K [N M]  (rows that differ)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch  # noqa
from saigegds_amd import synth
from saigegds_amd._lib import Scanner
from saigegds_amd.nullmod import init_nullmod
from oracle.oracle import Oracle

k = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000; m = int(sys.argv[3]) if len(sys.argv) > 3 else 400
mod = synth.synth_null_model(n, "binary", 0.1, n_cov=k, seed=11 + k)
sm = init_nullmod(mod, np.arange(n), float("nan"), 10, 0.1, 0.05, float(mod.var_ratio[0]))
thr = synth.variant_thresholds(0, m, 11 + k, log10_maf=(-2.5, -0.3), flip_frac=0.3, miss_rate=1e-2)
packed = synth.synth_packed(n, 0, m, 11 + k, thr)
ref, rv = Oracle(sm).scan_2bit(packed)
for opt in (None, ("force_dense", 1)):
    with Scanner(sm) as sc:
        if opt: sc.set_option(*opt)
        out, valid = sc.scan_2bit(packed)
        st = sc.stats()
    bad = []
    for j in range(m):
        if rv[j] and valid[j]:
            rel = np.abs(out[j, 3:6] - ref[j, 3:6]) / np.maximum(np.abs(ref[j, 3:6]), 1e-300)
            if np.nanmax(rel) > 1e-9: bad.append((j, out[j, 3:8].tolist(), ref[j, 3:8].tolist()))
    print("option", opt, "n_spa", st["n_spa"], "dense", st["n_spa_dense"], "slow", st["n_spa_slow"], "bad rows", len(bad))
    for b in bad[:6]: print("  ", b)
