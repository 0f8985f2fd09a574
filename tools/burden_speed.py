#!/usr/bin/env python3
"""Throughput of sgx_burden_2bit (device collapse + FP64 dosage scan) at N = 430 000:
python tools/burden_speed.py [N] [units] [variants_per_unit]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from saigegds_amd import synth
from saigegds_amd._lib import Scanner
from saigegds_amd.nullmod import init_nullmod

N = int(sys.argv[1]) if len(sys.argv) > 1 else 430000
U = int(sys.argv[2]) if len(sys.argv) > 2 else 400
V = int(sys.argv[3]) if len(sys.argv) > 3 else 20
mod = synth.synth_null_model(N, "binary", 0.01, n_cov=3, seed=20260)
sm = init_nullmod(mod, np.arange(N), 0.0, 0.0, 1.0, 0.05, float(mod.var_ratio[0]))
M = U * V
thr = synth.variant_thresholds(0, M, 5, log10_maf=(-3.3, -1.5), flip_frac=0.0, miss_rate=1e-3)
rng = np.random.default_rng(0)
packed = np.zeros((M, (N + 3) // 4), dtype=np.uint8)
for j in range(M):                      # cheap random rare variants (not the counter-based generator)
    p = 10 ** rng.uniform(-3.3, -1.5)
    idx = rng.integers(0, N, size=max(1, int(2 * p * N)))
    np.add.at(packed[j], idx // 4, (1 << (2 * (idx % 4))).astype(np.uint8))
lut = np.tile(np.array([0, 1, 2, 0.01]) / V, (M, 1))
row_ptr = np.arange(0, M + 1, V)
with Scanner(sm) as sc:
    sc.burden_2bit(packed[:V * 4], row_ptr[:5], np.arange(V * 4, dtype=np.int32), lut[:V * 4])
    t = time.time()
    out, valid = sc.burden_2bit(packed, row_ptr, np.arange(M, dtype=np.int32), lut)
    dt = time.time() - t
    st = sc.stats()
print(f"N={N} units={U} x {V} variants: {dt*1e3:.1f} ms wall ({U/dt:.0f} burden rows/s incl. H2D of {packed.nbytes/1e6:.0f} MB), "
      f"device score {st['ms_score']:.1f} ms spa {st['ms_spa']:.1f} ms, valid {int(valid.sum())}, n_spa {st['n_spa']}")
