#!/usr/bin/env python3
"""How many fixed-point bits do the covariate-projection columns of the MFMA score path need?

CPU experiment behind the limb counts in kern_score_mfma.h ("Limb counts"): quantise the
t_XVX_inv_XV columns (c') and the w*X columns (e) to a given number of bits, form
    var2 = c'.XVX.c' + w - 2 e.c'      S = s - S_a.c'
in long double for real variants, and report the largest relative change of var2 and the largest
change of S / sqrt(var2) against the unquantised columns.

    python tools/limb_sim.py [N=100000]       (reads tests/golden/*.npz; no GPU, no oracle)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from saigegds_amd import synth  # noqa: E402
from saigegds_amd.gds import unpack_dosage_2bit  # noqa: E402
from saigegds_amd.nullmod import NullModel, init_nullmod  # noqa: E402

CASES = [(40, 48), (32, 48), (40, 56), (48, 40), (40, 40)]     # (bits of c' columns, bits of e columns)


def quant(col, bits):
    ex = np.frexp(np.max(np.abs(col)))[1]
    es = bits - 2 - ex
    return np.ldexp(np.rint(np.ldexp(col, es)), -es)


def run(sm, codes_of, nvar, label):
    N, K = sm.n, sm.k
    A = np.asarray(sm.t_XVX_inv_XV).reshape(N, K)
    X = np.asarray(sm.t_X).reshape(N, K)
    w = np.ones(N) if sm.quant else np.asarray(sm.mu2)
    E = X * w[:, None]
    XVX = np.asarray(sm.XVX).reshape(K, K)
    S_a, ymu = np.asarray(sm.S_a), np.asarray(sm.y_mu)
    res = {b: [] for b in CASES}
    for j in range(nvar):
        G = codes_of(j).astype(np.float64)
        if G.sum() > N:
            G = 2 - G
        if G.sum() < 10:
            continue

        def stats(Aq, Eq):
            c = (G[:, None] * Aq).sum(0, dtype=np.longdouble)
            e = (G[:, None] * Eq).sum(0, dtype=np.longdouble)
            s = np.sum(G * ymu, dtype=np.longdouble)
            ww = np.sum(G * G * w, dtype=np.longdouble)
            return float(s - S_a @ c), float(c @ XVX @ c + ww - 2 * e @ c)

        S0, v0 = stats(A, E)
        for b in CASES:
            Aq = np.column_stack([quant(A[:, k], b[0]) for k in range(K)])
            Eq = np.column_stack([quant(E[:, k], b[1]) for k in range(K)])
            S1, v1 = stats(Aq, Eq)
            res[b].append((abs(v1 - v0) / abs(v0), abs(S1 - S0) / np.sqrt(abs(v0))))
    for b in CASES:
        r = np.array(res[b])
        print(f"{label}: c' {b[0]} bits, e {b[1]} bits, {len(r)} variants: "
              f"max rel change of var2 {r[:, 0].max():.2e}, max change of S/sqrt(var2) {r[:, 1].max():.2e}")


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    mod = synth.synth_null_model(N, "binary", 0.01, n_cov=3, seed=20260)
    sm = init_nullmod(mod, np.arange(N), float("nan"), 10, 0.1, 0.05, float(mod.var_ratio[0]))
    rng = np.random.default_rng(1)
    ps = 10 ** rng.uniform(-3.3, -0.3, 200)

    def codes(j):
        u = rng.random(N)
        return np.where(u < (1 - ps[j]) ** 2, 0, np.where(u < 1 - ps[j] ** 2, 1, 2))
    run(sm, codes, 60, f"synthetic N={N} (V = mu2)")

    g = np.load(os.path.join(ROOT, "tests/golden/grm1k_10k_snp.npz"))
    m = np.load(os.path.join(ROOT, "tests/golden/saige_model.npz"))
    gm = NullModel(trait_type="binary", tau=m["tau"], fitted_values=m["fitted_values"],
                   sample_id=list(m["sample_id"]), var_ratio=m["var_ratio"], y=m["y"], V=m["V"], X1=m["X1"],
                   XV=m["XV"], XXVX_inv=m["XXVX_inv"])
    sm2 = init_nullmod(gm, np.arange(1000), float("nan"), 4, 0.1, 0.05, float(np.mean(m["var_ratio"])))
    cd = unpack_dosage_2bit(g["packed"][:400], 1000)

    def codes2(j):
        c = cd[j].astype(float)
        c[cd[j] == 3] = 0
        return c
    run(sm2, codes2, 400, "golden model (V != mu2)")


if __name__ == "__main__":
    main()
