"""CPU experiment behind the cumulant-series SPA stage (kern_spa4.h).

For every flagged variant of a synthetic block the saddlepoint p-value is computed twice:
by the oracle's Saddle_Prob_Fast (exp/log sums over the carriers in every Newton step) and
from the carriers' cumulant sums kappa_n = sum_i g_i^n c_n(mu_i), n <= NC, with every
K1/K2/Korg evaluation a polynomial in t.  Prints the relative differences by x = max|g t|.

    python tests/diagnostics/cumulant_proto.py [N] [M] [NC]
"""
import os
import sys
import math

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from saigegds_amd import synth                      # noqa: E402
from saigegds_amd.nullmod import init_nullmod       # noqa: E402
from oracle import Oracle                           # noqa: E402
from oracle import oracle as orc                    # noqa: E402


def bernoulli_cumulant_polys(nmax):
    """c_n(mu) for n = 2..nmax as integer coefficient lists in u = mu(1-mu):
    even n: c_n = sum_k a[k] u^k; odd n: c_n = (1-2mu) sum_k b[k] u^k (k from 1)."""
    polys = {2: [0, 1]}
    for n in range(2, nmax):
        p = polys[n]
        if n % 2 == 0:      # D(sum a_k u^k) = d sum k a_k u^k
            polys[n + 1] = [k * a for k, a in enumerate(p)]
        else:               # D(d sum b_k u^k) = sum (k b_k - (4k-2) b_{k-1}) u^k
            q = [0] * (len(p) + 1)
            for k, b in enumerate(p):
                q[k] += k * b
                q[k + 1] -= (4 * (k + 1) - 2) * b
            polys[n + 1] = q
    return polys


def series_spa(g, mu, q, m1, var1, nc, polys):
    """Saddle_Prob_Fast from cumulant sums (cutoff 2, as single_test_bin calls it)."""
    u = mu * (1 - mu)
    d = 1 - 2 * mu
    kap = np.zeros(nc + 1)
    kap[1] = np.sum(g * mu)
    gp = g.copy()
    for n in range(2, nc + 1):
        gp = gp * g
        c = np.polyval(polys[n][::-1], u)
        if n % 2:
            c = c * d
        kap[n] = np.sum(gp * c)
    NAmu = m1 - kap[1]
    NAsigma = var1 - kap[2]
    fact = [math.factorial(n) for n in range(nc + 2)]

    def K1(t):
        return sum(kap[n] * t ** (n - 1) / fact[n - 1] for n in range(1, nc + 1))

    def K2(t):
        return sum(kap[n] * t ** (n - 2) / fact[n - 2] for n in range(2, nc + 1))

    def K0(t):
        return sum(kap[n] * t ** n / fact[n] for n in range(1, nc + 1))

    def root(qq):
        t = 0.0
        k1 = K1(t) - qq + NAmu + NAsigma * t
        prev = math.inf
        conv = False
        rt = 0.0
        for _ in range(1000):
            k2 = K2(t) + NAsigma
            tn = t - k1 / k2
            if not math.isfinite(tn):
                break
            if abs(tn - t) < 0.0001220703125:
                conv = True
                break
            nk1 = K1(tn) - qq + NAmu + NAsigma * tn
            if np.sign(k1) != np.sign(nk1):
                if abs(tn - t) > prev - 0.0001220703125:
                    tn = t + np.sign(nk1 - k1) * prev * 0.5
                    nk1 = K1(tn) - qq + NAmu + NAsigma * tn
                    prev *= 0.5
                else:
                    prev = abs(tn - t)
            rt = t = tn
            k1 = nk1
        return rt, conv

    def prob(t, qq):
        K = K0(t) + NAmu * t + 0.5 * NAsigma * t * t
        k2 = K2(t) + NAsigma
        w = np.sign(t) * math.sqrt(2 * (t * qq - K))
        v = t * math.sqrt(k2)
        z = w + math.log(v / w) / w
        return orc.pnorm(z, lower=False) if z > 0 else -orc.pnorm(z, lower=True)

    s = q - m1
    qinv = -s + m1
    r1, c1 = root(q)
    r2, c2 = root(qinv)
    p = abs(prob(r1, q)) + abs(prob(r2, qinv))
    xmax = np.max(np.abs(g)) * max(abs(r1), abs(r2))
    # the guard of kern_spa4.h: last two terms of the K2 series relative to its sum, at both roots
    def tail(t):
        return (abs(kap[nc] * t ** (nc - 2) / fact[nc - 2]) + abs(kap[nc - 1] * t ** (nc - 3) / fact[nc - 3])) / abs(K2(t))
    last = max(tail(r1), tail(r2))
    return p, c1 and c2, xmax, last


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 430_000
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    nc = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    prev = float(sys.argv[4]) if len(sys.argv) > 4 else 0.01
    seed = 20260
    mod = synth.synth_null_model(n, "binary", prev, n_cov=3, seed=seed)
    sm = init_nullmod(mod, np.arange(n), float("nan"), 10.0, 0.1, 0.05, float(mod.var_ratio[0]))
    thr = synth.variant_thresholds(0, m, seed)
    packed = synth.synth_packed(n, 0, m, seed, thr)
    ref, valid = Oracle(sm).scan_2bit(packed)
    flagged = np.where(valid.astype(bool) & (ref[:, 6] <= 0.05))[0]
    print(f"N={n} M={m}: {flagged.size} flagged")
    polys = bernoulli_cumulant_polys(nc)
    X = sm.t_X.reshape(n, sm.k)
    A = sm.t_XVX_inv_XV.reshape(n, sm.k)
    rows = []
    for j in flagged:
        code = (packed[j][np.arange(n) // 4] >> (2 * (np.arange(n) % 4))) & 3
        G = code.astype(np.float64)
        miss = code == 3
        num = n - miss.sum()
        AC = G[~miss].sum()
        AF = AC / (2 * num)
        G[miss] = 2 * AF
        minus = AF > 0.5
        if minus:
            G = 2 - G
        AC2 = 2 * num - AC if minus else AC
        c = A.T @ G
        adj = (G - X @ c) / math.sqrt(AC2)
        qv = float(np.dot(sm.y, adj))
        m1 = float(np.dot(sm.mu, adj))
        var2 = float(np.dot(sm.mu2, adj * adj))
        var1 = var2 * sm.var_ratio
        T = qv - m1
        qt = T / math.sqrt(var1) * math.sqrt(var2) + m1
        idx = np.nonzero(G)[0].astype(np.int32)
        if abs(qt - m1) / math.sqrt(var2) < 2.0:
            continue
        p_ref, conv, _ = orc.saddle_prob_fast(qt, m1, var2, sm.mu, adj, idx)
        try:
            p_ser, conv2, xmax, last = series_spa(adj[idx], sm.mu[idx], qt, m1, var2, nc, polys)
        except (ValueError, ZeroDivisionError, OverflowError):
            continue
        maf = min(AF, 1 - AF)
        rows.append((maf, idx.size, xmax, abs(p_ser / p_ref - 1), last, p_ref, conv, conv2))
    rows.sort(key=lambda r: r[2])
    print("   maf     nnz    x=max|g t|   |p_ser/p_ref-1|   K2 tail     p_ref")
    for r in rows[:: max(1, len(rows) // 60)]:
        print(f"{r[0]:8.5f} {r[1]:7d} {r[2]:10.4f} {r[3]:14.3e} {r[4]:12.3e} {r[5]:10.3e} {r[6]} {r[7]}")
    a = np.array([(r[2], r[3], r[1]) for r in rows])
    for xm in (0.05, 0.1, 0.15, 0.2, 0.3, 0.4, 0.5, 0.75, 1.0, 1.5, 2.0, 3.0):
        sel = a[:, 0] <= xm
        if sel.any():
            print(f"x <= {xm}: {sel.sum():5d} variants, carriers {a[sel, 2].sum() / a[:, 2].sum():.3f} of all, "
                  f"max rel diff {a[sel, 1].max():.3e}")


if __name__ == "__main__":
    main()
