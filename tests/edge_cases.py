"""Constructed inputs for the branches of Saddle_Prob_Fast / getroot_K1_fast that random genotypes
never reach (SPATest.cpp:145-149 root = Inf, :166-179 bisection safeguard, :361-365 no convergence,
:368-369 cutoff doubling; saige_main.cpp:390-391 p == 0 -> p_noadj).

A deterministic search: small case-control models with a low variance ratio, variants whose carriers
are (almost) all cases -- the score then sits at the edge of the statistic's support, where the
Newton search overshoots, leaves the support or ends in a vanishing tail.  The oracle's trace says
which branch a candidate takes; the first few of every kind are kept.
"""
import dataclasses

import numpy as np

from saigegds_amd import synth
from saigegds_amd.gds import pack_dosage_2bit
from saigegds_amd.nullmod import init_nullmod

KINDS = ("root_inf", "bisect", "not_converged", "cutoff_doubled")


def _model(n, prev, seed, var_ratio, w_scale=1.0):
    """w_scale != 1: the score weights mu2 (and XVX = X' diag(mu2) X with them) are that multiple of
    mu (1 - mu).  saige_score_test_init takes mu and mu2 as independent arrays (saige_main.cpp:121-123)
    and so does sgx_init; with mu2 too small the variance left to the non-carriers, NAsigma, turns
    negative and the root search runs away -- the one way to a NON-degenerate variant that does not
    converge (a genotype vector inside the covariates' span does not converge either, but its row is
    rounding noise in the reference itself)."""
    mod = synth.synth_null_model(n, "binary", prev, n_cov=3, seed=seed, var_ratio=var_ratio)
    sm = init_nullmod(mod, np.arange(n), float("nan"), 1.0, 0.95, 0.05, var_ratio)
    if w_scale != 1.0:
        X = sm.t_X.reshape(n, sm.k)
        mu2 = np.ascontiguousarray(sm.mu2 * w_scale)
        sm = dataclasses.replace(sm, mu2=mu2, XVX=np.ascontiguousarray(X.T @ (X * mu2[:, None])))
    return sm


def _candidates(rng, sm, count):
    n = sm.n
    cases = np.flatnonzero(sm.y > 0.5)
    ctrls = np.flatnonzero(sm.y < 0.5)
    codes = np.zeros((count, n), dtype=np.uint8)
    for r in range(count):
        kind = rng.integers(0, 7)
        nc = int(rng.integers(1, max(2, min(len(cases), 40)) + 1))
        idx = rng.choice(cases, size=min(nc, len(cases)), replace=False)
        codes[r, idx] = rng.choice([1, 2], size=idx.size, p=[0.8, 0.2])
        if kind == 1:            # a few carriers among the controls
            codes[r, rng.choice(ctrls, size=int(rng.integers(1, 4)), replace=False)] = 1
        elif kind == 2:          # many missing genotypes: imputed 2 AF makes every such sample a carrier
            miss = rng.random(n) < rng.uniform(0.2, 0.9)
            miss[idx] = False
            codes[r, miss] = 3
        elif kind == 3:          # alt allele is the major one: the flipped variant
            codes[r] = 2 - np.minimum(codes[r], 2)
        elif kind == 4:          # everybody a carrier: called cases + a few controls, the rest missing
            codes[r] = 3
            codes[r, idx] = rng.choice([1, 2], size=idx.size)
            keep = rng.choice(ctrls, size=int(rng.integers(0, 6)), replace=False)
            codes[r, keep] = rng.choice([0, 1], size=keep.size)
        elif kind == 5:          # all heterozygous but the chosen cases
            codes[r] = 1
            codes[r, idx] = 2
            codes[r, rng.random(n) < 0.3] = 3
    return codes


def degenerate(sm, codes_row):
    """True when the (imputed, flipped) genotype vector lies in the span of the covariates: the residual
    adj = G - X (X'VX)^-1 X'V G of saige_main.cpp:334-345 vanishes, var2 = sum mu2 adj^2 <= 1e-10 x
    sum mu2 G^2.  Score and variance are then zero up to rounding and the reference's own beta, SE and
    p-value are noise; what it still defines -- the filter, AF, mac, num, converged -- is compared."""
    c = codes_row.astype(np.int64)
    ok = c != 3
    if not ok.any():
        return True
    af = c[ok].sum() / (2.0 * ok.sum())
    g = np.where(ok, c, 2 * af).astype(np.float64)
    if af > 0.5:
        g = 2.0 - g
    n, k = sm.n, sm.k
    adj = g - sm.t_XXVX_inv.reshape(n, k) @ (sm.XV.reshape(n, k).T @ g)
    return float(np.sum(sm.mu2 * adj * adj)) <= 1e-10 * float(np.sum(sm.mu2 * g * g))


def build(per_kind=6, n_models=60, per_model=3000, seed=7):
    """-> list of (ScanModel, packed [m, bpv], census dict, packed_degenerate [d, bpv]) covering every kind
    in KINDS; the degenerate candidates of a model (see degenerate()) ride along apart."""
    from oracle import Oracle
    rng = np.random.default_rng(seed)
    found = {k: 0 for k in KINDS}
    out = []
    for mi in range(n_models):
        n = int(rng.choice([40, 120, 200, 333, 500, 1000]))
        prev = float(rng.choice([0.05, 0.1, 0.2, 0.4]))
        vr = float(rng.choice([0.05, 0.1, 0.3, 0.5, 0.8, 0.95]))
        # once the other kinds are in: models with inconsistent weights, for non-convergence
        others = all(found[k] >= per_kind for k in KINDS if k != "not_converged")
        ws = float(rng.choice([0.1, 0.3])) if others else 1.0
        try:
            sm = _model(n, prev, 100 + mi, vr, ws)
        except np.linalg.LinAlgError:       # separable toy data: no logistic fit
            continue
        if (sm.y > 0.5).sum() < 3:
            continue
        codes = _candidates(rng, sm, per_model)
        orc = Oracle(sm)
        keep, degen, census = [], [], {k: 0 for k in KINDS}
        for r in range(per_model):
            before = orc.trace.as_dict()
            pk = pack_dosage_2bit(codes[r:r + 1])
            row, ok = orc.scan_2bit(pk)
            after = orc.trace.as_dict()
            hit = [k for k in KINDS if after[k] > before[k]]
            if not ok[0]:
                continue
            # a genotype vector inside the span of the covariates (all heterozygous, say): only the
            # filter and the counts are comparable (degenerate())
            if degenerate(sm, codes[r]):
                if len(degen) < 8:
                    degen.append(r)
                continue
            if any(found[k] < per_kind for k in hit):
                keep.append(r)
                for k in hit:
                    found[k] += 1
                    census[k] += 1
        orc.close()
        if keep:
            # a few ordinary candidates ride along
            keep = sorted(set(keep) | set(range(0, per_model, per_model // 20)))
            out.append((sm, pack_dosage_2bit(codes[keep]), census,
                        pack_dosage_2bit(codes[degen]) if degen else np.zeros((0, (sm.n + 3) // 4), dtype=np.uint8)))
        if all(found[k] >= per_kind for k in KINDS):
            break
    return out, found


if __name__ == "__main__":
    import time
    t = time.time()
    cases, found = build()
    print(found, f"{len(cases)} models, {sum(c[1].shape[0] for c in cases)} variants, {time.time() - t:.1f} s")
    from oracle import Oracle
    print("degenerate candidates carried:", sum(c[3].shape[0] for c in cases))
    for sm, pk, census, _ in cases:
        o = Oracle(sm)
        ref, valid = o.scan_2bit(pk)
        print(sm.n, pk.shape, census, o.trace.as_dict(), "p==0 fallback rows:",
              int(np.sum((ref[:, 7] == 0) & (ref[:, 5] == ref[:, 6]))))
