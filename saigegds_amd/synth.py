"""Synthetic inputs of SURVEY.md section 8(d): model objects and 2-bit genotypes.

The genotype generator is counter-based (splitmix64 keyed by seed, variant,
sample), so any shard of any size can be regenerated on the device
(``sgx_synth_2bit_dev``) or here in numpy, bit for bit.
"""
from __future__ import annotations

import numpy as np

from .nullmod import NullModel

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    x = (np.asarray(x, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15)) & M64
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M64
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M64
    return x ^ (x >> np.uint64(31))


def variant_thresholds(first_variant: int, n_variants: int, seed: int,
                       log10_maf=(-3.3, -0.3), flip_frac=0.10, miss_rate=1e-3) -> np.ndarray:
    """Per-variant uint32 thresholds [M, 3] for the generator.

    MAF p = 10^U(lo, hi); with probability ``flip_frac`` the alt allele is the
    major one (alt frequency 1-p); genotype ~ Binomial(2, alt frequency);
    missing iid with ``miss_rate``.
    """
    with np.errstate(over="ignore"):
        j = np.arange(first_variant, first_variant + n_variants, dtype=np.uint64)
        h1 = splitmix64(j ^ np.uint64(seed * 2 + 1))
        h2 = splitmix64(h1)
    u1 = (h1 >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    u2 = (h2 >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    p = 10.0 ** (log10_maf[0] + (log10_maf[1] - log10_maf[0]) * u1)
    p = np.where(u2 < flip_frac, 1.0 - p, p)
    t0 = np.floor((1 - p) ** 2 * 4294967296.0)
    t1 = np.floor((1 - p * p) * 4294967296.0)
    thr = np.empty((n_variants, 3), dtype=np.uint32)
    thr[:, 0] = np.minimum(t0, 4294967295.0).astype(np.uint32)
    thr[:, 1] = np.minimum(t1, 4294967295.0).astype(np.uint32)
    thr[:, 2] = np.uint32(int(miss_rate * 4294967296.0))
    return thr


def synth_packed(n_samp: int, first_variant: int, n_variants: int, seed: int, thr: np.ndarray,
                 bytes_per_variant: int | None = None) -> np.ndarray:
    """numpy twin of ``sgx_synth_2bit_dev`` -> packed [M, bytes_per_variant]."""
    from .gds import pack_dosage_2bit
    if bytes_per_variant is None:
        bytes_per_variant = ((n_samp + 511) // 512) * 128
    out = np.zeros((n_variants, bytes_per_variant), dtype=np.uint8)
    i = np.arange(n_samp, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for r in range(n_variants):
            key = splitmix64(np.uint64(seed) ^ splitmix64(np.uint64(first_variant + r)))
            x = splitmix64((key + i) & M64)
            u = (x >> np.uint64(32)).astype(np.uint32)
            m = (x & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            code = np.where(u < thr[r, 0], 0, np.where(u < thr[r, 1], 1, 2)).astype(np.uint8)
            code[m < thr[r, 2]] = 3
            row = pack_dosage_2bit(code[None, :])[0]
            out[r, :row.size] = row
    return out


def _irls_logistic(X, y, iters=50):
    beta = np.zeros(X.shape[1])
    for _ in range(iters):
        eta = X @ beta
        mu = 1 / (1 + np.exp(-eta))
        W = mu * (1 - mu)
        step = np.linalg.solve(X.T @ (X * W[:, None]), X.T @ (y - mu))
        beta = beta + step
        if np.max(np.abs(step)) < 1e-12:
            break
    return beta


def synth_null_model(n_samp: int, trait: str = "binary", prevalence: float = 0.10,
                     n_cov: int = 3, seed: int = 20260, var_ratio: float | None = None,
                     outlier: float = 0.0) -> NullModel:
    """Model object with the structure SPAtest/seqFitNullGLMM_SPA would produce
    (tau = (1, 0): the GLMM fitted values equal the no-K glm fit)."""
    rng = np.random.default_rng(seed)
    X = np.ones((n_samp, n_cov))
    for k in range(1, n_cov):
        X[:, k] = rng.standard_normal(n_samp) if k % 2 == 1 else rng.integers(0, 2, n_samp)
    if outlier and n_cov > 1:
        X[7 % n_samp, 1] = outlier          # a heavy-tailed covariate: one sample far outside the column's scale
    bcov = np.full(n_cov - 1, 0.5)
    if outlier:
        bcov[0] = 0.0                       # (no effect on the trait: the fit stays well-posed)
    if trait == "binary":
        lo, hi = -20.0, 20.0
        lin = X[:, 1:] @ bcov
        for _ in range(80):   # solve b0 for the requested prevalence
            b0 = 0.5 * (lo + hi)
            if np.mean(1 / (1 + np.exp(-(b0 + lin)))) > prevalence:
                hi = b0
            else:
                lo = b0
        mu_true = 1 / (1 + np.exp(-(b0 + lin)))
        y = (rng.random(n_samp) < mu_true).astype(np.float64)
        beta = _irls_logistic(X, y)
        mu = 1 / (1 + np.exp(-(X @ beta)))
        V = mu * (1 - mu)
        vr = 0.94105067 if var_ratio is None else var_ratio
        tau = np.array([1.0, 0.0])
    elif trait == "quantitative":
        y = 5 + X[:, 1:] @ np.full(n_cov - 1, 0.3) + rng.standard_normal(n_samp)
        beta = np.linalg.lstsq(X, y, rcond=None)[0]
        mu = X @ beta
        V = np.ones(n_samp)
        vr = 1.03074434 if var_ratio is None else var_ratio
        tau = np.array([float(np.var(y - mu)), 0.0])
    else:
        raise ValueError(trait)
    XV = (X * V[:, None]).T
    XVX_inv = np.linalg.inv(X.T @ (X * V[:, None]))
    XXVX_inv = X @ XVX_inv
    return NullModel(trait_type=trait, tau=tau, fitted_values=mu,
                     sample_id=[f"s{i + 1}" for i in range(n_samp)], var_ratio=np.array([vr]),
                     y=y, V=V, X1=X, XV=XV, XXVX_inv=XXVX_inv, coefficients=beta)
