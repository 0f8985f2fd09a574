"""Null-model fit: Python mirror of ``seqFitNullGLMM_SPA()``.

Host driver (reference R/saige_main.r:223-654) and the AI-REML / PCG loop of
src/saige_fitnull.cpp:739-1474, written around the implicit-GRM operator of
``libsaigehip.so`` (``sgx_grm_*``, reference :159-230, 435-536, 581-614).  All
heavy work -- every ``PCG_diag_sigma`` solve and every ``get_crossprod_b_grm``
product -- runs on the GPU; what stays here is K x K dense algebra, the glm
start values and the reference's control flow.

Random numbers.  The reference draws its Hutchinson trace vectors with R's
``rbinom`` after ``set.seed(seed)`` and picks the variance-ratio markers with
``sample.int``.  To reproduce a model bit-for-bit in its random choices the
generator is restated: Mersenne-Twister with R's seed scrambling, ``unif_rand``,
``rbinom(n, 1, 0.5)`` (inversion branch) and ``sample.int`` with
``sample.kind = "Rounding"`` (what the reference's test suite selects,
inst/unitTests/test_SAIGE.R:15,48).
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from .gds import GdsFile, pack_dosage_2bit, unpack_dosage_2bit
from .nullmod import NullModel

# ---------------------------------------------------------------------------
# R's random number stream


class RRandom:
    """``RNGkind("Mersenne-Twister", "Inversion", "Rounding")``.

    The 32-bit generator is MT19937 itself (numpy's bit generator with the key
    words R's ``set.seed`` produces), so bulk draws are vectorised."""

    _I2_32M1 = 2.328306437080797e-10

    def __init__(self, seed: int = 0):
        self._bg = np.random.MT19937()
        self.set_seed(seed)

    def set_seed(self, seed: int):
        """``set.seed(seed)``: RNG_Init of R's RNG.c (initial scrambling, then
        625 LCG words: mti and the state; FixupSeeds sets mti = 624)."""
        s = int(seed) & 0xFFFFFFFF
        for _ in range(50):
            s = (69069 * s + 1) & 0xFFFFFFFF
        words = np.empty(625, dtype=np.uint32)
        for i in range(625):
            s = (69069 * s + 1) & 0xFFFFFFFF
            words[i] = s
        self._bg.state = {"bit_generator": "MT19937", "state": {"key": words[1:].copy(), "pos": 624}}

    def unif(self, n: int) -> np.ndarray:
        """``runif(n)``: MT_genrand * 2^-32, then fixup() into (0, 1)."""
        v = self._bg.random_raw(int(n)).astype(np.float64) * 2.3283064365386963e-10
        v[v <= 0.0] = 0.5 * self._I2_32M1
        v[1.0 - v <= 0.0] = 1.0 - 0.5 * self._I2_32M1
        return v

    def unif_rand(self) -> float:
        return float(self.unif(1)[0])

    def rbinom1_half(self, n: int) -> np.ndarray:
        """``rbinom(n, 1, 0.5)``: the inversion branch of rbinom.c reduces to
        ``unif_rand() >= 0.5``."""
        return (self.unif(n) >= 0.5).astype(np.float64)

    def sample_int(self, n: int) -> np.ndarray:
        """``sample.int(n, n)`` without replacement, sample.kind "Rounding"
        (do_sample -> R_unif_index = floor(dn * unif_rand()))."""
        js = np.floor(np.arange(n, 0, -1, dtype=np.float64) * self.unif(n)).astype(np.int64).tolist()
        x = list(range(1, n + 1))
        out = [0] * n
        nn = n
        for i, j in enumerate(js):
            out[i] = x[j]
            nn -= 1
            x[j] = x[nn]
        return np.asarray(out, dtype=np.int64)


# ---------------------------------------------------------------------------
# glm families and IRLS (stats::glm.fit)


class Family:
    def __init__(self, name: str):
        self.name = name

    def linkinv(self, eta):
        if self.name == "binomial":
            return 1.0 / (1.0 + np.exp(-eta))
        return eta

    def mu_eta(self, eta):
        if self.name == "binomial":
            e = np.exp(-np.abs(eta))
            return np.maximum(e / (1 + e) ** 2, np.finfo(float).eps)
        return np.ones_like(eta)

    def variance(self, mu):
        if self.name == "binomial":
            return mu * (1 - mu)
        return np.ones_like(mu)


@dataclass
class GlmFit:
    family: Family
    coefficients: np.ndarray
    linear_predictors: np.ndarray
    fitted_values: np.ndarray
    y: np.ndarray
    residuals: np.ndarray     # working residuals for gaussian = response residuals
    offset: Optional[np.ndarray] = None


def glm_fit(X: np.ndarray, y: np.ndarray, family: str) -> GlmFit:
    """``glm(formula, family)`` by IRLS to the defaults epsilon=1e-8, maxit=25."""
    fam = Family(family)
    if family == "gaussian":
        coef = np.linalg.lstsq(X, y, rcond=None)[0]
        eta = X @ coef
        return GlmFit(fam, coef, eta, eta.copy(), y.copy(), y - eta)
    mu = (y + 0.5) / 2.0
    eta = np.log(mu / (1 - mu))
    dev_old = np.inf
    coef = np.zeros(X.shape[1])
    for _ in range(25):
        me = fam.mu_eta(eta)
        z = eta + (y - mu) / me
        w = np.sqrt(me * me / fam.variance(mu))
        coef = np.linalg.lstsq(X * w[:, None], z * w, rcond=None)[0]
        eta = X @ coef
        mu = fam.linkinv(eta)
        with np.errstate(divide="ignore", invalid="ignore"):
            d = 2 * (np.where(y > 0, y * np.log(y / mu), 0.0) + np.where(y < 1, (1 - y) * np.log((1 - y) / (1 - mu)), 0.0))
        dev = float(d.sum())
        if abs(dev - dev_old) / (abs(dev) + 0.1) < 1e-8:
            break
        dev_old = dev
    return GlmFit(fam, coef, eta, mu, y.copy(), (y - mu) / fam.mu_eta(eta))


# ---------------------------------------------------------------------------
# AI-REML pieces of src/saige_fitnull.cpp


def _mat_inv(m: np.ndarray) -> np.ndarray:
    """``mat_inv`` (:722-733): inverse of the symmetrised matrix."""
    xs = np.triu(m) + np.triu(m, 1).T
    try:
        c = np.linalg.cholesky(xs)
        ci = np.linalg.inv(c)
        return ci.T @ ci
    except np.linalg.LinAlgError:
        return np.linalg.inv(xs)


def _calc_cv(x: np.ndarray) -> float:
    """``calcCV`` (:618-623)."""
    return float(np.std(x, ddof=1) / (np.mean(x) * len(x)))


@dataclass
class _Param:
    seed: int
    tol: float
    tolPCG: float
    maxiter: int
    maxiterPCG: int
    nrun: int
    num_marker: int
    traceCVcutoff: float
    ratioCVcutoff: float
    verbose: bool


class _Fitter:
    def __init__(self, op, X: np.ndarray, y: np.ndarray, fit0: GlmFit, param: _Param, rng: RRandom):
        self.op, self.X, self.y, self.fit0, self.p, self.rng = op, X, y, fit0, param, rng
        self.fam = fit0.family
        self.offset = np.zeros_like(y) if fit0.offset is None else fit0.offset

    def pcg(self, w, tau, b):
        x, _ = self.op.pcg(w, tau, b, self.p.maxiterPCG, self.p.tolPCG)
        return x

    # get_coeff_w (:739-758)
    def get_coeff_w(self, Y, w, tau):
        X = self.X
        Sigma_iY = self.pcg(w, tau, Y)
        Sigma_iX = np.column_stack([self.pcg(w, tau, np.ascontiguousarray(X[:, i])) for i in range(X.shape[1])])
        cov = _mat_inv(X.T @ Sigma_iX)
        alpha = cov @ (Sigma_iX.T @ Y)
        eta = Y - tau[0] * (Sigma_iY - Sigma_iX @ alpha) / w
        return Sigma_iY, Sigma_iX, cov, alpha, eta

    # get_coeff (:778-813)
    def get_coeff(self, tau, alpha0, eta0):
        fam, y, offset = self.fam, self.y, self.offset
        tol_coef = 0.1
        mu = fam.linkinv(eta0)
        mu_eta = fam.mu_eta(eta0)
        Y = eta0 - offset + (y - mu) / mu_eta
        W = mu_eta * mu_eta / fam.variance(mu)
        a0 = alpha0
        for _ in range(self.p.maxiter):
            Sigma_iY, Sigma_iX, cov, alpha, eta = self.get_coeff_w(Y, W, tau)
            eta = eta + offset
            mu = fam.linkinv(eta)
            mu_eta = fam.mu_eta(eta)
            Y = eta - offset + (y - mu) / mu_eta
            W = mu_eta * mu_eta / fam.variance(mu)
            if np.max(np.abs(alpha - a0) / (np.abs(alpha) + np.abs(a0) + tol_coef)) < tol_coef:
                break
            a0 = alpha
        return dict(Y=Y, mu=mu, alpha=alpha, eta=eta, W=W, cov=cov, Sigma_iY=Sigma_iY, Sigma_iX=Sigma_iX)

    # get_trace (:627-668) / get_trace_q (:672-718)
    def get_trace(self, Sigma_iX, w, tau, cov, quant: bool):
        p, n = self.p, self.op.n
        self.rng.set_seed(p.seed)
        nrun_start, nrun_end = 0, p.nrun
        buf, buf0 = np.zeros(p.nrun), np.zeros(p.nrun)
        while True:
            for i in range(nrun_start, nrun_end):
                u = 2 * self.rng.rbinom1_half(n) - 1
                Sigma_iu = self.pcg(w, tau, u)
                Pu = Sigma_iu - Sigma_iX @ (cov @ (Sigma_iX.T @ u))
                Au = self.op.crossprod(u)
                buf[i] = float(Au @ Pu)
                buf0[i] = float(u @ Pu)
            cv = _calc_cv(buf)
            cv0 = _calc_cv(buf0) if quant else 0.0
            if cv > p.traceCVcutoff or cv0 > p.traceCVcutoff:
                nrun_start, nrun_end = nrun_end, nrun_end + 10
                buf = np.concatenate([buf, np.zeros(10)])
                buf0 = np.concatenate([buf0, np.zeros(10)])
                if p.verbose:
                    print(f"CV for trace random estimator using {p.nrun} runs is {max(cv, cv0):g} > {p.traceCVcutoff:g}")
                    print(f"try {nrun_end} runs ...")
                continue
            break
        return (float(np.mean(buf0)), float(np.mean(buf))) if quant else float(np.mean(buf))

    def _proj(self, v, Sigma_iX, cov):
        return v - Sigma_iX @ (cov @ (Sigma_iX.T @ v))

    # get_AI_score (:817-833)
    def get_AI_score(self, Y, w, tau, Sigma_iY, Sigma_iX, cov):
        PY = Sigma_iY - Sigma_iX @ (cov @ (Sigma_iX.T @ Y))
        APY = self.op.crossprod(PY)
        YPAPY = float(PY @ APY)
        trace = self.get_trace(Sigma_iX, w, tau, cov, quant=False)
        PAPY_1 = self.pcg(w, tau, APY)
        PAPY = PAPY_1 - Sigma_iX @ (cov @ (Sigma_iX.T @ PAPY_1))
        return YPAPY, trace, float(APY @ PAPY)

    # get_AI_score_q (:836-864)
    def get_AI_score_q(self, Y, w, tau, Sigma_iY, Sigma_iX, cov):
        PY = Sigma_iY - Sigma_iX @ (cov @ (Sigma_iX.T @ Y))
        A0PY = PY
        APY = self.op.crossprod(PY)
        YPAPY = (float(PY @ APY), float(PY @ A0PY))
        trace = self.get_trace(Sigma_iX, w, tau, cov, quant=True)
        AI = np.zeros((2, 2))
        PA0PY_1 = self.pcg(w, tau, A0PY)
        PA0PY = PA0PY_1 - Sigma_iX @ (cov @ (Sigma_iX.T @ PA0PY_1))
        AI[0, 0] = float(A0PY @ PA0PY)
        PAPY_1 = self.pcg(w, tau, APY)
        PAPY = PAPY_1 - Sigma_iX @ (cov @ (Sigma_iX.T @ PAPY_1))
        AI[1, 1] = float(APY @ PAPY)
        AI[1, 0] = AI[0, 1] = float(A0PY @ PAPY)
        return YPAPY, trace, AI

    # fitglmmaiRPCG (:867-896)
    def update_tau(self, c, tau0):
        tol = self.p.tol
        YPAPY, trace, AI = self.get_AI_score(c["Y"], c["W"], tau0, c["Sigma_iY"], c["Sigma_iX"], c["cov"])
        Dtau = (YPAPY - trace) / AI
        tau = tau0.copy()
        tau[1] = tau0[1] + Dtau
        tau[tau < tol] = 0
        step = 1.0
        while tau[1] < 0.0:
            step *= 0.5
            tau[1] = tau0[1] + step * Dtau
        tau[tau < tol] = 0
        return tau

    # fitglmmaiRPCG_q (:899-929)
    def update_tau_q(self, c, tau0):
        tol = self.p.tol
        zero_v = tau0 < tol
        YPAPY, trace, AI = self.get_AI_score_q(c["Y"], c["W"], tau0, c["Sigma_iY"], c["Sigma_iX"], c["cov"])
        score = np.array([YPAPY[1] - trace[0], YPAPY[0] - trace[1]])
        Dtau = np.linalg.solve(AI, score)
        tau = tau0 + Dtau
        tau[zero_v & (tau < tol)] = 0
        step = 1.0
        while tau[0] < 0.0 or tau[1] < 0.0:
            step *= 0.5
            tau = tau0 + step * Dtau
            tau[zero_v & (tau < tol)] = 0
        tau[tau < tol] = 0
        return tau

    # saige_fit_AI_PCG_binary (:949-1099) / _quant (:1103-1248)
    def fit(self, tau_in, quant: bool):
        p, fit0, y = self.p, self.fit0, self.y
        tol, n = p.tol, len(y)
        tol_inv_2 = 1 / (tol * tol)
        eta = fit0.linear_predictors.copy()
        eta0 = eta.copy()
        alpha0 = fit0.coefficients.copy()
        alpha = alpha0.copy()
        tau = np.asarray(tau_in, dtype=np.float64).copy()
        tau0 = tau.copy()
        if p.verbose:
            print("Initial variance component estimates, tau:")
            print(f"    Sigma_E: {tau[0]:g}, Sigma_G: {tau[1]:g}")
        c = self.get_coeff(tau, alpha0, eta0)
        if quant:
            YPAPY, trace, _ = self.get_AI_score_q(c["Y"], c["W"], tau, c["Sigma_iY"], c["Sigma_iX"], c["cov"])
            tau[0] = max(0.0, tau0[0] + tau0[0] * tau0[0] * (YPAPY[1] - trace[0]) / n)
            tau[1] = max(0.0, tau0[1] + tau0[1] * tau0[1] * (YPAPY[0] - trace[1]) / n)
        else:
            YPAPY, trace, _ = self.get_AI_score(c["Y"], c["W"], tau, c["Sigma_iY"], c["Sigma_iX"], c["cov"])
            tau[1] = max(0.0, tau0[1] + tau0[1] * tau0[1] * (YPAPY - trace) / n)
        cov, mu = c["cov"], c["mu"]
        it = 1
        while it <= p.maxiter:
            if p.verbose:
                print(f"Iteration {it}:")
                print("    tau: (" + ", ".join(f"{v:.7g}" for v in tau) + ")")
                print("    fixed coeff: (" + ", ".join(f"{v:.7g}" for v in alpha) + ")")
            alpha0, tau0, eta0 = c["alpha"], tau.copy(), eta
            for itry in range(1, 12):
                c = self.get_coeff(tau0, alpha0, eta0)
                tau = self.update_tau_q(c, tau0) if quant else self.update_tau(c, tau0)
                if np.max(tau) > tol_inv_2:
                    if itry <= 10:
                        tau0[1] *= 0.5
                        if p.verbose:
                            print(f"    large variance estimate observed, retry ({itry}) ...")
                        continue
                    raise OverflowError("Large variance estimate observed in the iterations, model not converged!")
                break
            cov, alpha, eta, mu = c["cov"], c["alpha"], c["eta"], c["mu"]
            if quant:
                if tau[0] <= 0:
                    raise OverflowError("Sigma_E = 0, model not converged!")
            elif tau[1] == 0:
                break
            if np.max(np.abs(tau - tau0) / (np.abs(tau) + np.abs(tau0) + tol)) < tol:
                break
            it += 1
        c = self.get_coeff(tau, alpha0, eta0)
        cov, alpha, eta, mu = c["cov"], c["alpha"], c["eta"], c["mu"]
        if p.verbose:
            print("Final tau: (" + ", ".join(f"{v:.7g}" for v in tau) + ")")
            print("    fixed coeff: (" + ", ".join(f"{v:.7g}" for v in alpha) + ")")
        return dict(coefficients=alpha, tau=tau, linear_predictors=eta, fitted_values=mu,
                    residuals=y - mu, cov=cov, converged=bool(it <= p.maxiter))

    # saige_calc_var_ratio_binary (:1255-1362) / _quant (:1366-1474)
    def var_ratio(self, tau, obj_noK, codes_of_marker, rand_index, quant: bool):
        p, fit0 = self.p, self.fit0
        fam = fit0.family
        eta, mu = fit0.linear_predictors, fit0.fitted_values
        mu_eta = fam.mu_eta(eta)
        W = mu_eta * mu_eta / fam.variance(mu)
        X1 = obj_noK["X1"]
        Sigma_iX = np.column_stack([self.pcg(W, tau, np.ascontiguousarray(X1[:, i])) for i in range(X1.shape[1])])
        XSX_inv = _mat_inv(X1.T @ Sigma_iX)
        XXVX_inv, XV = obj_noK["XXVX_inv"], obj_noK["XV"]
        rows: List[tuple] = []
        ratio_cv = p.ratioCVcutoff + 0.1
        num_marker, num_tested, snp_idx = p.num_marker, 0, 0
        while ratio_cv > p.ratioCVcutoff and snp_idx < len(rand_index):
            while num_tested < num_marker and snp_idx < len(rand_index):
                i_snp = int(rand_index[snp_idx])
                snp_idx += 1
                G0 = codes_of_marker(i_snp - 1)
                fin = np.isfinite(G0)
                Num = int(fin.sum())
                AC = float(G0[fin].sum())
                AF = AC / (2 * Num) if Num > 0 else float("nan")
                G0 = np.where(fin, G0, 2 * AF)
                if AF > 0.5:
                    G0 = 2 - G0
                    AC = 2 * Num - AC
                    AF = 1 - AF
                if AC <= 20:
                    continue
                G = G0 - XXVX_inv @ (XV @ G0)
                g = G / math.sqrt(AC)
                Sigma_iG = self.pcg(W, tau, G)
                adj = Sigma_iX @ (XSX_inv @ (X1.T @ Sigma_iG))
                var1 = (float(G @ Sigma_iG) - float(G @ adj)) / AC
                var2 = float(np.sum(g * g)) if quant else float(np.sum(mu * (1 - mu) * g * g))
                num_tested += 1
                rows.append((i_snp, AF, AC, var1, var2, var1 / var2))
                if p.verbose:
                    print(f"{num_tested:6d}, maf: {AF:0.4f}, mac: {AC:g},\tratio: {var1 / var2:0.4f} (var1: {var1:.3g}, var2: {var2:.3g})")
            ratio_cv = _calc_cv(np.array([r[5] for r in rows]))
            if ratio_cv > p.ratioCVcutoff:
                num_marker += 10
        return rows


# ---------------------------------------------------------------------------
# host driver


def _parse_formula(formula: str):
    m = re.match(r"^\s*([A-Za-z_.][\w.]*)\s*~\s*(.+)$", formula)
    if not m:
        raise ValueError("inherits(formula, \"formula\") is not TRUE")
    terms = [t.strip() for t in m.group(2).split("+")]
    if any(not re.match(r"^[A-Za-z_.][\w.]*$", t) and t != "1" for t in terms):
        raise ValueError(f"unsupported term in formula '{formula}' (plain additive numeric covariates only)")
    return m.group(1), [t for t in terms if t != "1"]


def _rank_norm(x: np.ndarray) -> np.ndarray:
    """``.rank_norm`` (R/saige_main.r:66): qnorm((rank(x) - 0.5)/n), average ties."""
    from scipy.stats import norm, rankdata
    return norm.ppf((rankdata(x, method="average") - 0.5) / len(x))


@dataclass
class FittedNullModel(NullModel):
    linear_predictors: Optional[np.ndarray] = None
    residuals: Optional[np.ndarray] = None
    cov: Optional[np.ndarray] = None
    converged: bool = True
    var_ratio_table: Dict[str, np.ndarray] = field(default_factory=dict)
    mu_noK: Optional[np.ndarray] = None
    res_noK: Optional[np.ndarray] = None


def seqFitNullGLMM_SPA(formula: str, data: Dict[str, Any], gdsfile, trait_type: str = "binary",
                       sample_col: str = "sample.id", maf: float = 0.005, missing_rate: float = 0.01,
                       max_num_snp: int = 1000000, variant_id: Optional[Sequence[int]] = None,
                       inv_norm: bool = True, X_transform: bool = True, tol: float = 0.02, maxiter: int = 20,
                       nrun: int = 30, tolPCG: float = 1e-5, maxiterPCG: int = 500, num_marker: int = 30,
                       tau_init=(0, 0), traceCVcutoff: float = 0.0025, ratioCVcutoff: float = 0.001,
                       geno_sparse: bool = True, num_thread: int = 1, model_savefn: str = "", seed: int = 200,
                       fork_loading: bool = False, verbose: bool = True, operator_factory=None) -> FittedNullModel:
    """Fit the SAIGE null model ``formula + var(GRM)`` on MI355X.

    ``data``: mapping column -> sequence (a pandas DataFrame works);
    ``gdsfile``: SeqArray GDS path / ``GdsFile`` / ``GenotypeSource``.
    ``operator_factory(packed, n_samp)`` builds the GRM operator; the default is
    the GPU one (``GrmOperator``) -- tests inject the CPU oracle to exercise
    this host logic without a GPU.  ``geno_sparse``, ``num_thread`` and
    ``fork_loading`` only select storage/threads in the reference and are
    accepted for signature compatibility.
    """
    if trait_type not in ("binary", "quantitative"):
        raise ValueError("'arg' should be one of \"binary\", \"quantitative\"")
    phenovar, covars = _parse_formula(formula)
    cols = {k: np.asarray(v) for k, v in dict(data).items()}
    if phenovar not in cols:
        raise ValueError(f"There is no '{phenovar}' in the input data frame.")
    if sample_col in [phenovar] + covars:
        raise ValueError(f"'{sample_col}' should not be in the formula.")
    if sample_col not in cols:
        raise ValueError(f"'{sample_col}' should be one of the columns in 'data'.")
    sids = [str(s) for s in cols[sample_col]]
    if len(set(sids)) != len(sids):
        raise ValueError(f"'{sample_col}' in data should be unique.")
    if verbose:
        print("SAIGE association analysis:")

    # complete cases, then the GDS sample order (R/saige_main.r:299-311)
    num = np.column_stack([np.asarray(cols[c], dtype=np.float64) for c in [phenovar] + covars])
    keep = ~np.isnan(num).any(axis=1)
    from .assoc import GenotypeSource, _open_source
    src = _open_source(gdsfile, verbose)
    gsid = [str(s) for s in src.sample_id()]
    pos = {s: i for i, s in enumerate(sids) if keep[i]}
    sel = [i for i, s in enumerate(gsid) if s in pos]
    if not sel:
        raise ValueError("No common sample.id between 'data' and the GDS file.")
    rows = np.array([pos[gsid[i]] for i in sel])
    yv = num[rows, 0]
    Xc = num[rows, 1:]
    sample_ids = [gsid[i] for i in sel]
    n_samp = len(sel)

    # genotypes of the selected samples; variant filter (seqSetFilterCond, :314-321)
    if isinstance(src, GenotypeSource):
        packed_all, n_all = src.packed, len(gsid)
        var_ids = np.asarray(src.variant_id)
    else:
        packed_all, n_all, _ = src.dosage_alt_packed()
        var_ids = np.asarray(src.read("variant.id"))
    packed_all = np.asarray(packed_all)
    want = None if variant_id is None else set(int(v) for v in variant_id)
    gpu_counts = None
    if want is None and n_samp == n_all and operator_factory is None:
        # all samples selected: the per-variant counts of the filter come from the GPU
        from ._lib import geno_stats_2bit
        gpu_counts = geno_stats_2bit(packed_all, n_all)
    if variant_id is None and verbose:
        print("Filtering variants:")
    same_samples = (n_samp == n_all)
    sel_a = np.asarray(sel)
    keep_idx: List[np.ndarray] = []
    keep_packed: List[np.ndarray] = []
    CH = 4096                                  # markers per pass: bounded host memory at any M x N
    for s0 in range(0, packed_all.shape[0], CH):
        blk = packed_all[s0:s0 + CH]
        codes = None
        if gpu_counts is not None:
            nv = gpu_counts[0][s0:s0 + CH].astype(np.int64)
            ac = gpu_counts[1][s0:s0 + CH].astype(np.int64)
            with np.errstate(invalid="ignore", divide="ignore"):
                af = ac / (2.0 * nv)
            mafv = np.minimum(af, 1 - af)
            v = (mafv >= maf) & ((n_samp - nv) / n_samp <= missing_rate)
            loc = np.flatnonzero(v)
            keep_idx.append(loc + s0)
            keep_packed.append(np.ascontiguousarray(blk[loc]))
            continue
        if want is None or not same_samples:       # (a given variant list on all samples needs no decode)
            codes = unpack_dosage_2bit(blk, n_all)
            if not same_samples:
                codes = codes[:, sel_a]
        if want is None:
            valid = codes != 3
            nv = valid.sum(axis=1)
            ac = np.where(valid, codes, 0).sum(axis=1, dtype=np.int64)
            with np.errstate(invalid="ignore", divide="ignore"):
                af = ac / (2.0 * nv)
            mafv = np.minimum(af, 1 - af)
            v = (mafv >= maf) & ((n_samp - nv) / n_samp <= missing_rate)
        else:
            v = np.fromiter((int(x) in want for x in var_ids[s0:s0 + CH]), dtype=bool, count=blk.shape[0])
        loc = np.flatnonzero(v)
        keep_idx.append(loc + s0)
        keep_packed.append(np.ascontiguousarray(blk[loc]) if same_samples else pack_dosage_2bit(codes[loc]))
    rng = RRandom(seed)
    idx = np.concatenate(keep_idx) if keep_idx else np.zeros(0, dtype=np.int64)
    packed = np.concatenate(keep_packed) if keep_packed else np.zeros((0, (n_samp + 3) // 4), dtype=np.uint8)
    del keep_packed
    n_before = idx.size
    if max_num_snp > 0 and idx.size > max_num_snp:
        rng.set_seed(seed)
        pick = np.sort(rng.sample_int(idx.size)[:max_num_snp] - 1)     # sample(which(v), max.num.snp)
        idx, packed = idx[pick], np.ascontiguousarray(packed[pick])
    n_var = idx.size
    if verbose:
        print(f"Fit the null model: {formula} + var(GRM)")
        print(f"    # of samples: {n_samp:,}")
        print(f"    # of variants: {n_var:,}" + (f" (randomly selected from {n_before:,})" if n_before > n_var else ""))

    # design matrix, QR transform (:352-387)
    X = np.column_stack([np.ones(n_samp), Xc])
    X_name = ["(Intercept)"] + covars
    X_qrr = None
    do_tx = bool(X_transform) and X.shape[1] > 1
    if do_tx:
        Q, R = np.linalg.qr(X)
        X_fit = Q * math.sqrt(n_samp)
        X_qrr = R
        if verbose:
            print("Transform on the design matrix with QR decomposition:")
            print("    new formula: y ~ " + " + ".join(f"x_{i}" for i in range(X.shape[1])) + " - 1")
    else:
        X_fit = X

    # GRM operator (saige_store_2b_geno / saige_store_sp_geno)
    if operator_factory is None:
        from ._lib import GrmOperator
        op = GrmOperator(packed, n_samp)
    else:
        op = operator_factory(packed, n_samp)
    param = _Param(seed=seed, tol=tol, tolPCG=tolPCG, maxiter=int(maxiter), maxiterPCG=int(maxiterPCG),
                   nrun=int(nrun), num_marker=int(num_marker), traceCVcutoff=traceCVcutoff,
                   ratioCVcutoff=ratioCVcutoff, verbose=verbose)
    tau_init = np.nan_to_num(np.asarray(tau_init, dtype=np.float64), nan=0.0)
    tau_init[tau_init < 0] = 0

    def marker(i):
        c8 = unpack_dosage_2bit(packed[i:i + 1], n_samp)[0]
        c = c8.astype(np.float64)
        c[c8 == 3] = np.nan
        return c

    try:
        if trait_type == "binary":
            if len(np.unique(yv)) != 2:
                raise ValueError("The outcome variable has more than 2 categories!")
            fit0 = glm_fit(X_fit, yv, "binomial")
            if verbose:
                print("Initial fixed-effect coefficients:", fit0.coefficients)
            # SPAtest:::ScoreTest_wSaddleApprox_NULL_Model (full-rank X1)
            mu0 = fit0.fitted_values
            V = mu0 * (1 - mu0)
            XV = (X_fit * V[:, None]).T
            XXVX_inv = X_fit @ np.linalg.inv(X_fit.T @ (X_fit * V[:, None]))
            obj_noK = dict(y=yv, mu=mu0, res=yv - mu0, V=V, X1=X_fit, XV=XV, XXVX_inv=XXVX_inv)
            tau = np.array([1.0, 0.5 if tau_init[1] == 0 else tau_init[1]])     # :490-497
            fitter = _Fitter(op, X_fit, yv, fit0, param, rng)
            glmm = fitter.fit(tau, quant=False)
        else:
            y_fit = yv
            if inv_norm:
                f = glm_fit(X_fit, yv, "gaussian")
                resid_sd = float(np.std(f.residuals, ddof=1))
                y_fit = _rank_norm(f.residuals) * resid_sd
                if verbose:
                    print(f"Inverse normal transformation on residuals with standard deviation: {resid_sd}")
            fit0 = glm_fit(X_fit, y_fit, "gaussian")
            mu0 = fit0.fitted_values
            V = np.ones(n_samp)
            obj_noK = dict(y=fit0.y, mu=mu0, res=fit0.y - mu0, V=V, X1=X_fit, XV=X_fit.T.copy(),
                           XXVX_inv=X_fit @ np.linalg.inv(X_fit.T @ X_fit))
            tau = tau_init.copy()
            if tau.sum() == 0:
                tau = np.array([0.5, 0.5])
            Yw = fit0.linear_predictors + (fit0.y - mu0)
            tau = float(np.var(Yw, ddof=1)) * tau / tau.sum()              # :582-589
            fitter = _Fitter(op, X_fit, fit0.y, fit0, param, rng)
            glmm = fitter.fit(tau, quant=True)

        if verbose:
            print("Calculate the average ratio of variances:")
        rng.set_seed(seed)
        rand_index = rng.sample_int(n_var)
        rows_vr = fitter.var_ratio(glmm["tau"], obj_noK, marker, rand_index, quant=(trait_type != "binary"))
    finally:
        if hasattr(op, "close"):
            op.close()
    rows_vr.sort(key=lambda r: r[0])
    vr = {"id": var_ids[idx][[r[0] - 1 for r in rows_vr]],
          "maf": np.array([r[1] for r in rows_vr]), "mac": np.array([r[2] for r in rows_vr]),
          "var1": np.array([r[3] for r in rows_vr]), "var2": np.array([r[4] for r in rows_vr]),
          "ratio": np.array([r[5] for r in rows_vr])}
    if verbose:
        print(f"    ratio avg. is {np.mean(vr['ratio'])}, sd: {np.std(vr['ratio'], ddof=1)}")

    coef = glmm["coefficients"]
    if do_tx:
        coef = np.linalg.solve(X_qrr, coef * math.sqrt(n_samp))            # :614-622
    model = FittedNullModel(
        trait_type=trait_type, tau=glmm["tau"], fitted_values=glmm["fitted_values"], sample_id=sample_ids,
        var_ratio=vr["ratio"], y=obj_noK["y"], V=obj_noK["V"], X1=obj_noK["X1"], XV=obj_noK["XV"],
        XXVX_inv=obj_noK["XXVX_inv"], coefficients=coef, variant_id=var_ids[idx],
        linear_predictors=glmm["linear_predictors"], residuals=glmm["residuals"], cov=glmm["cov"],
        converged=glmm["converged"], var_ratio_table=vr, mu_noK=obj_noK["mu"], res_noK=obj_noK["res"])
    model.coef_names = X_name
    if model_savefn:
        from .results import save_model
        if verbose:
            print(f"Save the model to '{model_savefn}'")
        save_model(model, model_savefn)
    if verbose:
        print("Done.")
    return model


def glmmHeritability(modobj, adjust: bool = True) -> float:
    """``glmmHeritability(modobj, adjust)`` (R/saige_main.r:666-691): liability-scale estimate from
    tau; for binary outcomes optionally adjusted by the prevalence (Zhou et al. 2018, Suppl. Table 7)."""
    from .nullmod import load_modobj
    m = load_modobj(modobj)
    if m.trait_type == "binary":
        tau, r = float(m.tau[1]), 1.0
        if adjust:
            y = np.asarray(m.y)
            r = 2.970 + 0.372 * math.log10(float(np.sum(y == 1)) / y.size)
        return tau / (math.pi * math.pi / 3 + tau) * r
    if m.trait_type == "quantitative":
        return float(m.tau[1]) / float(np.sum(m.tau))
    raise ValueError("Invalid 'modobj$trait.type'.")
