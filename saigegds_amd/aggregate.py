"""Aggregate tests on MI355X: burden, ACAT-V and ACAT-O.

Python mirror of ``seqAssocGLMM_spaBurden()``, ``seqAssocGLMM_spaACAT_V()``,
``seqAssocGLMM_spaACAT_O()`` and ``pACAT()`` (reference R/assoc_aggregate.r:51-797,
native side src/saige_main.cpp:466-1052).  The reference calls one native routine per
unit; here the work of ALL units goes through the scan library in three batches:

1. one ``sgx_scan_2bit`` over every variant that occurs in a unit (thresholds 0 / 0 / 1 as
   ``.init_nullmod(modobj, ii, 0, 0, 1, ...)`` sets them): its AF / num columns give the
   per-variant maf and mac of ``ds_mat_mafmac`` and its p-values are the single-variant
   tests of ACAT-V;
2. one ``sgx_burden_2bit`` over all (unit, weight) burden rows: the weighted, mean-imputed,
   minor-allele-oriented collapse of ``ds_mat_burden`` is a device kernel, followed on the
   device by the same single-variant test;
3. the Cauchy combination (``acat_pval``) on the host.

Only 2-bit genotype input (the RAW branch of the reference routines) is supported.
"""
from __future__ import annotations

import math
from typing import Any, Dict, List, Optional, Sequence, Union

import numpy as np

from ._lib import Scanner
from .assoc import GenotypeSource, _is_num, _open_source
from .gds import GdsFile, pack_dosage_2bit, unpack_dosage_2bit
from .nullmod import ModelError, NullModel, init_nullmod, load_modobj

# AggrParamBeta, R/assoc_aggregate.r:18-19: columns (shape1, shape2) = (1,1) and (1,25)
AggrParamBeta = np.array([[1.0, 1.0], [1.0, 25.0]]).T     # [2, n_weights]

ROUND_ZERO = 1e-300
ROUND_ONE = 1 - 1e-16


def _dbeta(x: np.ndarray, a: float, b: float) -> np.ndarray:
    """``Rf_dbeta(x, a, b, FALSE)``."""
    from scipy.stats import beta
    return beta.pdf(np.asarray(x, dtype=np.float64), a, b)


def acat_pval(pval: Sequence[float], w: Optional[Sequence[float]] = None, throw_error: bool = False) -> float:
    """``acat_pval`` (src/saige_main.cpp:1001-1052)."""
    p = np.asarray(pval, dtype=np.float64)
    w = np.ones_like(p) if w is None else np.asarray(w, dtype=np.float64)
    ok = np.isfinite(p) & np.isfinite(w)
    sumw = float(np.sum(w[ok]))
    if sumw <= 0:
        if throw_error:
            raise ValueError("the sum of weights should be > 0.")
        return float("nan")
    tstat = 0.0
    for pi, wi in zip(p[ok], w[ok]):
        if pi < 0 or pi > 1:
            if throw_error:
                raise ValueError(f"Invalid input p-value: {pi:g}.")
            return float("nan")
        pi = ROUND_ZERO if pi < ROUND_ZERO else (ROUND_ONE if pi > ROUND_ONE else pi)
        if pi >= 1e-15:
            tstat += wi * math.tan(math.pi * (0.5 - pi))
        else:
            tstat += wi / pi / math.pi
    tstat /= sumw
    if tstat <= 5e14:
        return 0.5 - math.atan(tstat) / math.pi
    return 1.0 / tstat / math.pi


def pACAT(p: Sequence[float], w: Optional[Sequence[float]] = None) -> float:
    """``pACAT(p, w)`` (R/assoc_aggregate.r; native ``saige_acat_p``, :1054-1082)."""
    p = np.asarray(p, dtype=np.float64)
    if p.size <= 0:
        raise ValueError("the number of p-values should be > 0.")
    if p.size == 1:
        return float(p[0])
    if w is not None and len(w) != p.size:
        raise ValueError("weights should have the same length as p-values.")
    return acat_pval(p, w, throw_error=True)


def pACAT2(p: Sequence[float], maf: Sequence[float], wbeta=(1, 25)) -> float:
    """``pACAT2(p, maf, wbeta)`` (R/saige_main.r:150-156): ACAT with weights dbeta(maf)^2 maf (1 - maf)."""
    p, maf = np.asarray(p, dtype=np.float64), np.asarray(maf, dtype=np.float64)
    if len(wbeta) != 2:
        raise ValueError("length(wbeta) == 2L is not TRUE")
    if p.size != maf.size:
        raise ValueError("length(p) == length(maf) is not TRUE")
    w = _dbeta(maf, float(wbeta[0]), float(wbeta[1]))
    return pACAT(p, w * w * maf * (1 - maf))


def _mean_sd(x: np.ndarray):
    """``f64_mean_sd``: mean and sample sd over finite values."""
    x = x[np.isfinite(x)]
    if x.size == 0:
        return float("nan"), float("nan")
    return float(np.mean(x)), (float(np.std(x, ddof=1)) if x.size > 1 else float("nan"))


def _minmax(x: np.ndarray):
    x = x[np.isfinite(x)]
    if x.size == 0:
        return float("nan"), float("nan")
    return float(np.min(x)), float(np.max(x))


class _Units:
    """``SeqUnitListClass``: ``units$desp`` (columns) and ``units$index`` (1-based variant
    indices per unit).  Accepts a dict with keys 'desp'/'index' or a plain list of index lists."""

    def __init__(self, units):
        if isinstance(units, dict):
            self.desp = dict(units.get("desp", {}))
            self.index = [np.asarray(ix, dtype=np.int64) for ix in units["index"]]
        else:
            self.index = [np.asarray(ix, dtype=np.int64) for ix in units]
            self.desp = {}
        if not self.index:
            raise TypeError("inherits(units, \"SeqUnitListClass\") is not TRUE")


def _check_beta(wbeta) -> np.ndarray:
    """``.check_beta``: a length-two vector or a matrix with two rows -> [2, n_weights]."""
    wb = np.asarray(wbeta, dtype=np.float64)
    if not np.all(np.isfinite(wb)):
        raise TypeError("is.finite(wbeta) is not TRUE")
    if wb.ndim == 1:
        if wb.size != 2:
            raise ValueError("'wbeta' should be a length-two vector or a matrix with two rows.")
        wb = wb.reshape(2, 1)
    elif wb.ndim != 2 or wb.shape[0] != 2 or wb.shape[1] <= 0:
        raise ValueError("'wbeta' should be a length-two vector or a matrix with two rows.")
    return wb


class _Prepared:
    pass


def _prepare(gdsfile, modobj, units, wbeta, spa_pval, var_ratio, verbose, what, scanner_factory=None) -> _Prepared:
    """Common head of the three drivers (R/assoc_aggregate.r:55-150): checks, sample matching,
    model, and the per-variant scan of every variant that occurs in a unit."""
    if verbose:
        print(what)
    pr = _Prepared()
    pr.units = _Units(units)
    pr.wbeta = _check_beta(wbeta)
    for nm, v in (("spa.pval", spa_pval), ("var.ratio", var_ratio)):
        if not _is_num(v):
            raise TypeError(f"is.numeric({nm}) is not TRUE")
    mod: NullModel = load_modobj(modobj, verbose)
    src = _open_source(gdsfile, verbose)
    gsid = [str(s) for s in src.sample_id()]
    pos = {str(s): i for i, s in enumerate(mod.sample_id)}
    sel = [i for i, s in enumerate(gsid) if s in pos]
    if len(sel) != len(mod.sample_id):
        raise ModelError("Some of sample IDs are not available in the GDS file.")
    ii = np.array([pos[gsid[i]] for i in sel], dtype=np.int64)
    if isinstance(src, GenotypeSource):
        if src.packed is None:
            raise NotImplementedError("aggregate tests take 2-bit genotypes ($dosage_alt)")
        packed_all, n_all = src.packed, len(gsid)
    else:
        packed_all, n_all, _ = src.dosage_alt_packed()
    n_var_all = packed_all.shape[0]
    used = np.unique(np.concatenate(pr.units.index))
    if used.size == 0 or used.min() < 1 or used.max() > n_var_all:
        raise ValueError("No variant in the genotypic data set!")
    if len(sel) == n_all and np.array_equal(sel, np.arange(n_all)):
        packed = np.ascontiguousarray(packed_all[used - 1])
    else:
        packed = pack_dosage_2bit(unpack_dosage_2bit(packed_all[used - 1], n_all)[:, sel])
    pr.local = {int(v): k for k, v in enumerate(used)}        # 1-based variant index -> row of `packed`
    pr.packed = packed
    if not math.isfinite(var_ratio):
        var_ratio = float(np.nanmean(mod.var_ratio))
    sizes = np.array([len(ix) for ix in pr.units.index])
    if verbose:
        print(f"    # of samples: {len(sel):,}")
        print(f"    # of variants in total: {used.size:,}")
        print(f"    # of units: {len(pr.units.index):,}")
        print(f"    avg. # of variants per unit: {sizes.mean()}")
        print(f"    min # of variants in a unit: {sizes.min()}")
        print(f"    max # of variants in a unit: {sizes.max()}")
        print(f"    p-value threshold for SPA adjustment: {spa_pval}")
        print(f"    variance ratio for approximation: {var_ratio}")
        print("    variant weights: " + ", ".join(f"beta({a:g},{b:g})" for a, b in pr.wbeta.T))
    pr.mod = mod
    # .init_nullmod(modobj, ii, 0, 0, 1, spa.pval, var.ratio, ...): maf 0, mac 0, missing 1
    pr.sm = init_nullmod(mod, ii, 0.0, 0.0, 1.0, spa_pval, var_ratio)
    pr.binary = mod.trait_type == "binary"
    pr.wb_colnm = [f"b{a:g}_{b:g}" for a, b in pr.wbeta.T]
    pr.sc = Scanner(pr.sm) if scanner_factory is None else scanner_factory(pr.sm)   # tests inject the CPU oracle
    if verbose:
        print("Calculating p-values:")
    out, valid = pr.sc.scan_2bit(packed)
    # ds_mat_mafmac (src/saige_main.cpp:466-524), RAW branch: n non-missing, s = sum of codes
    num = out[:, 2]
    n_all_missing = ~np.isfinite(out[:, 0]) & (valid == 0)
    # a variant rejected by the scan's filter (monomorphic: maf = 0) still has counts: redo them on the host
    codes = None
    if (valid == 0).any():
        codes = unpack_dosage_2bit(packed[valid == 0], len(sel)).astype(np.int64)
    n = np.where(valid != 0, num, 0).astype(np.float64)
    s = np.where(valid != 0, np.rint(out[:, 0] * 2 * np.where(valid != 0, num, 0)), 0.0)
    if codes is not None:
        nm = (codes != 3).sum(axis=1)
        sm_ = np.where(codes != 3, codes, 0).sum(axis=1)
        n[valid == 0] = nm
        s[valid == 0] = sm_
    del n_all_missing
    with np.errstate(invalid="ignore", divide="ignore"):
        af = s / (2 * n)
    pr.n, pr.s = n, s
    pr.maf = np.where(n > 0, np.minimum(af, 1 - af), np.nan)
    pr.mac = np.minimum(s, 2 * n - s)
    pr.pval = np.where(valid != 0, out[:, 5], np.nan)          # single_test_*: NaN when the filter rejects
    return pr


def _summary_cols(pr: _Prepared) -> Dict[str, Any]:
    ans: Dict[str, Any] = dict(pr.units.desp)
    rows = [[pr.local[int(v)] for v in ix] for ix in pr.units.index]
    pr.rows = rows
    ans["numvar"] = np.array([len(r) for r in rows])
    st = np.array([[*_mean_sd(pr.maf[r]), *_minmax(pr.maf[r]), *_mean_sd(pr.mac[r]), *_minmax(pr.mac[r])] for r in rows])
    for k, nm in enumerate(("maf.avg", "maf.sd", "maf.min", "maf.max", "mac.avg", "mac.sd", "mac.min", "mac.max")):
        ans[nm] = st[:, k]
    return ans


def _burden_rows(pr: _Prepared, weight_sets: List[List[np.ndarray]]):
    """weight_sets[u][k] = per-variant weights (NaN = not in the burden) of unit u, row kind k.
    Returns the scan rows [n_units, n_kinds, 8] and validity; ``f64_normalize`` and the tables of
    ``ds_mat_burden`` (RAW branch) are formed here."""
    row_ptr, var_idx, lut = [0], [], []
    for u, sets in enumerate(weight_sets):
        r = np.asarray(pr.rows[u])
        for w in sets:
            w = np.asarray(w, dtype=np.float64).copy()
            fin = np.isfinite(w)
            sm = w[fin].sum()
            if sm > 0:
                w[fin] = w[fin] * (1 / sm)                         # f64_normalize
            for j in np.flatnonzero(fin):
                v = r[j]
                n, s = pr.n[v], pr.s[v]
                m = s / n if n > 0 else float("nan")             # (double)sum / n
                if s <= n:
                    t = [0 * w[j], 1 * w[j], 2 * w[j], m * w[j]]
                else:
                    t = [2 * w[j], 1 * w[j], 0 * w[j], (2 - m) * w[j]]
                var_idx.append(v)
                lut.append(t)
            row_ptr.append(len(var_idx))
    lut_a = np.asarray(lut, dtype=np.float64).reshape(-1, 4)
    out, valid = pr.sc.burden_2bit(pr.packed, np.asarray(row_ptr), np.asarray(var_idx, dtype=np.int32), lut_a)
    nk = len(weight_sets[0]) if weight_sets else 0
    return out.reshape(len(weight_sets), nk, 8), valid.reshape(len(weight_sets), nk)


def _row_result(o, ok, n_snp, summac_thr):
    """[summac, beta, SE, pval, p.norm, cvg] of one burden row: saige_burden_test_bin, :640-665."""
    summac = (2 * o[2] * o[0]) * n_snp if np.isfinite(o[0]) else 0.0   # f64_sum(G) * n_snp
    if not ok or not (summac >= summac_thr and summac > 0):
        return summac, float("nan"), float("nan"), float("nan"), float("nan"), 0.0
    return summac, o[3], o[4], o[5], o[6], o[7]


def seqAssocGLMM_spaBurden(gdsfile, modobj, units, wbeta=AggrParamBeta, summac: float = 3, dsnode: str = "",
                           spa_pval: float = 0.05, var_ratio: float = float("nan"), res_savefn: str = "",
                           res_compress: str = "LZMA", parallel=False, verbose: bool = True,
                           verbose_maf: bool = True, scanner_factory=None) -> Dict[str, Any]:
    """Burden tests per unit and weight set (R/assoc_aggregate.r:51-302)."""
    if not (_is_num(summac) and math.isfinite(summac)):
        raise TypeError("is.finite(summac) is not TRUE")
    pr = _prepare(gdsfile, modobj, units, wbeta, spa_pval, var_ratio, verbose, "SAIGE burden analysis:", scanner_factory)
    try:
        ans = _summary_cols(pr)
        sets = [[_dbeta(pr.maf[r], a, b) for a, b in pr.wbeta.T] for r in pr.rows]
        out, valid = _burden_rows(pr, sets)
    finally:
        pr.sc.close()
    for i, nm in enumerate(pr.wb_colnm):
        sfx = f".{nm}" if len(pr.wb_colnm) > 1 else ""
        res = np.array([_row_result(out[u, i], valid[u, i], len(pr.rows[u]), summac) for u in range(len(pr.rows))])
        ans["summac" + sfx], ans["beta" + sfx], ans["SE" + sfx], ans["pval" + sfx] = res[:, 0], res[:, 1], res[:, 2], res[:, 3]
        if pr.binary:
            ans["p.norm" + sfx] = res[:, 4]
            ans["cvg" + sfx] = res[:, 5] != 0
    _save(ans, res_savefn, res_compress, verbose)
    return ans


def _acatv(pr: _Prepared, burden_mac: float, burden_summac: float):
    """ACAT-V per unit and weight set (saige_acatv_test_bin, src/saige_main.cpp:720-830):
    returns p [n_units, n_w], and (n.single, n.burden, median/min/max of the combined p-values)."""
    nu, nw = len(pr.rows), pr.wbeta.shape[1]
    sets = []
    for r in pr.rows:
        rare = pr.mac[r] < burden_mac            # mac >= threshold -> single-variant test
        sets.append([np.where(rare, _dbeta(pr.maf[r], a, b), np.nan) for a, b in pr.wbeta.T])
    has_b = [bool(np.isfinite(s[0]).any()) for s in sets]
    out, valid = _burden_rows(pr, sets)
    p = np.full((nu, nw), np.nan)
    extra = np.full((nu, 2 + 3 * nw), np.nan)
    for u, r in enumerate(pr.rows):
        r = np.asarray(r)
        single = pr.mac[r] >= burden_mac
        n_burden = int((~single).sum())
        for i, (a, b) in enumerate(pr.wbeta.T):
            mf = pr.maf[r][single]
            pv = list(pr.pval[r][single])
            wp = list(_dbeta(mf, a, b) ** 2 * mf * (1 - mf))
            if has_b[u]:
                sm, _, _, pb, _, _ = _row_result(out[u, i], valid[u, i], len(r), burden_summac)
                if np.isfinite(pb):
                    pm = pr.maf[r][~single].sum() / n_burden
                    wp.append(float(_dbeta(pm, a, b)) ** 2 * pm * (1 - pm))
                    pv.append(pb)
            if i == 0:
                extra[u, 0] = len(pv) - n_burden
                extra[u, 1] = n_burden
            p[u, i] = acat_pval(pv, wp) if pv else float("nan")
            fin = np.asarray(pv, dtype=np.float64)
            fin = fin[np.isfinite(fin)]
            if fin.size:
                extra[u, 2 + 3 * i: 5 + 3 * i] = [np.median(fin), fin.min(), fin.max()]
    return p, extra


def seqAssocGLMM_spaACAT_V(gdsfile, modobj, units, wbeta=AggrParamBeta, burden_mac: float = 10,
                           burden_summac: float = 3, dsnode: str = "", spa_pval: float = 0.05,
                           var_ratio: float = float("nan"), res_savefn: str = "", res_compress: str = "LZMA",
                           parallel=False, verbose: bool = True, verbose_maf: bool = True, scanner_factory=None) -> Dict[str, Any]:
    """ACAT-V tests (R/assoc_aggregate.r:309-557); binary outcomes only, as in the reference."""
    pr = _prepare(gdsfile, modobj, units, wbeta, spa_pval, var_ratio, verbose, "SAIGE ACAT-V analysis:", scanner_factory)
    try:
        if not pr.binary:
            raise NotImplementedError("'saige_acatv_test_quant' not implemented.")
        ans = _summary_cols(pr)
        p, extra = _acatv(pr, burden_mac, burden_summac)
    finally:
        pr.sc.close()
    ans["n.single"], ans["n.burden"] = extra[:, 0].astype(np.int64), extra[:, 1].astype(np.int64)
    for i, nm in enumerate(pr.wb_colnm):
        sfx = f".v{nm[1:]}" if len(pr.wb_colnm) > 1 else ""      # wb_colnm = "v%g_%g", R/assoc_aggregate.r:389
        ans["pval" + sfx] = p[:, i]
        ans["p.med" + sfx], ans["p.min" + sfx], ans["p.max" + sfx] = extra[:, 2 + 3 * i], extra[:, 3 + 3 * i], extra[:, 4 + 3 * i]
    _save(ans, res_savefn, res_compress, verbose)
    return ans


def seqAssocGLMM_spaACAT_O(gdsfile, modobj, units, wbeta=AggrParamBeta, burden_mac: float = 10,
                           burden_summac: float = 3, dsnode: str = "", spa_pval: float = 0.05,
                           var_ratio: float = float("nan"), res_savefn: str = "", res_compress: str = "LZMA",
                           parallel=False, verbose: bool = True, verbose_maf: bool = True, scanner_factory=None) -> Dict[str, Any]:
    """ACAT-O: burden and ACAT-V p-values of every weight set combined by ACAT
    (R/assoc_aggregate.r:564-797, saige_acato_test_bin src/saige_main.cpp:845-976)."""
    pr = _prepare(gdsfile, modobj, units, wbeta, spa_pval, var_ratio, verbose, "SAIGE ACAT-O analysis:", scanner_factory)
    try:
        if not pr.binary:
            raise NotImplementedError("'saige_acato_test_quant' not implemented.")
        ans = _summary_cols(pr)
        sets = [[_dbeta(pr.maf[r], a, b) for a, b in pr.wbeta.T] for r in pr.rows]
        out, valid = _burden_rows(pr, sets)
        pb = np.array([[_row_result(out[u, i], valid[u, i], len(pr.rows[u]), burden_summac)[3]
                        for i in range(pr.wbeta.shape[1])] for u in range(len(pr.rows))])
        pv, _ = _acatv(pr, burden_mac, burden_summac)
    finally:
        pr.sc.close()
    both = np.empty((pb.shape[0], 2 * pb.shape[1]))
    both[:, 0::2], both[:, 1::2] = pb, pv
    ans["pval"] = np.array([acat_pval(row) for row in both])
    for i, nm in enumerate(pr.wb_colnm):
        ans["pval.b" + nm[1:]] = pb[:, i]
        ans["pval.v" + nm[1:]] = pv[:, i]
    _save(ans, res_savefn, res_compress, verbose)
    return ans


def _save(ans, res_savefn, res_compress, verbose):
    if not res_savefn:
        if verbose:
            print("Done.")
        return
    import re
    from .results import save_result
    if re.search(r"\.(rda|RData)$", res_savefn, re.I):
        if verbose:
            print(f"Save to '{res_savefn}' ...")
        save_result({k: v for k, v in ans.items()}, res_savefn, "ZIP" if res_compress == "none" else res_compress)
    else:
        raise ValueError("Unknown format of the output file, and it should be RData or gds.")
