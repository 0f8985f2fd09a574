"""CPU restatement of the reference's aggregate tests -- TEST INFRASTRUCTURE ONLY.

Follows src/saige_main.cpp line by line, one unit at a time, from the unit's dosage matrix
(RAW semantics: 0, 1, 2, 0xFF = missing), with the single-variant test delegated to the scan
oracle (oracle/saige_oracle.c, ``single_test_bin/quant``):

  ds_mat_mafmac          :466-524        burden_test_bin/quant   :615-705
  ds_mat_burden          :526-610        acatv_test_bin          :720-830
  acat_pval              :1001-1052      acato_test_bin          :845-976

Only tests/ may import this (see oracle/oracle.py).
"""
import math

import numpy as np

ROUND_ZERO = 1e-300
ROUND_ONE = 1 - 1e-16


def dbeta(x, a, b):
    from scipy.stats import beta
    return float(beta.pdf(x, a, b))


def acat_pval(pval, w):
    sumw = sum(wi for p, wi in zip(pval, w) if math.isfinite(p) and math.isfinite(wi))
    if sumw <= 0:
        return float("nan")
    t = 0.0
    for p, wi in zip(pval, w):
        if math.isfinite(p) and math.isfinite(wi):
            if p < 0 or p > 1:
                return float("nan")
            p = ROUND_ZERO if p < ROUND_ZERO else (ROUND_ONE if p > ROUND_ONE else p)
            t += wi * math.tan(math.pi * (0.5 - p)) if p >= 1e-15 else wi / p / math.pi
    t /= sumw
    return 0.5 - math.atan(t) / math.pi if t <= 5e14 else 1.0 / t / math.pi


def ds_mat_mafmac(ds):
    """ds: [n_snp, n_samp] uint8."""
    maf, mac = [], []
    for row in ds:
        ok = row != 0xFF
        n, s = int(ok.sum()), int(row[ok].sum())
        af = s / (2 * n) if n > 0 else float("nan")
        maf.append(min(af, 1 - af) if n > 0 else float("nan"))
        mac.append(float(min(s, 2 * n - s)))
    return np.array(maf), np.array(mac)


def normalize(w):
    w = np.array(w, dtype=np.float64)
    fin = np.isfinite(w)
    sm = w[fin].sum()
    if sm > 0:
        w[fin] *= 1 / sm
    return w


def ds_mat_burden(ds, weight):
    out = np.zeros(ds.shape[1])
    for row, w in zip(ds, weight):
        if not math.isfinite(w):
            continue
        ok = row != 0xFF
        n, s = int(ok.sum()), int(row[ok].sum())
        m = s / n
        r = row.astype(np.float64)
        if s <= n:
            out += np.where(ok, r * w, m * w)
        else:
            m = 2 - m
            out += np.where(ok, (2 - r) * w, m * w)
    return out


def single(oracle, G):
    """single_test_* on one dosage vector -> (beta, SE, pval, pval_noadj, converged) or NaNs."""
    out, valid = oracle.scan_f64(np.ascontiguousarray(G[None, :]))
    if not valid[0]:
        return (float("nan"),) * 4 + (False,)
    o = out[0]
    return o[3], o[4], o[5], o[6], bool(o[7] != 0) if math.isfinite(o[7]) else False


def burden_unit(oracle, ds, wbeta, summac_thr):
    maf, _ = ds_mat_mafmac(ds)
    res = []
    for b1, b2 in wbeta.T:
        ws = normalize([dbeta(m, b1, b2) for m in maf])
        G = ds_mat_burden(ds, ws)
        summac = G.sum() * len(ds)
        r = (float("nan"),) * 4 + (False,)
        if summac >= summac_thr and summac > 0:
            r = single(oracle, G)
        res.append((summac,) + r)
    return res


def acatv_unit(oracle, ds, wbeta, acatv_mac, summac_thr):
    maf, mac = ds_mat_mafmac(ds)
    ps = []
    for b1, b2 in wbeta.T:
        w_pval, pvals, w_burden = [], [], []
        n_burden, summaf = 0, 0.0
        for j in range(len(ds)):
            if mac[j] >= acatv_mac:
                G = ds[j].astype(np.float64)
                G[ds[j] == 0xFF] = np.nan
                pv = single(oracle, G)[2]
                p = maf[j]
                w_pval.append(dbeta(p, b1, b2) ** 2 * p * (1 - p))
                pvals.append(pv)
                w_burden.append(float("nan"))
            else:
                n_burden += 1
                summaf += maf[j]
                w_burden.append(dbeta(maf[j], b1, b2))
        if n_burden > 0:
            G = ds_mat_burden(ds, normalize(w_burden))
            summac = G.sum() * len(ds)
            if summac >= summac_thr and summac > 0:
                pv = single(oracle, G)[2]
                if math.isfinite(pv):
                    p = summaf / n_burden
                    w_pval.append(dbeta(p, b1, b2) ** 2 * p * (1 - p))
                    pvals.append(pv)
        ps.append(acat_pval(pvals, w_pval) if pvals else float("nan"))
    return ps


def acato_unit(oracle, ds, wbeta, acatv_mac, summac_thr):
    pb = [r[3] for r in burden_unit(oracle, ds, wbeta, summac_thr)]
    pv = acatv_unit(oracle, ds, wbeta, acatv_mac, summac_thr)
    both = []
    for a, b in zip(pb, pv):
        both += [a, b]
    return acat_pval(both, [1.0] * len(both)), pb, pv
