"""Aggregate tests (saigegds_amd/aggregate.py) against the CPU restatement of
src/saige_main.cpp:466-1052 (oracle/aggregate_oracle.py), plus the reference's own checks:
test.saige_acta_o (ACAT-O's burden / ACAT-V columns equal the stand-alone tests) and test.pACAT
(inst/unitTests/test_SAIGE.R:109-165)."""
import math
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _sliding_units(n_var, win=200, shift=100):
    """seqUnitSlidingWindows on variant index (the fixture's positions are 1..M)."""
    idx, st = [], 0
    while st < n_var:
        idx.append(np.arange(st, min(n_var, st + win)) + 1)
        st += shift
    return idx


def test_pacat_matches_r_formula():
    # test.pACAT: the R implementation of the ACAT p-value
    from saigegds_amd.aggregate import pACAT
    ps = 10.0 ** -np.arange(1, 15.01, 0.5)
    for p1 in ps:
        for p2 in ps[::3]:
            t = 0.5 * (math.tan((0.5 - p1) * math.pi) + math.tan((0.5 - p2) * math.pi))
            ref = 0.5 - math.atan(t) / math.pi
            got = pACAT([p1, p2])
            assert got == pytest.approx(ref, rel=1e-6 if min(p1, p2) > 1e-12 else 1e-3)
    assert pACAT([0.3]) == 0.3
    with pytest.raises(ValueError):
        pACAT([])
    with pytest.raises(ValueError):
        pACAT([0.1, 2.0])
    assert pACAT([1e-20, 0.5], [1.0, 1.0]) == pytest.approx(2e-20, rel=1e-6)
    from saigegds_amd.aggregate import pACAT2, _dbeta
    pp, mf = np.array([0.01, 0.2, 0.5]), np.array([0.001, 0.01, 0.2])
    w = _dbeta(mf, 1, 25)
    assert pACAT2(pp, mf) == pytest.approx(pACAT(pp, w * w * mf * (1 - mf)))


def test_acat_oracle_agrees_with_driver_combination():
    from oracle.aggregate_oracle import acat_pval as oacat
    from saigegds_amd.aggregate import acat_pval
    rng = np.random.default_rng(3)
    for _ in range(50):
        p = rng.random(7) ** 4
        w = rng.random(7)
        p[2] = np.nan
        assert acat_pval(p, w) == pytest.approx(oacat(list(p), list(w)), rel=1e-14)


@pytest.mark.parametrize("device", [pytest.param("gpu", marks=pytest.mark.gpu), "oracle"])
def test_aggregate_tests_match_cpu_restatement(device):
    """device = gpu: the product path; device = oracle: the driver's host logic (batching, CSR,
    tables, weights, combination) with the scan oracle in place of the library (no GPU)."""
    from oracle.aggregate_oracle import acato_unit, acatv_unit, burden_unit
    from oracle.oracle import Oracle
    from saigegds_amd.aggregate import (AggrParamBeta, seqAssocGLMM_spaACAT_O, seqAssocGLMM_spaACAT_V,
                                        seqAssocGLMM_spaBurden)
    from saigegds_amd.assoc import GenotypeSource
    from saigegds_amd.gds import unpack_dosage_2bit
    from saigegds_amd.nullmod import NullModel, init_nullmod
    g = np.load(os.path.join(GOLD, "grm1k_10k_snp.npz"))
    m = np.load(os.path.join(GOLD, "saige_model.npz"))
    mod = NullModel(trait_type="binary", tau=m["tau"], fitted_values=m["fitted_values"], sample_id=list(m["sample_id"]),
                    var_ratio=m["var_ratio"], y=m["y"], V=m["V"], X1=m["X1"], XV=m["XV"], XXVX_inv=m["XXVX_inv"])
    nv = 1500 if device == "gpu" else 500
    src = GenotypeSource(list(g["sample_id"]), packed=g["packed"][:nv], variant_id=g["variant_id"][:nv])
    units = _sliding_units(nv)
    kw = {}
    if device == "gpu":
        import torch  # noqa: F401
    else:
        from oracle.oracle import OracleScanner
        kw["scanner_factory"] = OracleScanner
    b = seqAssocGLMM_spaBurden(src, mod, units, verbose=False, **kw)
    v = seqAssocGLMM_spaACAT_V(src, mod, units, verbose=False, **kw)
    o = seqAssocGLMM_spaACAT_O(src, mod, units, verbose=False, **kw)
    # the reference's own check (test.saige_acta_o)
    for k in ("1_1", "1_25"):
        assert np.array_equal(o["pval.b" + k], b["pval.b" + k], equal_nan=True)
        assert np.array_equal(o["pval.v" + k], v["pval.v" + k], equal_nan=True)
    # unit by unit against the restatement
    sm = init_nullmod(mod, np.arange(1000), 0.0, 0.0, 1.0, 0.05, float(np.mean(m["var_ratio"])))
    orc = Oracle(sm)
    codes = unpack_dosage_2bit(g["packed"][:nv], 1000)
    ds_all = np.where(codes == 3, 0xFF, codes).astype(np.uint8)
    wb = AggrParamBeta

    def close(a, c, what):
        a, c = np.asarray(a, dtype=np.float64), np.asarray(c, dtype=np.float64)
        assert np.array_equal(np.isnan(a), np.isnan(c)), what
        ok = ~np.isnan(a)
        assert np.all(np.abs(a[ok] - c[ok]) <= 1e-9 * np.abs(c[ok]) + 1e-300), what

    for u in sorted({0, 3, min(7, len(units) - 1), len(units) - 1}):
        ds = ds_all[units[u] - 1]
        rb = burden_unit(orc, ds, wb, 3)
        for i, k in enumerate(("b1_1", "b1_25")):
            close([b["summac." + k][u], b["beta." + k][u], b["SE." + k][u], b["pval." + k][u], b["p.norm." + k][u]],
                  rb[i][:5], f"burden unit {u} {k}")
            assert bool(b["cvg." + k][u]) == bool(rb[i][5]) or math.isnan(rb[i][3])
        close([v["pval.v1_1"][u], v["pval.v1_25"][u]], acatv_unit(orc, ds, wb, 10, 3), f"ACAT-V unit {u}")
        po, pb, pv = acato_unit(orc, ds, wb, 10, 3)
        close([o["pval"][u]], [po], f"ACAT-O unit {u}")
    assert b["numvar"][0] == 200 and np.isfinite(b["maf.avg"]).all() and (v["n.burden"] >= 0).all()


@pytest.mark.gpu
def test_burden_rows_equal_single_scan_for_one_variant_units():
    """A unit of one variant with weight 1 collapses to that variant's (minor-oriented, imputed)
    dosage: the burden row must reproduce the single-variant scan of the same variant."""
    import torch  # noqa: F401
    from saigegds_amd._lib import Scanner
    from saigegds_amd.nullmod import NullModel, init_nullmod
    g = np.load(os.path.join(GOLD, "grm1k_10k_snp.npz"))
    m = np.load(os.path.join(GOLD, "saige_model.npz"))
    mod = NullModel(trait_type="binary", tau=m["tau"], fitted_values=m["fitted_values"], sample_id=list(m["sample_id"]),
                    var_ratio=m["var_ratio"], y=m["y"], V=m["V"], X1=m["X1"], XV=m["XV"], XXVX_inv=m["XXVX_inv"])
    sm = init_nullmod(mod, np.arange(1000), 0.0, 0.0, 1.0, 0.05, float(np.mean(m["var_ratio"])))
    packed = g["packed"][:64]
    with Scanner(sm) as sc:
        ref, rv = sc.scan_2bit(packed)
        n = ref[:, 2]
        s = np.rint(ref[:, 0] * 2 * n)
        lut = np.array([[0, 1, 2, s[j] / n[j]] if s[j] <= n[j] else [2, 1, 0, 2 - s[j] / n[j]] for j in range(64)], dtype=np.float64)
        out, valid = sc.burden_2bit(packed, np.arange(65), np.arange(64, dtype=np.int32), lut)
    ok = (rv != 0) & (valid != 0)
    assert ok.sum() > 40
    # beta keeps the sign of the minor-allele orientation in the burden row: compare magnitudes and p-values
    assert np.allclose(np.abs(out[ok, 3]), np.abs(ref[ok, 3]), rtol=1e-9)
    assert np.allclose(out[ok, 5], ref[ok, 5], rtol=1e-9)
    assert np.allclose(out[ok, 1], ref[ok, 1], rtol=0, atol=1e-9)
