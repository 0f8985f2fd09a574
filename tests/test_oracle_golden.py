"""Pins the CPU oracle against the reference's own golden tables
(inst/unitTests/test_SAIGE.R:79-106, test.saige_pval): same models, same
genotypes, mac=4.  The reference checks at 1e-7; the oracle is held to 1e-10."""
import numpy as np
import pytest

from oracle import Oracle, pchisq1_upper, pnorm, qnorm


def _cmp_float(ov, gv, tol=1e-10):
    rel = np.abs(ov - gv) / np.abs(gv)
    assert rel.max() <= tol, rel.max()


def test_binary_golden(grm1k, model_bin, golden_bin):
    o = Oracle(model_bin)
    out, valid = o.scan_2bit(grm1k["packed"])
    assert valid.all()          # mac=4 keeps all 10 000 variants
    assert np.array_equal(out[:, 0], golden_bin["AF_alt"])
    assert np.array_equal(out[:, 1], golden_bin["mac"])
    assert np.array_equal(out[:, 2], golden_bin["num"].astype(float))
    _cmp_float(out[:, 3], golden_bin["beta"])
    _cmp_float(out[:, 4], golden_bin["SE"])
    _cmp_float(out[:, 5], golden_bin["pval"])
    _cmp_float(out[:, 6], golden_bin["p_norm"])
    assert np.array_equal(out[:, 7] == 1, golden_bin["converged"])
    tr = o.trace.as_dict()
    # branch census of the golden set
    assert tr["sparse_path"] == 6305 and tr["dense_path"] == 3695
    assert tr["spa_entered"] == 436 and tr["cutoff_exit"] == 45 and tr["spa_done"] == 391


def test_quant_golden(grm1k, model_quant, golden_quant):
    o = Oracle(model_quant)
    out, valid = o.scan_2bit(grm1k["packed"])
    assert valid.all()
    assert np.array_equal(out[:, 0], golden_quant["AF_alt"])
    assert np.array_equal(out[:, 1], golden_quant["mac"])
    assert np.array_equal(out[:, 2], golden_quant["num"].astype(float))
    _cmp_float(out[:, 3], golden_quant["beta"])
    _cmp_float(out[:, 4], golden_quant["SE"])
    _cmp_float(out[:, 5], golden_quant["pval"])
    assert np.isnan(out[:, 6:]).all()


def test_readme_dosage_file_pin():
    """The reference's printed output for assoc_100snp.gds at mac = 10 (README.md:111-129): 38 survivors; ids 4, 12, 14
    with AF.alt 0.0100 / 0.0150 / 0.0375, mac 20 / 30 / 75, num 1000 (the model of that session is not stored, so its
    beta / SE / pval are not a pin; AF, mac, num and the filter do not depend on the model)."""
    import os
    from conftest import GOLDEN, scan_model
    z = np.load(os.path.join(GOLDEN, "assoc_100snp.npz"))
    sm = scan_model("saige_model.npz", mac=10, sample_ids=[str(s) for s in z["sample_id"]])
    for out, valid in (Oracle(sm).scan_f64(z["dosage_u8"].astype(np.float64)), Oracle(sm).scan_u8(z["dosage_u8"])):
        keep = np.flatnonzero(valid)
        assert keep.size == 38
        assert [int(x) for x in z["variant_id"][keep[:3]]] == [4, 12, 14]
        np.testing.assert_allclose(out[keep[:3], 0], [0.0100, 0.0150, 0.0375], rtol=0, atol=1e-15)
        assert np.array_equal(out[keep[:3], 1], [20.0, 30.0, 75.0]) and np.array_equal(out[keep[:3], 2], [1000.0] * 3)


def test_input_types_agree(grm1k, model_bin):
    """RAW / REAL / 2-bit inputs of get_ds (saige_main.cpp:162-186) give one answer."""
    from saigegds_amd.gds import unpack_dosage_2bit
    pk = grm1k["packed"][:300]
    ds = unpack_dosage_2bit(pk, 1000)
    o = Oracle(model_bin)
    a, va = o.scan_2bit(pk)
    b, vb = o.scan_u8(ds)
    c, vc = o.scan_f64(ds.astype(np.float64))
    assert np.array_equal(a, b, equal_nan=True) and np.array_equal(a, c, equal_nan=True)
    assert np.array_equal(va, vb) and np.array_equal(va, vc)


def test_rmath_standins():
    """erfc/AS241 restatements of Rf_pchisq / Rf_pnorm5 / Rf_qnorm5 vs 40-digit mpmath."""
    import mpmath as mp
    mp.mp.dps = 40
    for x in [1e-6, 0.3, 1.0, 3.84, 10.0, 40.0, 200.0, 1400.0]:
        ex = mp.erfc(mp.sqrt(mp.mpf(x) / 2))
        assert abs(float(pchisq1_upper(x) / ex - 1)) < 1e-13
    for z in [-30.0, -10.0, -3.0, -0.5, 0.0, 0.5, 3.0, 8.0]:
        assert abs(float(pnorm(z, True) / mp.ncdf(z) - 1)) < 1e-13
        assert abs(float(pnorm(z, False) / mp.ncdf(-z) - 1)) < 1e-13
    for p in [1e-300, 1e-100, 1e-20, 1e-8, 1e-3, 0.02, 0.075, 0.3, 0.5, 0.7, 0.999]:
        q = qnorm(p)
        # residual of the defining equation instead of an inverse: Phi(q) == p
        assert abs(float(mp.ncdf(mp.mpf(q)) / mp.mpf(p) - 1)) < 2e-15 * max(1.0, q * q), p
    assert qnorm(0.0) == -np.inf and qnorm(1.0) == np.inf and np.isnan(qnorm(np.nan))
