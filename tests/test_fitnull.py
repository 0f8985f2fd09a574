"""Null-model fit (``seqFitNullGLMM_SPA`` mirror, saigegds_amd/fitnull.py) against the
reference's golden models -- the check of inst/unitTests/test_SAIGE.R:46-76
(``checkEquals(mod, glmm, tolerance=1e-4)``, i.e. mean relative difference per field).

CPU tests inject the oracle's GRM operator (test infrastructure) to exercise the
host logic; the GPU tests run the product path (``GrmOperator`` -> sgx_grm_*).
"""
import os

import numpy as np
import pytest

from saigegds_amd.assoc import GenotypeSource
from saigegds_amd.fitnull import RRandom, glm_fit, seqFitNullGLMM_SPA
from saigegds_amd.nullmod import load_modobj

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-4          # test_SAIGE.R:68-75


def _inputs():
    g = np.load(os.path.join(GOLD, "grm1k_10k_snp.npz"))
    ph = np.load(os.path.join(GOLD, "pheno.npz"))
    src = GenotypeSource(list(g["sample_id"]), packed=g["packed"], variant_id=g["variant_id"])
    data = {"sample.id": ph["sample_id"], "y": ph["y"], "yy": ph["yy"], "x1": ph["x1"], "x2": ph["x2"]}
    return src, data


def _mean_rel(a, b):
    """R's all.equal.numeric: sum|a-b| / sum|b| (absolute when the target is ~0)."""
    a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
    assert a.shape == b.shape
    xy, xn = np.sum(np.abs(a - b)), np.sum(np.abs(b))
    return xy / xn if xn > 1e-300 else xy


def _check_model(m, gold, tol=TOL):
    assert m.trait_type == str(gold["trait_type"])
    assert list(m.sample_id) == list(gold["sample_id"])
    assert np.array_equal(np.asarray(m.variant_id), gold["variant_id"])
    assert bool(m.converged) == bool(gold["converged"])
    for name, val in [("tau", m.tau), ("coefficients", m.coefficients), ("fitted_values", m.fitted_values),
                      ("linear_predictors", m.linear_predictors), ("residuals", m.residuals), ("cov", m.cov),
                      ("y", m.y), ("V", m.V), ("X1", m.X1), ("XV", m.XV), ("XXVX_inv", m.XXVX_inv),
                      ("noK_mu", m.mu_noK), ("noK_res", m.res_noK)]:
        assert _mean_rel(val, gold[name]) < tol, name
    # the random marker choice is part of the model: same ids in the same order
    assert np.array_equal(np.asarray(m.var_ratio_table["id"], dtype=np.int64), gold["vr_id"].astype(np.int64))
    for k in ("maf", "mac", "var1", "var2", "ratio"):
        assert _mean_rel(m.var_ratio_table[k], gold["vr_" + k]) < tol, k


def test_r_random_stream():
    # known answers of R (RNGkind Mersenne-Twister / Inversion / Rounding):
    #   set.seed(42); runif(3)   set.seed(1); runif(3)   set.seed(123); runif(3)
    #   set.seed(42); sample(10)   (R < 3.6 / sample.kind="Rounding")
    for seed, want in [(42, [0.914806, 0.9370754, 0.2861395]), (1, [0.2655087, 0.3721239, 0.5728534]),
                       (123, [0.2875775, 0.7883051, 0.4089769])]:
        r = RRandom(seed)
        assert np.allclose([r.unif_rand() for _ in range(3)], want, atol=5e-8, rtol=0)
    assert RRandom(42).sample_int(10).tolist() == [10, 9, 3, 6, 4, 8, 5, 1, 2, 7]
    r = RRandom(7)
    u = [r.unif_rand() for _ in range(700)]           # crosses a state regeneration
    r.set_seed(7)
    assert r.rbinom1_half(700).tolist() == [1.0 if v >= 0.5 else 0.0 for v in u]


def test_glm_start_values_match_golden_noK():
    # obj.noK of the golden model is the no-GRM glm fit on the QR-transformed design
    src, data = _inputs()
    gold = np.load(os.path.join(GOLD, "saige_model.npz"))
    f = glm_fit(gold["X1"], data["y"].astype(np.float64), "binomial")
    assert _mean_rel(f.fitted_values, gold["noK_mu"]) < 1e-9
    assert _mean_rel(f.fitted_values * (1 - f.fitted_values), gold["V"]) < 1e-9


def test_argument_checks():
    src, data = _inputs()
    with pytest.raises(ValueError, match="no 'nope'"):
        seqFitNullGLMM_SPA("nope ~ x1", data, src, verbose=False, operator_factory=lambda p, n: None)
    with pytest.raises(ValueError, match="should not be in the formula"):
        seqFitNullGLMM_SPA("y ~ sample.id", data, src, verbose=False, operator_factory=lambda p, n: None)
    with pytest.raises(ValueError, match="should be one of"):
        seqFitNullGLMM_SPA("y ~ x1", data, src, trait_type="ordinal", verbose=False)
    d2 = dict(data)
    d2["sample.id"] = np.array(["s1"] * len(data["y"]))
    with pytest.raises(ValueError, match="should be unique"):
        seqFitNullGLMM_SPA("y ~ x1", d2, src, verbose=False, operator_factory=lambda p, n: None)


@pytest.mark.parametrize("trait", ["binary", "quantitative"])
def test_fit_host_logic_with_oracle_operator(trait, tmp_path):
    from oracle.oracle import GrmOracle
    src, data = _inputs()
    fn = str(tmp_path / "model.rds")
    if trait == "binary":
        m = seqFitNullGLMM_SPA("y ~ x1 + x2", data, src, verbose=False, operator_factory=GrmOracle, model_savefn=fn)
        gold = np.load(os.path.join(GOLD, "saige_model.npz"))
    else:
        m = seqFitNullGLMM_SPA("yy ~ x1 + x2", data, src, trait_type="quantitative", verbose=False,
                               operator_factory=GrmOracle, model_savefn=fn)
        gold = np.load(os.path.join(GOLD, "saige_model_quant.npz"))
    _check_model(m, gold)
    # the saved file is a ClassSAIGE_NullModel the scan accepts
    back = load_modobj(fn)
    assert back.trait_type == trait and list(back.sample_id) == list(m.sample_id)
    for k in ("tau", "fitted_values", "var_ratio", "y", "V", "X1", "XV", "XXVX_inv", "coefficients"):
        assert np.array_equal(np.asarray(getattr(back, k)), np.asarray(getattr(m, k))), k
    assert np.array_equal(np.asarray(back.variant_id), np.asarray(m.variant_id))


@pytest.mark.gpu
@pytest.mark.parametrize("trait", ["binary", "quantitative"])
def test_fit_on_gpu_matches_golden_model(trait):
    src, data = _inputs()
    if trait == "binary":
        m = seqFitNullGLMM_SPA("y ~ x1 + x2", data, src, verbose=False)
        gold = np.load(os.path.join(GOLD, "saige_model.npz"))
    else:
        m = seqFitNullGLMM_SPA("yy ~ x1 + x2", data, src, trait_type="quantitative", verbose=False)
        gold = np.load(os.path.join(GOLD, "saige_model_quant.npz"))
    _check_model(m, gold)


@pytest.mark.gpu
def test_fit_then_scan_end_to_end_on_gpu():
    """fit on the GPU, scan with the fitted model, compare with the golden p-values
    (test.saige_fit_null_model followed by test.saige_pval)."""
    from saigegds_amd import seqAssocGLMM_SPA
    src, data = _inputs()
    g = np.load(os.path.join(GOLD, "grm1k_10k_snp.npz"))
    full = GenotypeSource(list(g["sample_id"]), packed=g["packed"], variant_id=g["variant_id"],
                          chromosome=list(g["chromosome"]), position=g["position"], rs_id=list(g["rs_id"]),
                          ref=list(g["ref"]), alt=list(g["alt"]))
    m = seqFitNullGLMM_SPA("y ~ x1 + x2", data, src, verbose=False)
    res = seqAssocGLMM_SPA(full, m, mac=4, verbose=False)
    gold = np.load(os.path.join(GOLD, "saige_pval.npz"))
    assert np.array_equal(np.asarray(res["id"]), gold["id"])
    for k, gk in [("AF.alt", "AF_alt"), ("beta", "beta"), ("SE", "SE"), ("pval", "pval")]:
        assert _mean_rel(res[k], gold[gk]) < 1e-7, k       # test_SAIGE.R:97-98


def test_heritability_from_golden_models():
    from saigegds_amd import glmmHeritability
    from saigegds_amd.nullmod import NullModel
    import math
    for name, trait in (("saige_model.npz", "binary"), ("saige_model_quant.npz", "quantitative")):
        m = np.load(os.path.join(GOLD, name))
        mod = NullModel(trait_type=trait, tau=m["tau"], fitted_values=m["fitted_values"], sample_id=list(m["sample_id"]),
                        var_ratio=m["var_ratio"], y=m["y"], V=m["V"], X1=m["X1"], XV=m["XV"], XXVX_inv=m["XXVX_inv"])
        h = glmmHeritability(mod)
        if trait == "binary":
            p = float(np.mean(m["y"] == 1))
            assert h == pytest.approx(m["tau"][1] / (math.pi ** 2 / 3 + m["tau"][1]) * (2.970 + 0.372 * math.log10(p)))
            assert glmmHeritability(mod, adjust=False) == pytest.approx(m["tau"][1] / (math.pi ** 2 / 3 + m["tau"][1]))
        else:
            assert h == pytest.approx(m["tau"][1] / m["tau"].sum())
