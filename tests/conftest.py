import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: reads /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir(REFERENCE)
    skip_ref = pytest.mark.skip(reason="/root/reference not present on this machine")
    for it in items:
        if "reference" in it.keywords and not have_ref:
            it.add_marker(skip_ref)


def load_null_model(name):
    from saigegds_amd.nullmod import NullModel
    z = np.load(os.path.join(GOLDEN, name))
    return NullModel(trait_type=str(z["trait_type"]), tau=z["tau"], fitted_values=z["fitted_values"],
                     sample_id=[str(s) for s in z["sample_id"]], var_ratio=z["var_ratio"], y=z["y"],
                     V=z["V"], X1=z["X1"], XV=z["XV"], XXVX_inv=z["XXVX_inv"])


def scan_model(name, mac=4.0, maf=float("nan"), missing=0.1, spa_pval=0.05, sample_ids=None):
    """Flattened model exactly as test.saige_pval builds it (mac=4)."""
    from saigegds_amd.nullmod import init_nullmod
    mod = load_null_model(name)
    if sample_ids is None:
        ii = np.arange(len(mod.sample_id))
    else:
        pos = {s: i for i, s in enumerate(mod.sample_id)}
        ii = np.array([pos[s] for s in sample_ids])
    vr = float(np.nanmean(mod.var_ratio))
    return init_nullmod(mod, ii, maf, mac, missing, spa_pval, vr)


@pytest.fixture(scope="session")
def grm1k():
    z = np.load(os.path.join(GOLDEN, "grm1k_10k_snp.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden_bin():
    z = np.load(os.path.join(GOLDEN, "saige_pval.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden_quant():
    z = np.load(os.path.join(GOLDEN, "saige_pval_quant.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def model_bin():
    return scan_model("saige_model.npz")


@pytest.fixture(scope="session")
def model_quant():
    return scan_model("saige_model_quant.npz")


# ---- comparison helpers shared by CPU and GPU parity tests ----------------
REL_TOL = 1e-10   # north_star: beta / SE / pval within 1e-10 relative


Z_FLOOR = 1e-12  # absolute floor on the z-score beta/SE (see assert_table_close)


def table_errors(out, ref, quant=False):
    """Per-row error of beta, SE, pval[, pval_noadj] in units of the tolerance.

    pval, p.norm: relative error / REL_TOL.
    beta: the score S is a sum that cancels to ~0 under the null, so a purely
    relative bound on beta = S/var is meaningless for |z| -> 0 (the reference's
    own arithmetic moves such rows by more than 1e-10, cf. the long-double
    oracle).  The bound is REL_TOL*|beta| + Z_FLOOR*SE, i.e. an absolute 1e-12 on
    the z-score beta/SE.  SE = |beta/qnorm(p/2)| inherits the same floor divided
    by |z|.
    """
    with np.errstate(invalid="ignore", divide="ignore"):
        b, se = ref[:, 3], ref[:, 4]
        z = np.abs(b) / se
        tol_b = REL_TOL * np.abs(b) + Z_FLOOR * np.where(np.isfinite(se), se, 0.0)   # (SE is NaN where the reference's p is)
        tol_se = se * (REL_TOL + Z_FLOOR / np.maximum(z, 1e-300))
        errs = {"beta": np.abs(out[:, 3] - b) / tol_b, "SE": np.abs(out[:, 4] - se) / tol_se}
        for c, name in ((5, "pval"),) + (() if quant else ((6, "pval_noadj"),)):
            errs[name] = np.abs(out[:, c] - ref[:, c]) / (REL_TOL * np.abs(ref[:, c]))
    for k in errs:
        same = (out[:, {"beta": 3, "SE": 4, "pval": 5, "pval_noadj": 6}[k]] ==
                ref[:, {"beta": 3, "SE": 4, "pval": 5, "pval_noadj": 6}[k]])
        bothnan = np.isnan(out[:, {"beta": 3, "SE": 4, "pval": 5, "pval_noadj": 6}[k]]) & \
            np.isnan(ref[:, {"beta": 3, "SE": 4, "pval": 5, "pval_noadj": 6}[k]])
        errs[k] = np.where(same | bothnan, 0.0, errs[k])
    return errs


def assert_table_close(out, valid, ref, ref_valid, quant=False, rel=REL_TOL, what=""):
    """out/ref: [M, 8] tables; integer fields bit-exact, floats within tolerance."""
    assert np.array_equal(valid, ref_valid), f"{what}: filter mask differs"
    v = ref_valid.astype(bool)
    o, r = out[v], ref[v]
    for c, name in ((0, "AF"), (1, "mac"), (2, "num")):
        assert np.array_equal(o[:, c], r[:, c]), f"{what}: {name} not bit-exact"
    for name, e in table_errors(o, r, quant).items():
        e = e * (REL_TOL / rel)
        assert not np.isnan(e).any(), f"{what}: {name} NaN mismatch"
        if e.size:
            j = int(np.argmax(e))
            assert e[j] <= 1.0, f"{what}: {name} off by {e[j]:.3g} x tolerance at row {j}"
    if not quant:
        assert np.array_equal(o[:, 7], r[:, 7]), f"{what}: converged differs"
