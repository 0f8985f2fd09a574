#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz from the reference's own test fixtures.

Run in the build container only (it reads /root/reference, which does not
exist on the GPU box):   python tests/golden/make_fixtures.py

Inputs (data files of the reference's test-suite, inst/unitTests/test_SAIGE.R:79-106):
  inst/extdata/grm1k_10k_snp.gds        genotypes, N=1000 x M=10000
  inst/extdata/assoc_100snp.gds         imputed dosages, N=1000 x M=100
  inst/unitTests/saige_model.rds        binary null model       (input of test.saige_pval)
  inst/unitTests/saige_model_quant.rds  quantitative null model (input of test.saige_pval)
  inst/unitTests/saige_pval.rds         golden result table, binary
  inst/unitTests/saige_pval_quant.rds   golden result table, quantitative
  inst/extdata/pheno.txt.gz             phenotypes + covariates (input of test.saige_fit_null_model)
Outputs are data only (decoded arrays); no reference source is copied.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from saigegds_amd.gds import GdsFile  # noqa: E402
from saigegds_amd.nullmod import load_modobj  # noqa: E402
from saigegds_amd.rds import read_rds  # noqa: E402

REF = os.environ.get("SAIGE_REFERENCE", "/root/reference") + "/inst/"


def save_model(src, dst):
    m = load_modobj(REF + src)
    r = read_rds(REF + src)          # every field checkEquals() compares (test_SAIGE.R:67-75)
    vr, nk = r["var.ratio"], r["obj.noK"]
    np.savez_compressed(
        os.path.join(HERE, dst), trait_type=np.array(m.trait_type), tau=m.tau,
        fitted_values=m.fitted_values, sample_id=np.array(m.sample_id), var_ratio=m.var_ratio,
        y=m.y, V=m.V, X1=m.X1, XV=m.XV, XXVX_inv=m.XXVX_inv,
        coefficients=np.asarray(r["coefficients"], dtype=np.float64),
        linear_predictors=np.asarray(r["linear.predictors"], dtype=np.float64),
        residuals=np.asarray(r["residuals"], dtype=np.float64),
        cov=np.asarray(r["cov"], dtype=np.float64), converged=np.array(bool(np.asarray(r["converged"]).ravel()[0])),
        variant_id=np.asarray(r["variant.id"]),
        noK_mu=np.asarray(nk["mu"], dtype=np.float64), noK_res=np.asarray(nk["res"], dtype=np.float64),
        **{"vr_" + k: np.asarray(vr[k], dtype=np.float64) for k in ("id", "maf", "mac", "var1", "var2", "ratio")})


def save_pheno():
    import gzip
    with gzip.open(REF + "extdata/pheno.txt.gz", "rt") as f:
        hdr = f.readline().split()
        rows = [ln.split() for ln in f if ln.strip()]
    out = {}
    for i, h in enumerate(hdr):
        col = [r[i] for r in rows]
        out[h.replace(".", "_")] = np.array(col) if h == "sample.id" else np.array(
            [np.nan if c == "NA" else float(c) for c in col])
    np.savez_compressed(os.path.join(HERE, "pheno.npz"), **out)


def save_table(src, dst, cols):
    t = read_rds(REF + src)
    out = {}
    for c in cols:
        v = t[c]
        out[c.replace(".", "_")] = np.asarray(v) if not isinstance(v, list) else np.array(v)
    np.savez_compressed(os.path.join(HERE, dst), **out)


def main():
    g = GdsFile(REF + "extdata/grm1k_10k_snp.gds")
    packed, n, m = g.dosage_alt_packed()
    ref, alt = g.alleles()
    np.savez_compressed(
        os.path.join(HERE, "grm1k_10k_snp.npz"), packed=packed, n_samp=n, n_var=m,
        sample_id=np.array(g.sample_id()), variant_id=np.asarray(g.read("variant.id")),
        chromosome=np.array(g.read("chromosome")), position=np.asarray(g.read("position")),
        rs_id=np.array(g.read("annotation/id")), ref=np.array(ref), alt=np.array(alt))

    g2 = GdsFile(REF + "extdata/assoc_100snp.gds")
    ds = g2.dosage_real()
    assert np.array_equal(ds, np.round(ds)) and not np.isnan(ds).any()
    np.savez_compressed(
        os.path.join(HERE, "assoc_100snp.npz"), dosage_u8=ds.astype(np.uint8),
        sample_id=np.array(g2.sample_id()), variant_id=np.asarray(g2.read("variant.id")))

    save_model("unitTests/saige_model.rds", "saige_model.npz")
    save_model("unitTests/saige_model_quant.rds", "saige_model_quant.npz")
    save_pheno()
    save_table("unitTests/saige_pval.rds", "saige_pval.npz",
               ["id", "chr", "pos", "rs.id", "ref", "alt", "AF.alt", "mac", "num", "beta", "SE",
                "pval", "p.norm", "converged"])
    save_table("unitTests/saige_pval_quant.rds", "saige_pval_quant.npz",
               ["id", "chr", "pos", "rs.id", "ref", "alt", "AF.alt", "mac", "num", "beta", "SE",
                "pval"])
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
