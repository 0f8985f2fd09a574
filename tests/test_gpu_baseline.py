"""GPU parity at the BASELINE.json configurations (the sizes the small tests never reach: 105 SPA
segments, >= 8 sample splits per XCD, limb sums near 2^30) and on constructed inputs for the
root-finder branches random genotypes never take.  Everything goes through the C ABI."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_table_close, scan_model

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600, method="thread")]


@pytest.fixture(scope="module", autouse=True)
def _torch_first():
    import torch
    assert torch.cuda.is_available(), "these tests need an MI355X"
    yield


def _baseline_case(n, trait, prevalence, m, seed=20260, k=3):
    """Model and device-resident synthetic genotypes exactly as bench.py builds them."""
    import torch
    from saigegds_amd import synth
    from saigegds_amd._lib import Scanner
    from saigegds_amd.nullmod import init_nullmod
    mod = synth.synth_null_model(n, trait, prevalence, n_cov=k, seed=seed)
    sm = init_nullmod(mod, np.arange(n), float("nan"), 10.0, 0.1, 0.05, float(mod.var_ratio[0]))
    sc = Scanner(sm, device=0)
    bpv = sc.row_stride()
    dev = torch.device("cuda", 0)
    packed = torch.zeros((m, bpv), dtype=torch.uint8, device=dev)
    thr = synth.variant_thresholds(0, m, seed)
    thr_d = torch.from_numpy(thr.view(np.int32)).to(dev)
    torch.cuda.synchronize()
    sc.synth_2bit_dev(packed.data_ptr(), bpv, m, 0, seed, thr_d.data_ptr())
    sc.sync()
    return sm, sc, packed, bpv


def _scan_two_lanes(sc, packed, bpv, nblock=2):
    import torch
    m = packed.shape[0]
    out = torch.full((m, 8), -1.0, dtype=torch.float64, device=packed.device)
    valid = torch.zeros((m,), dtype=torch.uint8, device=packed.device)
    sc.set_option("lanes", 2)
    per = (m + nblock - 1) // nblock
    for b in range(nblock):
        lo, hi = b * per, min(m, (b + 1) * per)
        sc.scan_2bit_dev(packed[lo:hi].data_ptr(), bpv, hi - lo, out[lo:hi].data_ptr(), valid[lo:hi].data_ptr())
    sc.sync()
    tot, _ = sc.stats_total(reset=True)
    sc.set_option("lanes", 1)
    return out.cpu().numpy(), valid.cpu().numpy(), tot


@pytest.mark.parametrize("name,n,trait,prev", [
    ("C3 N=430K binary 1:99", 430_000, "binary", 0.01),
    ("C4 N=430K quantitative", 430_000, "quantitative", 0.0),
    ("C2 N=50K binary 1:9", 50_000, "binary", 0.10),
])
def test_baseline_configs(name, n, trait, prev):
    """3 000 synthetic variants of each BASELINE configuration, two lanes, against the oracle with
    the plain rule (assert_table_close)."""
    from oracle import Oracle
    m = 3000
    sm, sc, packed, bpv = _baseline_case(n, trait, prev, m)
    try:
        out, valid, tot = _scan_two_lanes(sc, packed, bpv)
    finally:
        sc.close()
    ref, ref_valid = Oracle(sm).scan_2bit(packed.cpu().numpy())
    assert_table_close(out, valid, ref, ref_valid, quant=sm.quant, what=name)
    if trait == "binary":
        # both SPA paths are exercised: the series (most variants) and the exact kernel (rare ones)
        assert tot["n_spa"] > 100 and 0 < tot["n_spa_slow"] < tot["n_spa"], tot


def test_baseline_c3_exact_path_agrees():
    """The series and the exact exp/log kernels are the same function: forcing every flagged variant
    of a C3 block through the exact path must reproduce the table to 1e-11."""
    sm, sc, packed, bpv = _baseline_case(430_000, "binary", 0.01, 1200)
    try:
        a, va, _ = _scan_two_lanes(sc, packed, bpv, 1)
        sc.set_option("spa_exact", 1)
        b, vb, tot = _scan_two_lanes(sc, packed, bpv, 1)
    finally:
        sc.close()
    assert np.array_equal(va, vb) and tot["n_spa"] >= tot["n_spa_slow"] > 30      # (the cutoff exits need no root search)
    v = va.astype(bool)
    np.testing.assert_allclose(a[v][:, 3:7], b[v][:, 3:7], rtol=1e-11)


def test_baseline_c3_both_kernel_forms_at_full_n():
    """C3 at N = 430 000: the automatic choice for K = 3 is the three-plane form; forcing the two-plane form (list pass
    with fourteen sample ranges, per-range sums added up by the epilogue) gives the same table -- the score columns
    are the same exact integers either way."""
    sm, sc, packed, bpv = _baseline_case(430_000, "binary", 0.01, 2000)
    try:
        a, va, ta = _scan_two_lanes(sc, packed, bpv, 1)
        sc.set_option("three_plane", 0)
        b, vb, tb = _scan_two_lanes(sc, packed, bpv, 1)
    finally:
        sc.close()
    assert ta["three_plane"] == 1 and tb["three_plane"] == 0, (ta, tb)
    assert tb["n_unlisted"] == 0 and ta["n_spa"] == tb["n_spa"]
    assert np.array_equal(va, vb)
    v = va.astype(bool)
    assert np.array_equal(a[v][:, :3], b[v][:, :3])                         # AF, mac, num
    np.testing.assert_allclose(a[v][:, 3:7], b[v][:, 3:7], rtol=1e-11)      # (the SPA lists are filled in another order)


def test_baseline_c3_thirteen_covariates():
    """K = 13 at N = 430 000: eleven B fragments in one pass of the score kernel, 1024-sample segments and the
    8-wave form of the cumulant pass (420 segments), against the oracle."""
    from oracle import Oracle
    sm, sc, packed, bpv = _baseline_case(430_000, "binary", 0.01, 800, k=13)
    try:
        out, valid, tot = _scan_two_lanes(sc, packed, bpv)
        limbs, ngroups = sc.score_layout()
    finally:
        sc.close()
    assert ngroups == 1 and int(limbs.sum()) + 1 > 8 * 16, (limbs, ngroups)      # one pass, more than 8 value fragments
    ref, ref_valid = Oracle(sm).scan_2bit(packed.cpu().numpy())
    assert_table_close(out, valid, ref, ref_valid, what="C3 K=13")
    assert tot["n_spa"] > 20, tot


def test_baseline_c3_hard_call_bytes():
    """RAW hard calls (0/1/2/0xFF) at N = 430 000: packed on the device and scanned by the MFMA path, so
    the table is the 2-bit scan's, bit for bit."""
    from saigegds_amd.gds import unpack_dosage_2bit
    n, m = 430_000, 1000
    sm, sc, packed, bpv = _baseline_case(n, "binary", 0.01, m)
    try:
        a, va, _ = _scan_two_lanes(sc, packed, bpv, 1)
        ds = unpack_dosage_2bit(packed.cpu().numpy()[:, :(n + 3) // 4], n)
        ds[ds == 3] = 0xFF                       # code 3 of the packed rows = missing = RAW 0xFF
        b, vb = sc.scan_u8(ds)
    finally:
        sc.close()
    assert np.array_equal(va, vb)
    v = va.astype(bool)
    assert np.array_equal(a[v], b[v])


def test_baseline_c3_real_dosages():
    """Real-valued dosage rows (the REAL branch of get_ds, saige_main.cpp:180-183) at N = 430 000: the tiled
    one-pass score kernels and the dense form of the cumulant pass at the size of the benchmark (105
    segments, items off the queue), against the oracle with the 1e-10 rule."""
    from oracle import Oracle
    from saigegds_amd.gds import unpack_dosage_2bit
    n, m = 430_000, 320
    sm, sc, packed, bpv = _baseline_case(n, "binary", 0.01, m)
    try:
        ds = unpack_dosage_2bit(packed.cpu().numpy()[:, :(n + 3) // 4], n)
        dsf = ds.astype(np.float64)
        dsf[ds == 3] = np.nan                  # code 3 of the packed rows = missing
        dsf[::2] *= 0.93                       # real-valued rows: AF/mac no longer integers
        out, valid = sc.scan_f64(dsf)
        tot, _ = sc.stats_total(reset=True)
    finally:
        sc.close()
    ref, ref_valid = Oracle(sm).scan_f64(dsf)
    assert np.array_equal(valid, ref_valid)
    v = ref_valid.astype(bool)
    # AF, mac, num differ by summation order only.  The reference adds the 430 000 doubles of a row one after
    # the other (vectorization.cpp:186-205): ~1e-10 of rounding at this N, which the oracle reproduces; the
    # device sums pairwise and sits at 1e-13 of the long-double sums.
    np.testing.assert_allclose(out[v][:, :3], ref[v][:, :3], rtol=1e-9)
    ac = np.nansum(dsf.astype(np.longdouble), axis=1)
    num = np.sum(~np.isnan(dsf), axis=1)
    np.testing.assert_allclose(out[v][:, 0], (ac / (2 * num)).astype(np.float64)[v], rtol=1e-13)
    o2 = out.copy()
    o2[:, :3] = ref[:, :3]
    assert_table_close(o2, valid, ref, ref_valid, what="C3 real dosages")
    assert tot["n_spa"] > 5, tot


def test_grm_crossprod_at_430k():
    """Config 5's operator at full sample count: sgx_grm_crossprod over 320 markers vs grm_oracle.c."""
    from oracle import GrmOracle
    from saigegds_amd import synth
    from saigegds_amd._lib import GrmOperator
    n, m = 430_000, 320
    thr = synth.variant_thresholds(0, m, 11, log10_maf=(-2.0, -0.3), flip_frac=0.2, miss_rate=5e-3)
    packed = synth.synth_packed(n, 0, m, 11, thr)
    orc = GrmOracle(packed, n)
    rng = np.random.default_rng(5)
    with GrmOperator(packed, n) as op:
        np.testing.assert_allclose(op.diag(), orc.diag(), rtol=1e-12, atol=0)
        for b in (rng.standard_normal(n), 2.0 * rng.integers(0, 2, n) - 1):
            ref = orc.crossprod(b)
            assert np.max(np.abs(op.crossprod(b) - ref)) <= 1e-11 * np.max(np.abs(ref))


def test_grm_pcg_at_430k():
    """Config 5's solver at full sample count: sgx_grm_pcg (PCG_diag_sigma, src/saige_fitnull.cpp:581-614)
    over 320 markers at N = 430 000 against grm_oracle.c -- binary-trait weights w = mu (1 - mu),
    tau = (1, 0.33) as in saige_model.rds, the reference's tolPCG = 1e-5 / maxiterPCG = 500: same iteration
    count, solution to 1e-8.  The vector kernels and block-partial dot products run at the benchmark size."""
    from oracle import GrmOracle
    from saigegds_amd import synth
    from saigegds_amd._lib import GrmOperator
    n, m = 430_000, 320
    thr = synth.variant_thresholds(0, m, 11, log10_maf=(-2.0, -0.3), flip_frac=0.2, miss_rate=5e-3)
    packed = synth.synth_packed(n, 0, m, 11, thr)
    rng = np.random.default_rng(17)
    mu = rng.uniform(0.002, 0.05, n)            # prevalence of config [2]
    w = mu * (1 - mu)
    b = rng.standard_normal(n)
    orc = GrmOracle(packed, n)
    with GrmOperator(packed, n) as op:
        for tau in ([1.0, 0.33220629], [1.0, 0.0]):
            xr, itr = orc.pcg(w, tau, b, 500, 1e-5)
            xg, itg = op.pcg(w, tau, b, 500, 1e-5)
            assert itg == itr and itr >= 1, (tau, itg, itr)
            np.testing.assert_allclose(xg, xr, rtol=1e-8, atol=1e-10 * np.max(np.abs(xr)))


def test_baseline_c3_one_call_of_50000_variants():
    """One sgx_scan_block call over a whole block of 50 000 variants at N = 430 000 (the bench step): every
    round of the score kernel's item list, all sixteen sample ranges of the missing-genotype lists, both
    tiers of the SPA stage and the per-variant kernels in one launch sequence; 2 000 rows spread over the
    block against the oracle."""
    import torch
    from oracle import Oracle
    from saigegds_amd import synth
    from saigegds_amd._lib import Block, Scanner
    from saigegds_amd.nullmod import init_nullmod
    n, m, seed = 430_000, 50_000, 20260
    mod = synth.synth_null_model(n, "binary", 0.01, n_cov=3, seed=seed)
    sm = init_nullmod(mod, np.arange(n), float("nan"), 10.0, 0.1, 0.05, float(mod.var_ratio[0]))
    dev = torch.device("cuda", 0)
    with Scanner(sm, device=0) as sc:
        bpv = sc.row_stride()
        rows = torch.empty((m, bpv), dtype=torch.uint8, device=dev)
        thr = synth.variant_thresholds(0, m, seed)
        thr_d = torch.from_numpy(thr.view(np.int32)).to(dev)
        torch.cuda.synchronize()
        sc.synth_2bit_dev(rows.data_ptr(), bpv, m, 0, seed, thr_d.data_ptr())
        out = torch.empty((m, 8), dtype=torch.float64, device=dev)
        valid = torch.empty(m, dtype=torch.uint8, device=dev)
        with Block(n, m, device=0) as blk:
            sc.load_block_dev(blk, rows.data_ptr(), bpv, m)
            sc.scan_block(blk, out.data_ptr(), valid.data_ptr())
            sc.sync()
            st = sc.stats()
            # the row-major entry point on the same rows gives the same table, bit for bit
            out2 = torch.empty_like(out)
            valid2 = torch.empty_like(valid)
            sc.scan_2bit_dev(rows.data_ptr(), bpv, m, out2.data_ptr(), valid2.data_ptr())
            sc.sync()
        assert st["n_variants"] == m and st["n_spa"] > 1500
        pick = np.unique(np.concatenate([np.arange(0, m, 25), np.arange(m - 40, m)]))
        sample = rows[torch.from_numpy(pick).to(dev)].cpu().numpy()
        got, got_valid = out.cpu().numpy(), valid.cpu().numpy()
        assert np.array_equal(got_valid, valid2.cpu().numpy())
        assert np.array_equal(np.nan_to_num(got, nan=-7.0), np.nan_to_num(out2.cpu().numpy(), nan=-7.0))
    ref, ref_valid = Oracle(sm).scan_2bit(sample)
    assert ref_valid.sum() > 1800
    assert_table_close(got[pick], got_valid[pick], ref, ref_valid, what="C3, one call of 50 000 variants")


def test_assoc_100snp_dosage_scan():
    """Config 1's second file: the dosages of assoc_100snp.gds (annotation/format/DS, dPackedReal8U;
    the file holds 0 / 1 / 2 only) through sgx_scan_f64 -- the REALSXP branch of get_ds
    (saige_main.cpp:173-174) that seqApply takes for a format node -- and through sgx_scan_u8, with
    the README's mac = 10 (README.md:111)."""
    from oracle import Oracle
    from saigegds_amd._lib import Scanner
    z = np.load(os.path.join(GOLDEN, "assoc_100snp.npz"))
    raw = z["dosage_u8"]
    ds = raw.astype(np.float64)
    sm = scan_model("saige_model.npz", mac=10, sample_ids=[str(s) for s in z["sample_id"]])
    ref, ref_valid = Oracle(sm).scan_f64(ds)
    with Scanner(sm, device=0) as sc:
        out, valid = sc.scan_f64(ds)
        out8, valid8 = sc.scan_u8(raw)
    assert_table_close(out, valid, ref, ref_valid, what="assoc_100snp f64")
    assert_table_close(out8, valid8, ref, ref_valid, what="assoc_100snp u8")
    # The reference's own printed output for this file (README.md:111-129): 38 variants survive mac = 10, the first
    # three are ids 4, 12, 14 with AF.alt 0.0100 / 0.0150 / 0.0375, mac 20 / 30 / 75, num 1000.  (beta / SE / pval
    # of the README come from a model fitted in that session, which is not stored; AF, mac and num do not depend on it.)
    vid = z["variant_id"] if "variant_id" in z.files else np.arange(1, raw.shape[0] + 1)
    for name, o, v in (("f64", out, valid), ("u8", out8, valid8)):
        keep = np.flatnonzero(v)
        assert keep.size == 38, f"{name}: {keep.size} survivors, the README prints 38"
        assert [int(x) for x in vid[keep[:3]]] == [4, 12, 14], f"{name}: first survivors {vid[keep[:3]]}"
        np.testing.assert_allclose(o[keep[:3], 0], [0.0100, 0.0150, 0.0375], rtol=0, atol=1e-15)
        assert np.array_equal(o[keep[:3], 1], [20.0, 30.0, 75.0]) and np.array_equal(o[keep[:3], 2], [1000.0] * 3)


def test_root_finder_edge_branches():
    """root = Inf, the bisection safeguard, non-convergence (1000 iterations / non-finite step),
    cutoff doubling and the p == 0 -> p_noadj fallback: the oracle's trace proves the inputs reach
    every branch, then GPU = oracle."""
    from edge_cases import KINDS, build
    from oracle import Oracle
    from saigegds_amd._lib import Scanner
    cases, found = build()
    assert all(found[k] >= 3 for k in KINDS), found
    p0_rows = n_degen = 0
    for sm, packed, census, packed_degen in cases:
        orc = Oracle(sm)
        ref, ref_valid = orc.scan_2bit(packed)
        tr = orc.trace.as_dict()
        for k in KINDS:
            assert tr[k] >= census[k]
        p0_rows += int(np.sum((ref[:, 7] == 0) & (ref[:, 5] == ref[:, 6]) & (ref_valid == 1)))
        with Scanner(sm, device=0) as sc:
            out, valid = sc.scan_2bit(packed)
            assert_table_close(out, valid, ref, ref_valid, what=f"edge branches N={sm.n} {census}")
            sc.set_option("spa_exact", 1)           # and the same rows through the exact kernels only
            out2, valid2 = sc.scan_2bit(packed)
            assert_table_close(out2, valid2, ref, ref_valid, what=f"edge branches (exact) N={sm.n}")
            if packed_degen.shape[0]:
                # genotype vectors inside the covariates' span: score and variance are rounding noise in the
                # reference itself; the filter, the counts and (where the oracle's row is finite) the
                # convergence flag are still defined and must agree
                sc.set_option("spa_exact", 0)
                refd, refd_valid = orc.scan_2bit(packed_degen)
                outd, validd = sc.scan_2bit(packed_degen)
                assert np.array_equal(validd, refd_valid), f"degenerate rows N={sm.n}: filter mask differs"
                v = refd_valid.astype(bool)
                assert np.array_equal(outd[v][:, :3], refd[v][:, :3]), f"degenerate rows N={sm.n}: AF/mac/num"
                n_degen += int(v.sum())
    assert p0_rows >= 3
    assert n_degen >= 1, "no degenerate candidate was carried"
