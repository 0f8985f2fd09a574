"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle
and the reference's golden tables (inst/unitTests/test_SAIGE.R:79-106)."""
import numpy as np
import pytest

from conftest import assert_table_close, scan_model

# every test here needs the GPU; a hung kernel must fail the test, not stall the run
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300, method="thread")]


@pytest.fixture(scope="module", autouse=True)
def _torch_first():
    """Import torch before the first HIP call of this process: libsaigehip.so then
    binds to the HIP runtime of the torch wheel and nothing is loaded mid-run."""
    import torch
    assert torch.cuda.is_available(), "these tests need an MI355X"
    yield


def _scanner(sm):
    from saigegds_amd._lib import Scanner
    return Scanner(sm, device=0)


def _oracle(sm):
    from oracle import Oracle
    return Oracle(sm)


def _golden_table(g, quant):
    cols = ["AF_alt", "mac", "num", "beta", "SE", "pval"] + ([] if quant else ["p_norm", "converged"])
    t = np.full((g["AF_alt"].size, 8), np.nan)
    for c, n in enumerate(cols):
        t[:, c] = g[n].astype(np.float64)
    return t


def test_golden_binary(grm1k, model_bin, golden_bin):
    """test.saige_pval, binary: all 10 000 variants, mac=4."""
    with _scanner(model_bin) as sc:
        out, valid = sc.scan_2bit(grm1k["packed"])
        st = sc.stats()
    assert st["n_spa"] == 391 and st["n_valid"] == 10000      # 436 under spa.pval, 45 of them leave at the cutoff (SURVEY census)
    ref = _golden_table(golden_bin, False)
    assert_table_close(out, valid, ref, np.ones(10000, np.uint8), what="golden binary")
    oref, ovalid = _oracle(model_bin).scan_2bit(grm1k["packed"])
    assert_table_close(out, valid, oref, ovalid, what="oracle binary")


def test_golden_quant(grm1k, model_quant, golden_quant):
    """test.saige_pval, quantitative."""
    with _scanner(model_quant) as sc:
        out, valid = sc.scan_2bit(grm1k["packed"])
    ref = _golden_table(golden_quant, True)
    assert_table_close(out, valid, ref, np.ones(10000, np.uint8), quant=True, what="golden quant")
    oref, ovalid = _oracle(model_quant).scan_2bit(grm1k["packed"])
    assert_table_close(out, valid, oref, ovalid, quant=True, what="oracle quant")


def _synthetic_case(n, m, trait, prevalence, seed, k=3, miss=1e-2, flip=0.3, lo=-2.5):
    from saigegds_amd import synth
    from saigegds_amd.nullmod import init_nullmod
    mod = synth.synth_null_model(n, trait, prevalence, n_cov=k, seed=seed)
    sm = init_nullmod(mod, np.arange(n), float("nan"), 10, 0.1, 0.05, float(mod.var_ratio[0]))
    thr = synth.variant_thresholds(0, m, seed, log10_maf=(lo, -0.3), flip_frac=flip, miss_rate=miss)
    packed = synth.synth_packed(n, 0, m, seed, thr)
    return sm, packed


@pytest.mark.parametrize("trait,prev", [("binary", 0.1), ("binary", 0.02), ("quantitative", 0.0)])
def test_synthetic_flip_missing(trait, prev):
    """Branches the goldens never reach: AF>0.5 flip, missing genotypes (imputed
    2*AF), filter rejections; ragged N (not a multiple of 64)."""
    sm, packed = _synthetic_case(3001, 1500, trait, prev, seed=7)
    orc = _oracle(sm)
    ref, ref_valid = orc.scan_2bit(packed)
    tr = orc.trace.as_dict()
    assert (ref_valid == 0).any() and (ref_valid == 1).any()
    if trait == "binary":
        assert tr["flipped"] > 50 and tr["spa_done"] > 10
    with _scanner(sm) as sc:
        out, valid = sc.scan_2bit(packed)
    assert_table_close(out, valid, ref, ref_valid, quant=sm.quant, what=f"synthetic {trait}")


@pytest.mark.parametrize("k", [1, 2, 4, 5, 8, 9, 13, 16])
def test_covariate_counts(k):
    sm, packed = _synthetic_case(2000, 400, "binary", 0.1, seed=11 + k, k=k)
    ref, ref_valid = _oracle(sm).scan_2bit(packed)
    with _scanner(sm) as sc:
        out, valid = sc.scan_2bit(packed)
    assert_table_close(out, valid, ref, ref_valid, what=f"K={k}")


@pytest.mark.parametrize("trait", ["binary", "quantitative"])
def test_heavy_tailed_covariate_widens_limbs(trait):
    """A covariate with one sample 10^4 x the column scale (K = 5): sgx_init measures the dynamic
    range of every fixed-point column and widens the affected ones; results stay inside 1e-10."""
    from saigegds_amd import synth
    from saigegds_amd.nullmod import init_nullmod
    n, m = 5000, 1200
    thr = synth.variant_thresholds(0, m, 5, log10_maf=(-2.5, -0.3), flip_frac=0.3, miss_rate=1e-2)
    packed = synth.synth_packed(n, 0, m, 5, thr)
    layouts = {}
    for outlier in (0.0, 1e4):
        mod = synth.synth_null_model(n, trait, 0.1, n_cov=5, seed=77, outlier=outlier)
        sm = init_nullmod(mod, np.arange(n), float("nan"), 10, 0.1, 0.05, float(mod.var_ratio[0]))
        ref, ref_valid = _oracle(sm).scan_2bit(packed)
        with _scanner(sm) as sc:
            layouts[outlier] = sc.score_layout()
            out, valid = sc.scan_2bit(packed)
        assert_table_close(out, valid, ref, ref_valid, quant=sm.quant, what=f"{trait} outlier={outlier}")
    plain, wide = layouts[0.0], layouts[1e4]
    assert plain[1] >= 1 and wide[1] >= 1                      # both on the exact-integer path
    assert (wide[0] >= plain[0]).all(), (plain, wide)
    if trait == "binary":                                       # the 5-limb c' column of the outlier covariate
        assert wide[0].sum() > plain[0].sum(), (plain, wide)


def test_dosage_inputs(grm1k, model_bin):
    """RAW and REAL branches of get_ds (saige_main.cpp:171-183)."""
    from saigegds_amd.gds import unpack_dosage_2bit
    ds = unpack_dosage_2bit(grm1k["packed"][:2000], 1000)
    ds[5, 17] = 0xFF
    ds[9, :40] = 0xFF
    orc = _oracle(model_bin)
    ref, ref_valid = orc.scan_u8(ds)
    with _scanner(model_bin) as sc:
        out, valid = sc.scan_u8(ds)
        assert_table_close(out, valid, ref, ref_valid, what="u8 dosage")
        dsf = ds.astype(np.float64)
        dsf[ds == 0xFF] = np.nan
        dsf[100:200] *= 0.93          # real-valued dosages: AF/mac no longer integers
        ref2, ref_valid2 = orc.scan_f64(dsf)
        out2, valid2 = sc.scan_f64(dsf)
    assert np.array_equal(valid2, ref_valid2)
    v = ref_valid2.astype(bool)
    # integer genotype rows stay bit-exact; real-valued ones differ by summation order
    np.testing.assert_allclose(out2[v][:, :3], ref2[v][:, :3], rtol=1e-13)
    o2, r2 = out2.copy(), ref2.copy()
    o2[:, :3] = r2[:, :3]
    assert_table_close(o2, valid2, r2, ref_valid2, what="f64 dosage")       # the 1e-10 rule


def test_hard_call_dosages_take_the_packed_path(grm1k, model_bin):
    """RAW / INTEGER rows that hold 0, 1, 2, missing only are packed to 2-bit on the device and run
    the MFMA score path (six score-stage launches per chunk); a block with any other value does not.
    The host pipeline is driven with a tiny chunk so that several chunks are in flight."""
    from saigegds_amd.gds import unpack_dosage_2bit
    codes = unpack_dosage_2bit(grm1k["packed"][:3000], 1000)
    u8 = codes.copy()
    u8[codes == 3] = 0xFF
    i32 = codes.astype(np.int32)
    i32[codes == 3] = -2147483648
    orc = _oracle(model_bin)
    ref, ref_valid = orc.scan_2bit(grm1k["packed"][:3000])
    with _scanner(model_bin) as sc:
        sc.set_option("pipe_mb", 1)                    # 1 MiB chunks: 1048 u8 rows, 262 i32 rows
        out, valid = sc.scan_u8(u8)
        st = sc.stats()
        assert st["n_variants"] == 3000 and st["score_launches"] == 5 * 3, st      # 3 chunks on the MFMA path (5 launches each)
        assert_table_close(out, valid, ref, ref_valid, what="u8 hard calls")
        out, valid = sc.scan_i32(i32)
        st = sc.stats()
        nchunk = -(-3000 // ((1 << 20) // (12 * 1000)))           # i32 rows keep room for their doubles: 12 N bytes
        assert st["n_variants"] == 3000 and st["score_launches"] == 5 * nchunk, st
        assert_table_close(out, valid, ref, ref_valid, what="i32 hard calls")
        out, valid = sc.scan_2bit(grm1k["packed"][:3000])
        assert_table_close(out, valid, ref, ref_valid, what="2-bit, chunked")
        # one value outside 0/1/2/NA: that chunk goes through the dosage kernels, same numbers
        i32b = i32.copy()
        i32b[7, 11] = 5
        u8b = u8.copy()
        u8b[7, 11] = 5
        refb, refb_valid = orc.scan_u8(u8b)
        out, valid = sc.scan_i32(i32b)
        assert sc.stats()["score_launches"] == 5 * nchunk - 4     # one chunk on the (one-launch) dosage score kernels
        assert_table_close(out, valid, refb, refb_valid, what="i32 with a non-call value")
        out, valid = sc.scan_u8(u8b)
        assert_table_close(out, valid, refb, refb_valid, what="u8 with a non-call value")


def test_empty_and_errors(model_bin):
    from saigegds_amd._lib import SgxError
    with _scanner(model_bin) as sc:
        out, valid = sc.scan_2bit(np.zeros((0, 250), np.uint8))
        assert out.shape == (0, 8) and valid.shape == (0,)
        with pytest.raises(SgxError):          # ERR_DS_LEN, saige_main.cpp:157,417-418
            sc.scan_2bit(np.zeros((3, 100), np.uint8))
        # monomorphic and all-missing variants are filtered, not crashed on
        pk = np.zeros((2, 250), np.uint8)
        pk[1] = 0xFF
        out, valid = sc.scan_2bit(pk)
        assert not valid.any() and np.isnan(out).all()


def test_device_resident_and_generator():
    """sgx_scan_2bit_dev on HBM-resident rows produced by sgx_synth_2bit_dev;
    the numpy twin of the generator must give identical bytes."""
    import torch
    from saigegds_amd import synth
    from saigegds_amd.nullmod import init_nullmod
    n, m, seed = 5000, 3000, 20260
    mod = synth.synth_null_model(n, "binary", 0.05, seed=seed)
    sm = init_nullmod(mod, np.arange(n), float("nan"), 10, 0.1, 0.05, float(mod.var_ratio[0]))
    thr = synth.variant_thresholds(100, m, seed)
    with _scanner(sm) as sc:
        bpv = sc.row_stride()
        dev = torch.device("cuda:0")
        packed = torch.zeros((m, bpv), dtype=torch.uint8, device=dev)
        thr_d = torch.from_numpy(thr.view(np.int32)).to(dev)
        out = torch.empty((m, 8), dtype=torch.float64, device=dev)
        valid = torch.empty((m,), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        sc.synth_2bit_dev(packed.data_ptr(), bpv, m, 100, seed, thr_d.data_ptr())
        sc.scan_2bit_dev(packed.data_ptr(), bpv, m, out.data_ptr(), valid.data_ptr())
        sc.sync()
        pk_host = packed.cpu().numpy()
        out_h, valid_h = out.cpu().numpy(), valid.cpu().numpy()
    assert np.array_equal(pk_host, synth.synth_packed(n, 100, m, seed, thr, bpv))
    ref, ref_valid = _oracle(sm).scan_2bit(pk_host)
    assert_table_close(out_h, valid_h, ref, ref_valid, what="device-resident")


def test_seqAssocGLMM_SPA_driver(grm1k, golden_bin, tmp_path):
    """test.saige_pval through the host driver: data.frame columns, filter
    intersection (mac=40 drops variants), sample subset + reorder, .rds output."""
    from conftest import load_null_model
    from saigegds_amd import GenotypeSource, seqAssocGLMM_SPA
    from saigegds_amd.rds import read_rds
    mod = load_null_model("saige_model.npz")
    sid = [str(s) for s in grm1k["sample_id"]]
    src = GenotypeSource(sid, packed=grm1k["packed"], variant_id=grm1k["variant_id"],
                         chromosome=[str(c) for c in grm1k["chromosome"]], position=grm1k["position"],
                         rs_id=[str(c) for c in grm1k["rs_id"]], ref=[str(c) for c in grm1k["ref"]],
                         alt=[str(c) for c in grm1k["alt"]])
    ans = seqAssocGLMM_SPA(src, mod, mac=4, verbose=False)
    assert list(ans) == ["id", "chr", "pos", "rs.id", "ref", "alt", "AF.alt", "mac", "num", "beta",
                         "SE", "pval", "p.norm", "converged"]
    assert np.array_equal(ans["id"], golden_bin["id"]) and ans["rs.id"][:2] == ["rs1", "rs2"]
    assert np.array_equal(ans["AF.alt"], golden_bin["AF_alt"]) and ans["num"].dtype == np.int32
    for c, g in (("beta", "beta"), ("SE", "SE"), ("pval", "pval"), ("p.norm", "p_norm")):
        assert np.max(np.abs(ans[c] / golden_bin[g] - 1)) <= 1e-10, c
    assert np.array_equal(ans["converged"], golden_bin["converged"])
    # stricter MAC filter -> fewer rows, same values on the survivors
    fn = str(tmp_path / "out.rds")
    assert seqAssocGLMM_SPA(src, mod, mac=40, res_savefn=fn, verbose=False) is None
    r = read_rds(fn)
    keep = golden_bin["mac"] >= 40
    assert 0 < keep.sum() < 10000 and np.array_equal(np.asarray(r["id"]), golden_bin["id"][keep])
    assert np.max(np.abs(np.asarray(r["pval"]) / golden_bin["pval"][keep] - 1)) <= 1e-10


def test_file_to_table_equals_in_memory_scan(grm1k, tmp_path, monkeypatch):
    """seqAssocGLMM_SPA on GDS files written by the committed generator (genotype/data raw, LZMA_RA, ZIP_RA,
    LZ4_RA) = the scan of the same rows from memory, bit for bit: the whole ingest path (block-wise decode
    one block ahead, pinned pipeline, block load on the device) moves nothing.  With the model's samples a
    reordered subset of the file's (selection in the 2-bit domain) the counts stay exact and the statistics
    move by rounding only (the sums run in the file's sample order).  An .rda model file gives the same table."""
    from types import SimpleNamespace
    from conftest import load_null_model
    from saigegds_amd import GenotypeSource, seqAssocGLMM_SPA
    from saigegds_amd import assoc as assoc_mod
    from saigegds_amd.gds_write import write_seqarray_genotypes
    from saigegds_amd.results import save_model
    mod = load_null_model("saige_model.npz")
    n = 1000
    m = 2300
    pk = grm1k["packed"][:m]
    # the file holds 1 000 + 7 samples in another order than the model
    rng = np.random.default_rng(3)
    from saigegds_amd.gds import pack_dosage_2bit, unpack_dosage_2bit
    codes = unpack_dosage_2bit(pk, n)
    perm = rng.permutation(n + 7)
    wide = np.concatenate([codes, rng.integers(0, 3, (m, 7)).astype(np.uint8)], axis=1)[:, perm]
    sid_all = ([str(s) for s in grm1k["sample_id"]] + [f"extra{i}" for i in range(7)])
    sid_file = [sid_all[i] for i in perm]
    monkeypatch.setattr(assoc_mod, "BLOCK_SIZE", 600)            # four blocks: decode runs ahead of the scan
    src = GenotypeSource([str(s) for s in grm1k["sample_id"]], packed=pk)
    ref = seqAssocGLMM_SPA(src, mod, mac=4, verbose=False)
    assert len(ref["pval"]) > 1500
    for comp in ("none", "LZMA_RA", "ZIP_RA", "LZ4_RA"):
        fn = str(tmp_path / f"same_{comp}.gds")
        write_seqarray_genotypes(fn, pk, n, sample_id=[str(s) for s in grm1k["sample_id"]], compress=comp, ra_block=100_000)
        tm = {}
        ans = seqAssocGLMM_SPA(fn, mod, mac=4, verbose=False, timing=tm)
        assert tm["decode_s"] > 0 and tm["scan_s"] > 0
        for c in ("AF.alt", "mac", "num", "beta", "SE", "pval", "p.norm", "converged"):
            assert np.array_equal(np.asarray(ans[c]), np.asarray(ref[c])), (comp, c)
        # a reordered superset of the model's samples
        fn = str(tmp_path / f"geno_{comp}.gds")
        write_seqarray_genotypes(fn, pack_dosage_2bit(wide), n + 7, sample_id=sid_file, compress=comp, ra_block=100_000)
        ans = seqAssocGLMM_SPA(fn, mod, mac=4, verbose=False)
        for c in ("AF.alt", "mac", "num", "converged"):
            assert np.array_equal(np.asarray(ans[c]), np.asarray(ref[c])), (comp, c)
        for c in ("beta", "SE", "pval", "p.norm"):
            np.testing.assert_allclose(np.asarray(ans[c]), np.asarray(ref[c]), rtol=1e-10, err_msg=f"{comp} {c}")
    # the model from an .rda file
    nr = len(mod.var_ratio)
    full = SimpleNamespace(
        coefficients=np.zeros(3), coef_names=None, tau=mod.tau, linear_predictors=np.zeros(n), fitted_values=mod.fitted_values,
        residuals=mod.y - mod.fitted_values, cov=np.eye(3), converged=True, y=mod.y, mu_noK=mod.fitted_values,
        res_noK=mod.y - mod.fitted_values, V=mod.V, X1=mod.X1, XV=mod.XV, XXVX_inv=mod.XXVX_inv, trait_type=mod.trait_type,
        sample_id=list(mod.sample_id), variant_id=np.arange(1, 4),
        var_ratio_table={"id": np.arange(1, nr + 1), "maf": np.full(nr, 0.1), "mac": np.full(nr, 30.0), "var1": np.ones(nr),
                         "var2": np.ones(nr), "ratio": np.asarray(mod.var_ratio)})
    mfn = str(tmp_path / "model.rda")
    save_model(full, mfn)
    ans = seqAssocGLMM_SPA(str(tmp_path / "same_ZIP_RA.gds"), mfn, mac=4, verbose=False)
    assert np.array_equal(np.asarray(ans["pval"]), np.asarray(ref["pval"]))


@pytest.mark.parametrize("option,value,counter", [
    ("spa_exact", 1, "n_spa_slow"),       # every flagged variant through the exact exp/log kernel
    ("force_dense", 1, "n_spa_dense"),    # exact g_pos / g_neg pass (SPATest.cpp:328-332)
    ("score_v1", 1, None),                # gather score kernel instead of the MFMA path
])
def test_fallback_paths_give_identical_rows(option, value, counter):
    """Every slow path is the same algorithm: forcing it must not move a result."""
    sm, packed = _synthetic_case(3001, 1200, "binary", 0.05, seed=23)
    ref, ref_valid = _oracle(sm).scan_2bit(packed)
    with _scanner(sm) as sc:
        base, _ = sc.scan_2bit(packed)
        sc.set_option(option, value)
        out, valid = sc.scan_2bit(packed)
        st = sc.stats()
    if counter:
        assert st[counter] > 10, st
    assert_table_close(out, valid, ref, ref_valid, what=f"{option}={value}")
    v = ref_valid.astype(bool)
    np.testing.assert_allclose(out[v][:, 3:7], base[v][:, 3:7], rtol=1e-11)


def _scan_block(sc, blk, m):
    """scan of a loaded block into fresh device buffers -> (out, valid) as numpy"""
    import torch
    dev = torch.device("cuda", 0)
    out = torch.full((m, 8), -1.0, dtype=torch.float64, device=dev)
    valid = torch.zeros(m, dtype=torch.uint8, device=dev)
    sc.scan_block(blk, out.data_ptr(), valid.data_ptr())
    sc.sync()
    return out.cpu().numpy(), valid.cpu().numpy()


def test_block_carrier_lists_equal_row_scans():
    """Resident genotype blocks list the carriers of the rare variants (both orientations: 10 % of the synthetic
    variants have the alt allele as the major one); the per-variant SPA kernels walk those lists.  Same rows when
    the lists are ignored ("spa_abl" 512: every variant scans its row), from the row-major call (which never has
    lists), when the block's list is too small for all of them (clist_avg: the later variants go unlisted), and
    through the exact sweeps."""
    from saigegds_amd._lib import Block
    sm, packed = _synthetic_case(3001, 1200, "binary", 0.05, seed=29)
    ref, ref_valid = _oracle(sm).scan_2bit(packed)
    v = ref_valid.astype(bool)
    with _scanner(sm) as sc, Block(sm.n, 1200) as blk:
        sc.load_block(blk, packed)
        assert blk.n_variants == 1200
        base, valid = _scan_block(sc, blk, 1200)
        assert sc.stats()["n_spa"] > 30
        assert_table_close(base, valid, ref, ref_valid, what="carrier lists")
        sc.set_option("spa_abl", 512)
        rows, _ = _scan_block(sc, blk, 1200)
        np.testing.assert_allclose(rows[v][:, 3:7], base[v][:, 3:7], rtol=1e-11)
        sc.set_option("spa_abl", 0)
        direct, valid = sc.scan_2bit(packed)
        assert_table_close(direct, valid, ref, ref_valid, what="row-major call")
        np.testing.assert_allclose(direct[v][:, 3:7], base[v][:, 3:7], rtol=1e-11)
        assert np.array_equal(direct[v][:, :3], base[v][:, :3])
        sc.set_option("spa_exact", 1)
        exact, valid = _scan_block(sc, blk, 1200)
        assert_table_close(exact, valid, ref, ref_valid, what="carrier lists, exact sweeps")
        sc.set_option("spa_exact", 0)
        with Block(sm.n, 1200, clist_avg=40) as small:      # 48 000 entries for 1 200 variants with ~300 carriers each
            sc.load_block(small, packed)
            part, valid = _scan_block(sc, small, 1200)
    assert_table_close(part, valid, ref, ref_valid, what="carrier lists, full list")
    np.testing.assert_allclose(part[v][:, 3:7], base[v][:, 3:7], rtol=1e-11)


def test_block_reload_without_sync_waits_for_its_readers():
    """Two lanes: a block is scanned and reloaded with other rows right away, no sync in between -- the load has
    to wait for the scan that still reads the block (and for a deferred dense pass that points at it); both
    tables must be their own rows' tables."""
    import torch
    from saigegds_amd._lib import Block
    sm, packed = _synthetic_case(3001, 1600, "binary", 0.05, seed=41)
    ref, ref_valid = _oracle(sm).scan_2bit(packed)
    dev = torch.device("cuda", 0)
    with _scanner(sm) as sc, Block(sm.n, 800) as blk:
        sc.set_option("lanes", 2)
        sc.set_option("force_dense", 1)               # every flagged variant onto the deferred dense pass
        bpv = sc.row_stride()
        pk = torch.zeros((2, 800, bpv), dtype=torch.uint8, device=dev)
        pk[:, :, :packed.shape[1]] = torch.from_numpy(packed.reshape(2, 800, -1)).to(dev)
        out = torch.full((2, 800, 8), -1.0, dtype=torch.float64, device=dev)
        valid = torch.zeros((2, 800), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        for rep in range(3):
            for half in range(2):
                sc.load_block_dev(blk, pk[half].data_ptr(), bpv, 800)
                sc.scan_block(blk, out[half].data_ptr(), valid[half].data_ptr())
        sc.sync()
        got, gv = out.cpu().numpy().reshape(-1, 8), valid.cpu().numpy().reshape(-1)
    assert_table_close(got, gv, ref, ref_valid, what="reload without sync")


@pytest.mark.parametrize("trait,k,miss", [("binary", 3, 1e-3), ("binary", 3, 0.04), ("quantitative", 3, 0.02), ("binary", 6, 0.02),
                                          ("binary", 13, 0.01), ("quantitative", 13, 0.0)])
def test_three_plane_form_equals_two_plane_form(trait, k, miss):
    """The contraction kernel has two forms: two planes + the sparse pass over the listed missing genotypes, and
    three planes (the sums over the missing samples from a third MFMA plane: no lists).  Both are exact integer
    arithmetic on the same fixed-point values, so the score stage's integers are the same and the tables agree to
    the last bit wherever the SPA stage is not involved, to rounding where it is; both against the oracle."""
    sm, packed = _synthetic_case(3001, 900, trait, 0.05, seed=53, k=k, miss=miss)
    ref, ref_valid = _oracle(sm).scan_2bit(packed)
    res = {}
    with _scanner(sm) as sc:
        for form in (0, 1):
            sc.set_option("three_plane", form)
            out, valid = sc.scan_2bit(packed)
            st = sc.stats()
            assert st["three_plane"] == form, st
            assert_table_close(out, valid, ref, ref_valid, quant=sm.quant, what=f"three_plane={form} {trait} K={k} miss={miss}")
            res[form] = (out, valid)
    assert np.array_equal(res[0][1], res[1][1])
    v = ref_valid.astype(bool)
    assert np.array_equal(res[0][0][v][:, :3], res[1][0][v][:, :3])
    cols = slice(3, 6) if sm.quant else slice(3, 7)
    np.testing.assert_allclose(res[0][0][v][:, cols], res[1][0][v][:, cols], rtol=1e-11)


def test_missing_rate_picks_the_form():
    """Automatic choice for row-major calls: models of up to four B fragments always take the three-plane form; a wider
    model starts on the two-plane form (lists of the missing genotypes), a step that finds more than 0.5 % of its
    genotypes missing turns the following calls to the three-plane form, a three-plane step with few missing
    genotypes turns them back; resident blocks decide from the census of their own load.  Tables are the oracle's
    either way."""
    import torch
    from saigegds_amd._lib import Block
    dev = torch.device("cuda", 0)
    seen = set()
    for trait, k in (("binary", 5), ("binary", 3), ("quantitative", 3)):
        sm, dirty = _synthetic_case(3001, 800, trait, 0.05, seed=59, k=k, miss=0.03)
        _, clean = _synthetic_case(3001, 800, trait, 0.05, seed=59, k=k, miss=1e-3)
        orc = _oracle(sm)
        with _scanner(sm) as sc, Block(sm.n, 800) as b_dirty, Block(sm.n, 800) as b_clean:
            limbs, ngroups = sc.score_layout()
            nbf = (int(limbs.sum()) + 1 + 15) // 16 + 1          # B fragments: value columns + constant, and the bit-1 fragment
            narrow = nbf <= 4
            seen.add(narrow)
            bpv = sc.row_stride()

            def dev_rows(p):
                t = torch.zeros((800, bpv), dtype=torch.uint8, device=dev)
                t[:, :p.shape[1]] = torch.from_numpy(p).to(dev)
                return t

            rows = {"dirty": dev_rows(dirty), "clean": dev_rows(clean)}
            out = torch.zeros((800, 8), dtype=torch.float64, device=dev)
            valid = torch.zeros(800, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            for name, wide_want in [("dirty", 0), ("dirty", 1), ("clean", 1), ("clean", 0), ("clean", 0)]:
                want = 1 if narrow else wide_want
                sc.scan_2bit_dev(rows[name].data_ptr(), bpv, 800, out.data_ptr(), valid.data_ptr())
                st = sc.stats()
                assert st["three_plane"] == want, (trait, k, nbf, name, want, st)
                ref, ref_valid = orc.scan_2bit(dirty if name == "dirty" else clean)
                assert_table_close(out.cpu().numpy(), valid.cpu().numpy(), ref, ref_valid, quant=sm.quant, what=f"{trait} K={k} {name} rows, three_plane={want}")
            sc.load_block(b_dirty, dirty)
            sc.load_block(b_clean, clean)
            for blk, p, want in ((b_dirty, dirty, 1), (b_clean, clean, 0)):
                o, v = _scan_block(sc, blk, 800)
                assert sc.stats()["three_plane"] == want
                ref, ref_valid = orc.scan_2bit(p)
                assert_table_close(o, v, ref, ref_valid, quant=sm.quant, what=f"block, three_plane={want}")
    assert seen == {True, False}, seen      # both rules were exercised


def _uncentred_model(n, trait, seed=5):
    """A design as cohorts have them when nothing is centred or orthogonalised (X.transform = FALSE): intercept,
    age ~ N(57, 8), age squared, sex, one principal component -- cond(X'VX) ~ 2e10."""
    from saigegds_amd.nullmod import NullModel, init_nullmod
    rng = np.random.default_rng(seed)
    age = rng.normal(57, 8, n)
    sex = rng.integers(0, 2, n).astype(float)
    pc = rng.standard_normal(n)
    X = np.column_stack([np.ones(n), age, age * age, sex, pc])
    if trait == "binary":
        mu_t = 1 / (1 + np.exp(-(-6 + 0.05 * age + 0.3 * sex + 0.2 * pc)))
        y = (rng.random(n) < mu_t).astype(float)
        Xs = X / np.abs(X).max(0)                 # (the fit in scaled columns; the model keeps the raw ones)
        beta = np.zeros(5)
        for _ in range(100):
            mu = 1 / (1 + np.exp(-(Xs @ beta)))
            W = mu * (1 - mu)
            step = np.linalg.solve(Xs.T @ (Xs * W[:, None]), Xs.T @ (y - mu))
            beta += step
            if np.max(np.abs(step)) < 1e-13:
                break
        mu = 1 / (1 + np.exp(-(Xs @ beta)))
        V, tau, vr = mu * (1 - mu), np.array([1.0, 0.0]), 0.94
    else:
        y = 5 + 0.02 * age + 0.3 * sex + rng.standard_normal(n)
        beta = np.linalg.lstsq(X, y, rcond=None)[0]
        mu = X @ beta
        V, tau, vr = np.ones(n), np.array([float(np.var(y - mu)), 0.0]), 1.03
    XVX = X.T @ (X * V[:, None])
    assert np.linalg.cond(XVX) > 1e9
    mod = NullModel(trait_type=trait, tau=tau, fitted_values=mu, sample_id=[f"s{i}" for i in range(n)],
                    var_ratio=np.array([vr]), y=y, V=V, X1=X, XV=(X * V[:, None]).T, XXVX_inv=X @ np.linalg.inv(XVX),
                    coefficients=beta)
    return init_nullmod(mod, np.arange(n), float("nan"), 10, 0.1, 0.05, vr)


@pytest.mark.parametrize("trait", ["binary", "quantitative"])
@pytest.mark.parametrize("n,m", [(5000, 1500), (430_000, 300)])
def test_uncentred_collinear_design(trait, n, m):
    """The fixed-point score path on a design whose covariate projections cancel across columns (VERDICT r03,
    weak 2): the 1e-10 rule against the oracle, whose own double arithmetic stays within 1e-13 of its long-double
    twin on this design.  sgx_score_layout / stats show what the library did about it."""
    from saigegds_amd import synth
    sm = _uncentred_model(n, trait)
    thr = synth.variant_thresholds(0, m, 11, log10_maf=(-2.5 if n < 10000 else -3.0, -0.3), flip_frac=0.2, miss_rate=1e-3)
    packed = synth.synth_packed(n, 0, m, 11, thr)
    ref, ref_valid = _oracle(sm).scan_2bit(packed)
    with _scanner(sm) as sc:
        limbs, ngroups = sc.score_layout()
        out, valid = sc.scan_2bit(packed)
        st = sc.stats()
    from conftest import table_errors
    v = ref_valid.astype(bool)
    worst = {k: float(np.nanmax(e)) if e.size else 0.0 for k, e in table_errors(out[v], ref[v], sm.quant).items()}
    print(f"uncentred design {trait} N={n}: limbs {[int(x) for x in limbs]}, guarded {st['n_guarded']} of {st['n_valid']}, "
          f"worst errors in units of the tolerance {worst}")
    assert_table_close(out, valid, ref, ref_valid, quant=sm.quant, what=f"uncentred collinear design, {trait}, N = {n}")
    assert st["n_guarded"] < st["n_valid"], "the fixed-point path should hold most of this design's variants"


def test_fixed_point_guard():
    """score3_epilogue bounds, per variant, what the quantisation of the fixed-point columns can have done to the
    z-score and hands the variant to the FP64 kernel beyond 2e-11.  On well-conditioned models (those of the benchmark)
    it never fires; with the guard at 10^-300 every variant takes the FP64 kernel and the table is still the oracle one;
    with the guard off nothing is handed over."""
    sm, packed = _synthetic_case(3001, 900, "binary", 0.05, seed=61, miss=1e-3)
    ref, ref_valid = _oracle(sm).scan_2bit(packed)
    with _scanner(sm) as sc:
        out, valid = sc.scan_2bit(packed)
        st = sc.stats()
        assert st["n_guarded"] == 0, st
        assert_table_close(out, valid, ref, ref_valid, what="guard at its default")
        sc.set_option("guard_exp", 300)
        out, valid = sc.scan_2bit(packed)
        st = sc.stats()
        assert st["n_guarded"] >= st["n_valid"] > 100, st          # (rejected variants pass through the guard first)
        assert_table_close(out, valid, ref, ref_valid, what="every variant through the FP64 kernel")
        sc.set_option("guard_exp", 0)
        out, valid = sc.scan_2bit(packed)
        assert sc.stats()["n_guarded"] == 0
        assert_table_close(out, valid, ref, ref_valid, what="guard off")


def test_rows_longer_than_the_list_kernels_registers():
    """N > 524 288: a sample range of the list kernels (one sixteenth of a row) no longer fits the eight 16-byte
    pieces a lane holds in registers, so they walk it twice / in strides (kern_lists.h, the `long ranges` branches),
    in the row-major call and in a block load; both tables against the oracle, 0.3 % missing."""
    from saigegds_amd._lib import Block
    n, m = 540_001, 40
    sm, packed = _synthetic_case(n, m, "binary", 0.02, seed=67, miss=3e-3, lo=-2.0)
    ref, ref_valid = _oracle(sm).scan_2bit(packed)
    assert ref_valid.sum() >= 30
    with _scanner(sm) as sc, Block(n, m) as blk:
        sc.set_option("three_plane", 0)
        out, valid = sc.scan_2bit(packed)
        assert_table_close(out, valid, ref, ref_valid, what="long rows, row-major call")
        sc.load_block(blk, packed)
        o2, v2 = _scan_block(sc, blk, m)
        assert sc.stats()["three_plane"] == 0 and sc.stats()["n_unlisted"] == 0
        assert_table_close(o2, v2, ref, ref_valid, what="long rows, resident block")


def test_two_lanes_give_identical_tables():
    """"lanes" = 2 .. 4: successive device-resident scans go round-robin over that many streams with their own
    workspace; every block's table must equal the single-lane one, and the totals must add up."""
    import torch
    sm, packed = _synthetic_case(3001, 2400, "binary", 0.05, seed=31)
    dev = torch.device("cuda", 0)
    with _scanner(sm) as sc:
        bpv = sc.row_stride()
        pk = torch.zeros((4, 600, bpv), dtype=torch.uint8, device=dev)
        pk[:, :, :packed.shape[1]] = torch.from_numpy(packed.reshape(4, 600, -1)).to(dev)
        res = {}
        for lanes in (1, 2, 4, 3):
            sc.set_option("lanes", lanes)
            out = torch.full((4, 600, 8), -1.0, dtype=torch.float64, device=dev)
            valid = torch.zeros((4, 600), dtype=torch.uint8, device=dev)
            sc.stats_total(reset=True)
            for rep in range(3):              # lanes are reused: 12 calls
                for b in range(4):
                    sc.scan_2bit_dev(pk[b].data_ptr(), bpv, 600, out[b].data_ptr(), valid[b].data_ptr())
            tot, ncalls = sc.stats_total(reset=True)
            assert ncalls == 12 and tot["n_variants"] == 12 * 600
            res[lanes] = (out.cpu().numpy(), valid.cpu().numpy(), tot)
        sc.set_option("lanes", 1)
    for lanes in (2, 3, 4):
        assert np.array_equal(res[1][1], res[lanes][1])
        assert np.array_equal(np.nan_to_num(res[1][0], nan=-7.0), np.nan_to_num(res[lanes][0], nan=-7.0))
        assert res[1][2]["n_spa"] == res[lanes][2]["n_spa"] and res[1][2]["n_valid"] == res[lanes][2]["n_valid"]
    ref, ref_valid = _oracle(sm).scan_2bit(packed)
    assert_table_close(res[2][0].reshape(-1, 8), res[2][1].reshape(-1), ref, ref_valid, what="lanes=2")


def test_dense_pass_of_device_resident_calls_runs_at_the_sync():
    """sgx_scan_block / sgx_scan_2bit_dev leave the exact dense pass (g_pos / g_neg over all N: normally without a
    single variant) to the lane's next sync, which reads the step's counters and launches it if a variant asked
    for it.  With every flagged variant forced onto that pass, the device-resident calls -- one lane and two, the
    lanes reused before any explicit sync -- must give the host call's rows."""
    import torch
    sm, packed = _synthetic_case(3001, 2400, "binary", 0.05, seed=37)
    ref, ref_valid = _oracle(sm).scan_2bit(packed)
    dev = torch.device("cuda", 0)
    with _scanner(sm) as sc:
        sc.set_option("force_dense", 1)
        host, host_valid = sc.scan_2bit(packed)
        assert sc.stats()["n_spa_dense"] > 10
        bpv = sc.row_stride()
        pk = torch.zeros((4, 600, bpv), dtype=torch.uint8, device=dev)
        pk[:, :, :packed.shape[1]] = torch.from_numpy(packed.reshape(4, 600, -1)).to(dev)
        for lanes in (1, 2):
            sc.set_option("lanes", lanes)
            out = torch.full((4, 600, 8), -1.0, dtype=torch.float64, device=dev)
            valid = torch.zeros((4, 600), dtype=torch.uint8, device=dev)
            sc.stats_total(reset=True)
            for b in range(4):
                sc.scan_2bit_dev(pk[b].data_ptr(), bpv, 600, out[b].data_ptr(), valid[b].data_ptr())
            tot, ncalls = sc.stats_total(reset=True)          # (syncs)
            assert ncalls == 4 and tot["n_spa_dense"] > 10
            o, v = out.cpu().numpy().reshape(-1, 8), valid.cpu().numpy().reshape(-1)
            assert np.array_equal(v, host_valid)
            assert np.array_equal(np.nan_to_num(o, nan=-7.0), np.nan_to_num(host, nan=-7.0)), f"lanes={lanes}"
        sc.set_option("lanes", 1)
    assert_table_close(host, host_valid, ref, ref_valid, what="force_dense, host call")


def test_mfma_lane_map_selftest():
    """v_mfma_i32_16x16x64_i8 operand/result lane maps assumed by the score kernel."""
    from saigegds_amd import _lib
    _lib.check(_lib.load().sgx_selftest(0))


def test_scan_sharded_single_rank_rccl():
    """dist.scan_sharded on a one-rank RCCL group: blocks on two lanes, then the gather."""
    import torch
    import torch.distributed as dist
    from saigegds_amd.dist import scan_sharded
    sm, packed = _synthetic_case(2000, 1300, "binary", 0.1, seed=41)
    ref, ref_valid = _oracle(sm).scan_2bit(packed)
    dev = torch.device("cuda", 0)
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1, device_id=dev)
        created = True
    try:
        with _scanner(sm) as sc:
            bpv = sc.row_stride()
            pk = torch.zeros((1300, bpv), dtype=torch.uint8, device=dev)
            pk[:, :packed.shape[1]] = torch.from_numpy(packed).to(dev)
            out, valid = scan_sharded(sc, pk, bpv, block=500)      # 3 blocks, ragged tail
        assert_table_close(out.cpu().numpy(), valid.cpu().numpy(), ref, ref_valid, what="scan_sharded")
    finally:
        if created:
            dist.destroy_process_group()


def test_hard_calls_many_short_rows_and_odd_chunks():
    """Two corners of the host pipeline: (a) RAW / INTEGER hard calls with more than 65 535 rows in one call at
    a small N (the device pack kernel carries the row in grid.y: the chunk is capped), (b) INTEGER rows with a
    non-call value at an odd N and an odd number of rows per chunk (the doubles behind the INTEGER rows start
    on a 16-byte boundary whatever the chunk)."""
    n, m = 333, 70_000
    sm, packed = _synthetic_case(n, 1500, "binary", 0.2, seed=5)
    from saigegds_amd.gds import unpack_dosage_2bit
    codes = unpack_dosage_2bit(packed, n)
    reps = -(-m // codes.shape[0])
    big = np.tile(codes, (reps, 1))[:m]
    u8 = big.copy()
    u8[big == 3] = 0xFF
    ref, ref_valid = _oracle(sm).scan_2bit(packed)
    with _scanner(sm) as sc:
        out, valid = sc.scan_u8(u8)                 # one call, 70 000 rows, default chunk size
        assert sc.stats()["n_variants"] == m
        k = codes.shape[0]
        for r in range(reps):
            lo, hi = r * k, min(m, (r + 1) * k)
            assert np.array_equal(valid[lo:hi], ref_valid[:hi - lo])
        assert_table_close(out[:k], valid[:k], ref, ref_valid, what="u8, 70 000 rows")
        assert np.array_equal(np.nan_to_num(out[:k], nan=-7.0), np.nan_to_num(out[k:2 * k], nan=-7.0))
        # (b) odd N, odd rows per chunk, one value that is not a call
        i32 = codes[:601].astype(np.int32)
        i32[codes[:601] == 3] = -2147483648
        i32[5, 7] = 4
        u8b = u8[:601].copy()
        u8b[5, 7] = 4
        refb, refb_valid = _oracle(sm).scan_u8(u8b)
        sc.set_option("pipe_mb", 1)                 # 1 MiB / (12 x 333 B) = 262 rows per chunk: 262 x 333 x 4 is not a multiple of 16
        outb, validb = sc.scan_i32(i32)
        assert_table_close(outb, validb, refb, refb_valid, what="i32, odd N, odd chunk, a non-call value")
