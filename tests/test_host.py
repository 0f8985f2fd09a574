"""CPU tests: host logic, file formats, ABI surface (no GPU compute)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, REFERENCE, ROOT, load_null_model


def test_abi_exports_every_declared_symbol():
    """libsaigehip.so loads and exports exactly what include/saigehip.h declares."""
    from saigegds_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "saigehip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(sgx_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS)
    L = _lib.load()
    for s in declared:
        assert hasattr(L, s), s
    assert b"gfx950" in L.sgx_version()
    assert L.sgx_row_stride(430000) == 107520 and L.sgx_row_stride(1000) == 256
    # struct layouts agree with the header (sizes the C side computes)
    assert ctypes.sizeof(_lib.SgxModel) == 4 * 4 + 7 * 8 + 10 * 8
    assert ctypes.sizeof(_lib.SgxStats) == 5 * 8 + 3 * 4 + 2 * 4 + 2 * 4 + 3 * 4     # + ms_kernel, ms_lists; three_plane, n_unlisted, n_guarded


def test_init_rejects_bad_models_without_gpu():
    from saigegds_amd import _lib
    L = _lib.load()
    h = ctypes.c_void_p()
    m = _lib.SgxModel()
    assert L.sgx_init(None, 0, ctypes.byref(h)) == -1
    m.n_samp, m.n_coeff = 10, 99
    assert L.sgx_init(ctypes.byref(m), 0, ctypes.byref(h)) == -1
    assert b"n_coeff" in L.sgx_last_error()


def test_model_consistency_diagnostic(monkeypatch):
    """SAIGEHIP_CHECK_MODEL=1: the two arrays of the list the scan does not read (t_XXVX_inv, XV) are held
    against the ones it reads; both golden models pass (the call then gets as far as looking for a device),
    a model whose XV belongs to other weights is refused by name."""
    import torch
    from saigegds_amd import _lib
    from saigegds_amd.nullmod import init_nullmod
    monkeypatch.setenv("SAIGEHIP_CHECK_MODEL", "1")
    for fn in ("saige_model.npz", "saige_model_quant.npz"):
        mod = load_null_model(fn)
        sm = init_nullmod(mod, np.arange(len(mod.sample_id)), 0, 0, 1, 0.05, 1.0)
        if not torch.cuda.is_available():
            with pytest.raises(_lib.SgxError, match="no HIP device") as e:
                _lib.Scanner(sm)
            assert e.value.code == -4
        else:
            _lib.Scanner(sm).close()
        sm.XV = sm.XV.copy()
        sm.XV[7, 1] *= 1.001
        with pytest.raises(_lib.SgxError, match=r"XV\[1,7\]") as e:
            _lib.Scanner(sm)
        assert e.value.code == -1


def test_init_nullmod_matches_reference_layout():
    """.init_nullmod (R/assoc_single.r:17-67): shapes, orientation, derived terms."""
    from saigegds_amd.nullmod import init_nullmod, ModelError
    mod = load_null_model("saige_model.npz")
    n = len(mod.sample_id)
    perm = np.random.default_rng(0).permutation(n)
    sm = init_nullmod(mod, perm, float("nan"), 4, 0.1, 0.05, 0.9)
    assert sm.k == 3 and sm.n == n and sm.t_X.shape == (n, 3) and sm.XVX.shape == (3, 3)
    assert np.array_equal(sm.y, mod.y[perm]) and np.array_equal(sm.mu, mod.fitted_values[perm])
    assert np.array_equal(sm.y_mu, (mod.y - mod.fitted_values)[perm])
    assert np.array_equal(sm.XV, mod.XV[:, perm].T)
    assert np.allclose(sm.t_XVX_inv_XV, mod.XXVX_inv[perm] * mod.V[perm, None], rtol=0, atol=0)
    assert np.allclose(sm.XVX, sm.t_X.T @ (sm.t_X * sm.mu2[:, None]))
    assert np.allclose(sm.S_a, (sm.t_X * sm.y_mu[:, None]).sum(0))
    with pytest.raises(ModelError, match="Invalid variance ratio"):
        init_nullmod(mod, perm, float("nan"), 4, 0.1, 0.05, float("nan"))
    q = init_nullmod(load_null_model("saige_model_quant.npz"), np.arange(n), 0, 0, 1, 0.05, 1.0)
    assert np.allclose(q.XVX, q.t_X.T @ q.t_X)


def test_assoc_argument_checks():
    """stopifnot()s and messages of seqAssocGLMM_SPA (R/assoc_single.r:96-107,138-142)."""
    from saigegds_amd import GenotypeSource, seqAssocGLMM_SPA
    from saigegds_amd.nullmod import ModelError
    mod = load_null_model("saige_model.npz")
    z = np.load(os.path.join(GOLDEN, "grm1k_10k_snp.npz"))
    src = GenotypeSource([str(s) for s in z["sample_id"]], packed=z["packed"][:10])
    with pytest.raises(TypeError):
        seqAssocGLMM_SPA(src, mod, maf="x", verbose=False)
    with pytest.raises(ValueError, match="res.compress"):
        seqAssocGLMM_SPA(src, mod, res_compress="bz2", verbose=False)
    with pytest.raises(TypeError):
        seqAssocGLMM_SPA(12, mod, verbose=False)
    short = GenotypeSource([str(s) for s in z["sample_id"][:900]],
                           packed=np.zeros((10, 225), np.uint8))
    with pytest.raises(ModelError, match="Some of sample IDs are not available"):
        seqAssocGLMM_SPA(short, mod, verbose=False)
    with pytest.raises(ModelError):
        seqAssocGLMM_SPA(src, 3.5, verbose=False)


def test_shard_ranges_cover_in_order():
    from saigegds_amd.dist import shard_range
    for m in (0, 1, 7, 8, 9, 10_000_000):
        for w in (1, 2, 3, 8):
            r = [shard_range(m, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == m
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def test_result_writers_roundtrip(tmp_path):
    from saigegds_amd.rds import read_rds
    from saigegds_amd.results import save_result
    z = np.load(os.path.join(GOLDEN, "saige_pval.npz"))
    ans = {"id": z["id"], "chr": [str(s) for s in z["chr"]], "AF.alt": z["AF_alt"],
           "num": z["num"].astype(np.int32), "pval": z["pval"], "converged": z["converged"]}
    for cm in ("LZMA", "ZIP", "none"):
        fn = str(tmp_path / "res.rds")
        save_result(ans, fn, cm)
        r = read_rds(fn)
        assert r.rclass() == ["data.frame"] and r.names == list(ans)
        assert np.array_equal(np.asarray(r["pval"]), ans["pval"])
        assert list(r["chr"]) == ans["chr"] and np.array_equal(np.asarray(r["converged"]), ans["converged"])
    with pytest.raises(ValueError, match="Unknown format"):
        save_result(ans, str(tmp_path / "res.txt"))
    # GDS SAIGE_OUTPUT container (R/assoc_single.r:243-286), read back with the GDS decoder
    from saigegds_amd.gds import GdsFile
    full = dict(ans)
    full.update({"rs.id": [f"rs{i}" for i in range(len(z["id"]))], "beta": z["beta"], "p.norm": z["p_norm"]})
    for cm in ("LZMA", "none"):
        fn = str(tmp_path / f"res_{cm}.gds")
        save_result(full, fn, cm, sample_id=[f"s{i}" for i in range(7)])
        g = GdsFile(fn)
        # same nodes in the order the reference writes them
        assert g.ls() == ["sample.id", "id", "chr", "rs.id", "AF.alt", "num", "beta", "pval", "p.norm", "converged"]
        assert g.sample_id() == [f"s{i}" for i in range(7)]
        assert list(g.read("chr")) == full["chr"] and list(g.read("rs.id")) == full["rs.id"]
        for k in ("AF.alt", "beta", "pval", "p.norm"):
            assert g.node(k).cls == "dFloat64" and np.array_equal(np.asarray(g.read(k)), full[k], equal_nan=True)
        assert g.node("num").cls == "dInt32" and np.array_equal(np.asarray(g.read("num")), full["num"])
        assert np.array_equal(np.asarray(g.read("converged")) == 1, full["converged"])
        assert np.array_equal(g.read_rows("pval", 4000, 4100), full["pval"][4000:4100])     # through the block index
        root = g.stream(g.root_id)
        assert b"\nFileFormat\x0e\x0cSAIGE_OUTPUT" in root and b"\x07Version\x0e" in root
        assert b"\tR.logical\x00" in g.stream(g.node("converged").block_id)


@pytest.mark.reference
def test_gds_writer_emits_the_reference_files_records():
    """A node written by GdsWriter with the data of a node of the reference's own file is that node's
    stream byte for byte (up to the stream id, the stored size and SeqArray's md5 attribute)."""
    import tempfile
    from saigegds_amd.gds import GdsFile
    from saigegds_amd.gds_write import GdsWriter
    g = GdsFile(REFERENCE + "/inst/extdata/assoc_100snp.gds")
    with tempfile.TemporaryDirectory() as d:
        w = GdsWriter(os.path.join(d, "t.gds"))
        w.put_attr("FileFormat", "SEQ_ARRAY")
        w.put_attr("FileVersion", "v1.0")
        w.add("sample.id", g.sample_id())
        w.add("position", np.asarray(g.read("position")))
        w.close()
        t = GdsFile(os.path.join(d, "t.gds"))
        mine, ref = t.stream(t.node("sample.id").block_id), g.stream(g.node("sample.id").block_id)
        k = ref.find(b"\x02\xc9FmP\xe0Q%\x04\x10") + 10 + 8       # PIPE_SIZE: raw size equal, stored size may differ
        assert len(mine) == len(ref) and mine[:k] == ref[:k] and mine[k + 8:-19] == ref[k + 8:-19]
        assert mine[-11:] == ref[-11:]                                  # attribute count record
        mine, ref = t.stream(t.node("position").block_id), g.stream(g.node("position").block_id)
        k = ref.find(b"\x02\xc9FmP\xe0Q%\x04\x10") + 10 + 8
        assert mine[8:k] == ref[8:k]                                    # class, pipe, raw size (the reference's node has one more record: md5)
        assert np.array_equal(t.read("position"), g.read("position")) and t.sample_id() == g.sample_id()
        # root folder: directory records and the attribute block as SeqArray's own root has them
        assert t.stream(t.root_id)[-46:] == g.stream(g.root_id)[-46:]


def test_synthetic_generator_is_counter_based():
    from saigegds_amd import synth
    thr = synth.variant_thresholds(0, 64, 5)
    a = synth.synth_packed(777, 0, 64, 5, thr)
    b = synth.synth_packed(777, 32, 32, 5, thr[32:])
    assert np.array_equal(a[32:], b)                 # any shard regenerates its rows
    assert a.shape[1] % 16 == 0 and not a[:, 195:].any()   # padding beyond ceil(N/4) is zero


@pytest.mark.reference
def test_formats_against_reference_files():
    """RDS + GDS decoders reproduce the committed fixtures from the raw files."""
    from saigegds_amd.gds import GdsFile
    from saigegds_amd.nullmod import load_modobj
    g = GdsFile(REFERENCE + "/inst/extdata/grm1k_10k_snp.gds")
    packed, n, m = g.dosage_alt_packed()
    z = np.load(os.path.join(GOLDEN, "grm1k_10k_snp.npz"))
    assert (n, m) == (1000, 10000) and np.array_equal(packed, z["packed"])
    assert g.sample_id()[:2] == ["s1", "s2"] and g.read("annotation/id")[0] == "rs1"
    mod = load_modobj(REFERENCE + "/inst/unitTests/saige_model_quant.rds")
    ref = load_null_model("saige_model_quant.npz")
    assert mod.trait_type == "quantitative" and np.array_equal(mod.y, ref.y)
    g2 = GdsFile(REFERENCE + "/inst/extdata/assoc_100snp.gds")
    assert not g2.has_genotype()
    ds = g2.dosage_real()
    assert ds.shape == (100, 1000) and set(np.unique(ds)) == {0.0, 1.0, 2.0}


@pytest.mark.reference
def test_gds_row_ranges_equal_whole_reads():
    """Ranged reads through the LZMA_RA block index (what the streaming driver uses) give the same
    bytes as decoding the whole node, across block borders and for sample subsets."""
    from saigegds_amd.gds import GdsFile, unpack_dosage_2bit
    g = GdsFile(REFERENCE + "/inst/extdata/grm1k_10k_snp.gds")
    z = np.load(os.path.join(GOLDEN, "grm1k_10k_snp.npz"))
    for lo, hi in ((0, 1), (4700, 4800), (4718, 4719), (9437, 9440), (9990, 10000), (0, 10000)):
        assert np.array_equal(g.dosage_alt_packed_range(lo, hi), z["packed"][lo:hi]), (lo, hi)
    sel = np.array([5, 3, 999, 0, 17])
    got = unpack_dosage_2bit(g.dosage_alt_packed_range(4000, 5000, sel), 5)
    assert np.array_equal(got, unpack_dosage_2bit(z["packed"][4000:5000], 1000)[:, sel])
    assert np.array_equal(g.read_rows("position", 100, 250), np.asarray(g.read("position"))[100:250])
    g2 = GdsFile(REFERENCE + "/inst/extdata/assoc_100snp.gds")
    assert np.array_equal(g2.dosage_real_range("annotation/format/DS", 10, 37), g2.dosage_real()[10:37])


def test_dosage_alt_counts_every_non_reference_allele():
    """$dosage_alt of a site with alleles 0/1/2 stored in 2 bits: allele 2 counts as non-reference,
    allele code 3 makes the sample missing (SeqArray's $dosage_alt, R/assoc_single.r:69-85)."""
    from saigegds_amd.gds import GdsFile
    lut = GdsFile._nibble_lut()

    def byte(s0, s1):      # two samples' allele pairs in one byte of genotype/data
        return s0[0] | (s0[1] << 2) | (s1[0] << 4) | (s1[1] << 6)
    cases = {((0, 0), (0, 1)): (0, 1), ((1, 1), (0, 2)): (2, 1), ((2, 2), (1, 2)): (2, 2),
             ((3, 1), (2, 0)): (3, 1), ((0, 3), (3, 3)): (3, 3)}
    for (s0, s1), (d0, d1) in cases.items():
        v = int(lut[byte(s0, s1)])
        assert (v & 3, v >> 2) == (d0, d1), (s0, s1)


def test_sites_with_more_than_three_alleles(tmp_path):
    """genotype/@data > 1: the allele index of such a site takes several rows of 2-bit digits (all bits set =
    missing); $dosage_alt counts the non-reference alleles whatever their index.  Parity unpinned: the
    reference's fixtures hold biallelic sites only; the layout is SeqArray's as documented."""
    from saigegds_amd.gds import GdsFile, unpack_dosage_2bit
    from saigegds_amd.gds_write import write_seqarray_alleles
    rng = np.random.default_rng(1)
    for n in (37, 40):                                   # rows that are and are not whole bytes
        m = 23
        al = rng.integers(0, 2, size=(m, n, 2))
        for v, top in ((3, 3), (4, 5), (10, 14), (11, 15), (17, 40)):       # 2, 2, 2, 3 and 3 rows
            al[v] = rng.integers(0, top + 1, size=(n, 2))
            al[v, 0, 0] = top
        al[rng.random((m, n)) < 0.1] = -1
        al[5, 7, 1] = -1                                 # one allele missing makes the sample missing
        want = (al[:, :, 0] > 0).astype(int) + (al[:, :, 1] > 0)
        want[(al < 0).any(2)] = 3
        for comp in ("none", "LZMA"):
            fn = str(tmp_path / f"multi_{n}_{comp}.gds")
            write_seqarray_alleles(fn, al, compress=comp, ra_block=64)
            g = GdsFile(fn)
            assert g.genotype_dims() == (m, n) and g.node("genotype/data").dims[0] == 23 + 1 * 3 + 2 * 2
            assert np.array_equal(unpack_dosage_2bit(g.dosage_alt_packed_range(0, m), n), want)
            assert np.array_equal(unpack_dosage_2bit(g.dosage_alt_packed_range(5, 9), n), want[5:9])       # single rows, offset
            rev = np.arange(n)[::-1]
            assert np.array_equal(unpack_dosage_2bit(g.dosage_alt_packed_range(2, 13, rev), n), want[2:13, ::-1])


def test_scan_blocks_runs_one_thread_per_gpu():
    """The block loop of seqAssocGLMM_SPA with parallel = 2: two scanners work at the same time, every
    block lands at its own offset, a failing block stops the scan with its error."""
    import threading
    import time
    from saigegds_amd.assoc import scan_blocks
    state = {"live": 0, "peak": 0, "made": []}
    lock = threading.Lock()

    class FakeScanner:
        def __init__(self, d):
            self.n, self.d = 7, d
            state["made"].append(d)

        def scan_2bit(self, blk):
            with lock:
                state["live"] += 1
                state["peak"] = max(state["peak"], state["live"])
            time.sleep(0.05)
            with lock:
                state["live"] -= 1
            o = np.repeat(blk[:, :1].astype(np.float64), 8, axis=1)
            return o, np.ones(blk.shape[0], np.uint8)

        def close(self):
            pass

    m = 1050
    blocks = [(o, min(m, o + 100)) for o in range(0, m, 100)]
    out, valid = np.zeros((m, 8)), np.zeros(m, np.uint8)
    scan_blocks(FakeScanner, 2, blocks, lambda lo, hi: (np.arange(lo, hi) % 251).astype(np.uint8)[:, None], True, out, valid)
    assert sorted(state["made"]) == [0, 1] and state["peak"] == 2
    assert valid.all() and np.array_equal(out[:, 0], np.arange(m) % 251)

    def bad_read(lo, hi):
        if lo == 300:
            raise RuntimeError("decode failed")
        return np.zeros((hi - lo, 1), np.uint8)
    with pytest.raises(RuntimeError, match="decode failed"):
        scan_blocks(FakeScanner, 2, blocks, bad_read, True, out, valid)


def test_scan_blocks_decodes_one_block_ahead():
    """Each scanner's decoder thread works on block i + 1 while block i is being scanned: with a decode and a
    scan of equal length the wall time is ~ (n + 1) x one of them, not 2 n; the timing dict adds them up."""
    import time
    from saigegds_amd.assoc import scan_blocks
    events = []

    class FakeScanner:
        def __init__(self, d):
            pass

        def scan_2bit(self, blk):
            events.append(("scan+", int(blk[0, 0])))
            time.sleep(0.06)
            events.append(("scan-", int(blk[0, 0])))
            return np.zeros((blk.shape[0], 8)), np.ones(blk.shape[0], np.uint8)

        def close(self):
            pass

    def read(lo, hi):
        events.append(("dec+", lo // 10))
        time.sleep(0.06)
        events.append(("dec-", lo // 10))
        return np.full((hi - lo, 1), lo // 10, np.uint8)
    n = 6
    blocks = [(10 * i, 10 * i + 10) for i in range(n)]
    out, valid, timing = np.zeros((10 * n, 8)), np.zeros(10 * n, np.uint8), {}
    t = time.perf_counter()
    scan_blocks(FakeScanner, 1, blocks, read, True, out, valid, timing)
    wall = time.perf_counter() - t
    assert wall < 0.06 * (2 * n) * 0.8, wall                    # overlapped: far from the serial 2 n x 60 ms
    assert valid.all() and timing["decode_s"] >= 0.06 * n * 0.9 and timing["scan_s"] >= 0.06 * n * 0.9
    # block i + 1 starts decoding before block i has been scanned
    for i in range(n - 1):
        assert events.index(("dec+", i + 1)) < events.index(("scan-", i)), (i, events)


def test_gds_codecs_and_bounded_block_decode(tmp_path):
    """genotype/data through every storage the reader knows (raw, LZMA_RA, ZIP_RA, LZ4_RA), an odd sample
    count, a sample subset taken in the 2-bit domain, a caller's buffer, tiny work pieces -- all equal to
    the in-memory codes; 8- and 16-bit packed reals."""
    import tracemalloc
    from saigegds_amd import synth
    from saigegds_amd.gds import GdsFile, pack_dosage_2bit, unpack_dosage_2bit
    from saigegds_amd.gds_write import GdsWriter, write_seqarray_genotypes
    for n in (1003, 1000):
        m = 150
        pk = synth.synth_packed(n, 0, m, 5, synth.variant_thresholds(0, m, 5))[:, :(n + 3) // 4]
        codes = unpack_dosage_2bit(pk, n)
        sel = np.arange(n)[::-3].copy()
        for comp in ("none", "LZMA_RA", "ZIP_RA", "LZ4_RA"):
            fn = str(tmp_path / f"g_{n}_{comp}.gds")
            write_seqarray_genotypes(fn, pk, n, compress=comp, ra_block=30_000)
            f = GdsFile(fn)
            assert f.genotype_dims() == (m, n) and len(f.sample_id()) == n
            got, N, M = f.dosage_alt_packed()
            assert (N, M) == (n, m) and np.array_equal(got, pk)
            sub = f.dosage_alt_packed_range(10, 131, sel, chunk_bytes=4000)
            assert np.array_equal(sub, pack_dosage_2bit(codes[10:131][:, sel]))
            buf = np.full((200, (n + 3) // 4 + 5), 0xAB, np.uint8)
            res = f.dosage_alt_packed_range(0, m, out=buf)
            assert np.array_equal(res[:, :(n + 3) // 4], pk) and not res[:, (n + 3) // 4:].any()
    # the work pieces bound the temporaries: a block of 4 000 x 20 000 with a sample subset stays far under
    # the 80 MB of one byte per (variant, sample)
    n, m = 20_000, 4_000
    pk = np.zeros((m, n // 4), np.uint8)
    pk[:, ::7] = 0x61
    fn = str(tmp_path / "big.gds")
    write_seqarray_genotypes(fn, pk, n)
    f = GdsFile(fn)
    sel = np.arange(0, n, 2)
    tracemalloc.start()
    sub = f.dosage_alt_packed_range(0, m, sel, chunk_bytes=4 << 20)
    peak = tracemalloc.get_traced_memory()[1]
    tracemalloc.stop()
    assert np.array_equal(sub[:50], pack_dosage_2bit(unpack_dosage_2bit(pk[:50], n)[:, sel]))
    assert peak < 40 << 20, peak
    # packed reals
    w = GdsWriter(str(tmp_path / "ds.gds"))
    raw16 = np.array([[0, 1000, 20000, 0xFFFF], [5, 6, 7, 8]], dtype="<u2")
    raw8 = np.array([[0, 127, 254, 0xFF]], dtype=np.uint8)
    w.add("annotation/format/DS/data", raw16, "ZIP_RA", cls="dPackedReal16U", dims=raw16.shape, scale=1e-4, offset=0.0)
    w.add("annotation/format/D8/data", raw8, "none", cls="dPackedReal8U", dims=raw8.shape, scale=1 / 127, offset=0.0)
    w.add("sample.id", ["a", "b", "c", "d"], "none")
    w.close()
    g = GdsFile(str(tmp_path / "ds.gds"))
    d16 = g.dosage_real_range("annotation/format/DS", 0, 2)
    assert np.allclose(d16[0, :3], [0, 0.1, 2.0]) and np.isnan(d16[0, 3]) and np.allclose(d16[1], np.array([5, 6, 7, 8]) * 1e-4)
    assert np.array_equal(g.read("annotation/format/DS/data"), d16, equal_nan=True)
    d8 = g.read("annotation/format/D8/data")
    assert np.allclose(d8[0, :3], [0, 1, 2]) and np.isnan(d8[0, 3])


def test_model_files_rds_and_rda(tmp_path):
    """.check_modobj loads .rda / .RData as well as .rds (R/saige_main.r:93-111): a model saved in both forms
    comes back identical, field by field, and equal to what was saved."""
    from types import SimpleNamespace
    from saigegds_amd.nullmod import load_modobj
    from saigegds_amd.results import save_model
    mod = load_null_model("saige_model.npz")
    n = len(mod.sample_id)
    nr = len(mod.var_ratio)
    full = SimpleNamespace(
        coefficients=np.array([0.1, -0.2, 0.3]), coef_names=["(Intercept)", "x1", "x2"], tau=mod.tau,
        linear_predictors=np.linspace(-2, 2, n), fitted_values=mod.fitted_values, residuals=mod.y - mod.fitted_values,
        cov=np.eye(3), converged=True, y=mod.y, mu_noK=mod.fitted_values * 0.99, res_noK=mod.y - mod.fitted_values * 0.99,
        V=mod.V, X1=mod.X1, XV=mod.XV, XXVX_inv=mod.XXVX_inv, trait_type=mod.trait_type, sample_id=list(mod.sample_id),
        variant_id=np.arange(1, 6), var_ratio_table={"id": np.arange(1, nr + 1), "maf": np.full(nr, 0.1), "mac": np.full(nr, 30.0),
                                                     "var1": np.ones(nr), "var2": np.ones(nr), "ratio": np.asarray(mod.var_ratio)})
    a, b, c = str(tmp_path / "m.rds"), str(tmp_path / "m.rda"), str(tmp_path / "m.RData")
    for fn in (a, b, c):
        save_model(full, fn)
    ma, mb, mc = load_modobj(a), load_modobj(b), load_modobj(c)
    for got in (ma, mb, mc):
        assert got.trait_type == mod.trait_type and [str(x) for x in got.sample_id] == [str(x) for x in mod.sample_id]
        for fld in ("tau", "fitted_values", "var_ratio", "y", "V", "X1", "XV", "XXVX_inv"):
            assert np.array_equal(np.asarray(getattr(got, fld)), np.asarray(getattr(mod, fld))), fld
        assert np.array_equal(got.coefficients, full.coefficients)
    with pytest.raises(Exception):
        load_modobj(str(tmp_path / "m.txt"))
