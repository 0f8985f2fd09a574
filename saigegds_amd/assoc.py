"""Host driver of the single-variant scan: Python mirror of
``seqAssocGLMM_SPA()`` (reference R/assoc_single.r:92-334) over the C ABI.

Same arguments, same checks and messages, same result columns.  ``parallel``
is reinterpreted as the number of GPUs driven from this process (the reference
forks SeqArray worker processes instead, R/assoc_single.r:167-204).
"""
from __future__ import annotations

import math
from typing import Any, Dict, List, Optional, Union

import time

import numpy as np

from .gds import GdsError, GdsFile, pack_dosage_2bit, unpack_dosage_2bit
from .nullmod import ModelError, NullModel, ScanModel, init_nullmod, load_modobj

BLOCK_SIZE = 50_000   # .bl_size=50000L, R/assoc_single.r:204


class GenotypeSource:
    """In-memory stand-in for an opened SeqArray GDS file (synthetic data,
    tests): 2-bit packed ``$dosage_alt`` rows or real-valued dosages."""

    def __init__(self, sample_id: List[str], packed: Optional[np.ndarray] = None,
                 dosage: Optional[np.ndarray] = None, variant_id=None, chromosome=None,
                 position=None, rs_id=None, ref=None, alt=None):
        if (packed is None) == (dosage is None):
            raise ValueError("give exactly one of packed / dosage")
        self._sample_id = list(sample_id)
        self.packed, self.dosage = packed, dosage
        m = (packed if packed is not None else dosage).shape[0]
        self.variant_id = np.arange(1, m + 1) if variant_id is None else np.asarray(variant_id)
        self.chromosome = ["1"] * m if chromosome is None else list(chromosome)
        self.position = np.arange(1, m + 1) if position is None else np.asarray(position)
        self.rs_id = rs_id
        self.ref = ["A"] * m if ref is None else list(ref)
        self.alt = ["C"] * m if alt is None else list(alt)

    def sample_id(self):
        return self._sample_id


def _open_source(gdsfile, verbose):
    if isinstance(gdsfile, GenotypeSource):
        return gdsfile
    if isinstance(gdsfile, str):
        if verbose:
            print(f"    open '{gdsfile}'")
        gdsfile = GdsFile(gdsfile)
    if not isinstance(gdsfile, GdsFile):
        raise TypeError("inherits(gdsfile, \"SeqVarGDSClass\") | is.character(gdsfile) is not TRUE")
    return gdsfile


def _dsnode(src, nm: str) -> str:
    """``.dsnode`` (R/assoc_single.r:69-85)."""
    if isinstance(src, GenotypeSource):
        return "$dosage_alt" if src.packed is not None else "annotation/format/DS"
    if nm == "":
        if src.node("genotype/data", silent=True) is not None:
            return "$dosage_alt"
        nm = "annotation/format/DS"
        if src.node(nm, silent=True) is None:
            raise GdsError("Dosages should be stored in genotype or annotation/format/DS.")
    return nm


def _pretty(n: int) -> str:
    return f"{n:,}"


def _is_num(x) -> bool:
    return isinstance(x, (int, float, np.integer, np.floating)) and not isinstance(x, bool)


def seqAssocGLMM_SPA(gdsfile: Union[str, GdsFile, GenotypeSource], modobj: Any, maf: float = float("nan"),
                     mac: float = 10, missing: float = 0.1, dsnode: str = "", spa_pval: float = 0.05,
                     var_ratio: float = float("nan"), res_savefn: str = "", res_compress: str = "LZMA",
                     parallel: Union[bool, int] = False, verbose: bool = True, timing: Optional[Dict[str, float]] = None):
    """SAIGE single-variant association scan on MI355X.

    Returns a ``dict`` of equal-length columns (``pandas.DataFrame(result)``
    gives the reference's data.frame): id, chr, pos, [rs.id], ref, alt, AF.alt,
    mac, num, beta, SE, pval and, for binary traits, p.norm, converged
    (man/seqAssocGLMM_SPA.Rd:60-74).
    """
    # argument checks, R/assoc_single.r:96-107
    for nm, v in (("maf", maf), ("mac", mac), ("missing", missing), ("spa.pval", spa_pval),
                  ("var.ratio", var_ratio)):
        if not _is_num(v):
            raise TypeError(f"is.numeric({nm}) is not TRUE")
    if not isinstance(dsnode, str):
        raise TypeError("is.character(dsnode) is not TRUE")
    if not isinstance(res_savefn, str):
        raise TypeError("is.character(res.savefn) is not TRUE")
    if res_compress not in ("LZMA", "LZMA_RA", "ZIP", "ZIP_RA", "none"):
        raise ValueError("`res.compress` should be one of LZMA, LZMA_RA, ZIP, ZIP_RA and none.")
    if not isinstance(verbose, bool):
        raise TypeError("is.logical(verbose) is not TRUE")
    if verbose:
        print("SAIGE association analysis:")

    mod: NullModel = load_modobj(modobj, verbose)
    src = _open_source(gdsfile, verbose)
    node = _dsnode(src, dsnode)

    # sample matching, R/assoc_single.r:135-142
    gsid = [str(s) for s in src.sample_id()]
    pos = {str(s): i for i, s in enumerate(mod.sample_id)}
    sel = [i for i, s in enumerate(gsid) if s in pos]
    if len(sel) != len(mod.sample_id):
        raise ModelError("Some of sample IDs are not available in the GDS file.")
    ii = np.array([pos[gsid[i]] for i in sel], dtype=np.int64)
    sel = np.asarray(sel, dtype=np.int64)
    n_samp = sel.size

    # where the genotypes come from; nothing is decoded yet (the reference never holds more than one
    # block of .bl_size variants either, R/assoc_single.r:202-209)
    if isinstance(src, GenotypeSource):
        kind = "packed" if src.packed is not None else "dosage"
        n_all = len(gsid)
        n_var = (src.packed if src.packed is not None else src.dosage).shape[0]
    elif node == "$dosage_alt":
        kind = "packed"
        n_var, n_all = src.genotype_dims()
    else:
        kind = "dosage"
        n_var, n_all = src.node(node + "/data").dims[:2]
    if verbose:
        print(f"    # of samples: {_pretty(n_samp)}")
        print(f"    # of variants: {_pretty(n_var)}")
        print(f"    MAF threshold: {maf}")
        print(f"    MAC threshold: {mac}")
        print(f"    missing threshold for variants: {missing}")
        print(f"    p-value threshold for SPA adjustment: {spa_pval}")
    if not math.isfinite(var_ratio):
        var_ratio = float(np.nanmean(mod.var_ratio))
    if verbose:
        print(f"    variance ratio for approximation: {var_ratio}")
    if n_samp <= 0:
        raise ValueError("No sample in the genotypic data set!")
    if n_var <= 0:
        raise ValueError("No variant in the genotypic data set!")
    all_samples = n_samp == n_all and np.array_equal(sel, np.arange(n_all))

    def read_block(off: int, end: int):
        """Genotypes of variants [off, end) for the selected samples."""
        if isinstance(src, GenotypeSource):
            if kind == "packed":
                blk = src.packed[off:end]
                if not all_samples:
                    blk = pack_dosage_2bit(unpack_dosage_2bit(blk, n_all)[:, sel])
                return blk
            blk = src.dosage[off:end]
            return blk if all_samples else np.ascontiguousarray(blk[:, sel])
        if kind == "packed":
            return src.dosage_alt_packed_range(off, end, None if all_samples else sel)
        blk = src.dosage_real_range(node, off, end)
        return blk if all_samples else np.ascontiguousarray(blk[:, sel])

    mobj: ScanModel = init_nullmod(mod, ii, maf, mac, missing, spa_pval, var_ratio)
    if mod.trait_type not in ("binary", "quantitative"):
        raise ModelError("Invalid 'modobj$trait.type'.")

    # devices, R/assoc_single.r:167-171 ('parallel' = number of GPUs here)
    from ._lib import Scanner, load
    ngpu = 1 if parallel in (False, None, 0, 1) else int(parallel)
    ndev = load().sgx_device_count()
    if ndev <= 0:
        raise RuntimeError("seqAssocGLMM_SPA: no MI355X device is visible (there is no CPU fallback)")
    if ngpu > ndev:
        raise ValueError(f"parallel={ngpu} but only {ndev} GPU(s) are visible")
    if verbose:
        print(f"    # of GPUs: {ngpu}")

    # scan by blocks of .bl_size variants, R/assoc_single.r:199-223; results land at the blocks' own
    # offsets, so the table keeps the file's order whatever GPU finishes first (:226-227)
    out = np.empty((n_var, 8), dtype=np.float64)
    valid = np.zeros(n_var, dtype=np.uint8)
    blocks = [(off, min(n_var, off + BLOCK_SIZE)) for off in range(0, n_var, BLOCK_SIZE)]
    t_loop = time.perf_counter()
    scan_blocks(lambda d: Scanner(mobj, device=d), ngpu, blocks, read_block, kind == "packed", out, valid, timing)
    if timing is not None:
        timing["blocks_s"] = time.perf_counter() - t_loop      # handles + every block (decode and scan overlapped)

    x = valid.astype(bool)           # R/assoc_single.r:225-234
    if verbose:
        print("# of variants after filtering by MAF, MAC and missing thresholds: "
              f"{_pretty(int(x.sum()))}")
    ans = assemble_result(src, x, out, mod.trait_type)

    if res_savefn:
        from .results import save_result
        if verbose:
            print(f"Save to '{res_savefn}' ...")
        save_result(ans, res_savefn, res_compress, sample_id=[gsid[i] for i in sel])
        if verbose:
            print("Done.")
        return None
    if verbose:
        print("Done.")
    return ans


def scan_blocks(make_scanner, ngpu: int, blocks, read_block, packed_rows: bool, out: np.ndarray, valid: np.ndarray,
                timing: Optional[Dict[str, float]] = None):
    """One host thread per GPU (the reference forks one SeqArray worker per core, seqParallel,
    R/assoc_single.r:202-204): each thread owns a scanner and pulls blocks from a shared queue.  Every
    scanner has a decoder thread of its own one block ahead of it: while block i crosses PCIe and is
    scanned (the C ABI call releases the GIL), block i + 1 is being read and decoded (lzma / zlib and
    numpy release the GIL), so neither the GPU nor the decoder waits for the other beyond the slower of
    the two.  timing: seconds spent decoding / scanning (summed over threads) and decoded bytes."""
    import queue
    import threading
    import time
    from concurrent.futures import ThreadPoolExecutor
    todo: "queue.Queue" = queue.Queue()
    for b in blocks:
        todo.put(b)
    errors: List[BaseException] = []
    tlock = threading.Lock()

    def take():
        try:
            return todo.get_nowait()
        except queue.Empty:
            return None

    def decode(rng):
        t = time.perf_counter()
        blk = read_block(*rng)
        if timing is not None:
            with tlock:
                timing["decode_s"] = timing.get("decode_s", 0.0) + time.perf_counter() - t
                timing["decoded_bytes"] = timing.get("decoded_bytes", 0.0) + blk.nbytes
        return blk

    def worker(d: int):
        sc = None
        try:
            sc = make_scanner(d)
            with ThreadPoolExecutor(1, thread_name_prefix=f"sgx-decode{d}") as ex:
                rng = take()
                fut = ex.submit(decode, rng) if rng is not None else None
                while fut is not None and not errors:
                    off, end = rng
                    blk = fut.result()
                    rng = take()                                       # the next block decodes while this one is scanned
                    fut = ex.submit(decode, rng) if rng is not None else None
                    t = time.perf_counter()
                    if packed_rows:
                        o, v = sc.scan_2bit(blk)            # 2-bit packed rows
                    elif blk.dtype == np.uint8:
                        o, v = sc.scan_u8(blk)
                    elif np.issubdtype(blk.dtype, np.integer):
                        o, v = sc.scan_i32(blk)
                    else:
                        o, v = sc.scan_f64(blk)
                    out[off:end], valid[off:end] = o, v
                    if timing is not None:
                        with tlock:
                            timing["scan_s"] = timing.get("scan_s", 0.0) + time.perf_counter() - t
        except BaseException as e:      # noqa: BLE001 -- re-raised in the caller's thread
            errors.append(e)
        finally:
            if sc is not None:
                sc.close()

    if ngpu == 1:
        worker(0)
    else:
        threads = [threading.Thread(target=worker, args=(d,), name=f"sgx-gpu{d}") for d in range(ngpu)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    if errors:
        raise errors[0]


def assemble_result(src, keep: np.ndarray, out: np.ndarray, trait_type: str) -> Dict[str, Any]:
    """The data.frame of R/assoc_single.r:287-308 as an ordered dict of columns."""
    def sub(a):
        return [v for v, k in zip(a, keep) if k] if isinstance(a, list) else np.asarray(a)[keep]
    if isinstance(src, GenotypeSource):
        vid, chrom, posn, rs = src.variant_id, src.chromosome, src.position, src.rs_id
        ref, alt = src.ref, src.alt
    else:
        vid, chrom, posn = src.read("variant.id"), src.read("chromosome"), src.read("position")
        rs = src.read("annotation/id") if src.node("annotation/id", silent=True) is not None else None
        ref, alt = src.alleles()
    o = out[keep]
    ans: Dict[str, Any] = {"id": sub(vid), "chr": sub(list(chrom)), "pos": sub(posn)}
    if rs is not None:
        ans["rs.id"] = sub(list(rs))
    ans["ref"], ans["alt"] = sub(list(ref)), sub(list(alt))
    ans["AF.alt"], ans["mac"] = o[:, 0].copy(), o[:, 1].copy()
    ans["num"] = o[:, 2].astype(np.int32)
    ans["beta"], ans["SE"], ans["pval"] = o[:, 3].copy(), o[:, 4].copy(), o[:, 5].copy()
    if trait_type == "binary":
        ans["p.norm"] = o[:, 6].copy()
        ans["converged"] = o[:, 7] == 1
    return ans
