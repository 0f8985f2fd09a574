"""ctypes binding of libsaigehip.so (include/saigehip.h).

There is no CPU fallback: if the HIP library is missing or no MI355X is
visible, every compute entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SAIGEHIP_LIB") or os.path.join(_HERE, "libsaigehip.so")   # (the override: A/B runs of two builds)

EXPORTS = (
    "sgx_version", "sgx_last_error", "sgx_device_count", "sgx_init", "sgx_free",
    "sgx_set_thresholds", "sgx_score_layout", "sgx_scan_2bit", "sgx_scan_2bit_dev", "sgx_scan_u8", "sgx_scan_i32", "sgx_scan_f64", "sgx_host_alloc", "sgx_host_free", "sgx_burden_2bit", "sgx_geno_stats_2bit", "sgx_decode_dbit2",
    "sgx_block_bytes", "sgx_block_create", "sgx_block_create_ex", "sgx_block_free", "sgx_block_load_dev", "sgx_block_load", "sgx_block_variants", "sgx_scan_block",
    "sgx_sync", "sgx_get_stats", "sgx_get_stats_total", "sgx_row_stride", "sgx_synth_2bit_dev", "sgx_selftest", "sgx_set_option",
    "sgx_grm_init", "sgx_grm_init_dev", "sgx_grm_crossprod_dev", "sgx_grm_sync", "sgx_grm_free", "sgx_grm_diag", "sgx_grm_crossprod", "sgx_grm_pcg",
)


class SgxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libsaigehip error {code}: {msg}")
        self.code = code


class SgxModel(C.Structure):
    _fields_ = [
        ("n_samp", C.c_int32), ("n_coeff", C.c_int32), ("trait", C.c_int32), ("reserved", C.c_int32),
        ("tau", C.c_double * 2), ("var_ratio", C.c_double),
        ("maf", C.c_double), ("mac", C.c_double), ("missing", C.c_double), ("spa_pval", C.c_double),
        ("y", C.c_void_p), ("mu", C.c_void_p), ("y_mu", C.c_void_p), ("mu2", C.c_void_p),
        ("t_XXVX_inv", C.c_void_p), ("XV", C.c_void_p), ("t_XVX_inv_XV", C.c_void_p),
        ("XVX", C.c_void_p), ("t_X", C.c_void_p), ("S_a", C.c_void_p),
    ]


class SgxStats(C.Structure):
    _fields_ = [
        ("n_variants", C.c_uint64), ("n_valid", C.c_uint64), ("n_spa", C.c_uint64),
        ("n_spa_dense", C.c_uint64), ("n_spa_slow", C.c_uint64),
        ("ms_score", C.c_float), ("ms_spa", C.c_float), ("ms_total", C.c_float),
        ("score_launches", C.c_uint32), ("spa_launches", C.c_uint32), ("ms_kernel", C.c_float),
        ("ms_lists", C.c_float), ("three_plane", C.c_uint32), ("n_unlisted", C.c_uint32), ("n_guarded", C.c_uint32),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own libamdhip64.so.  Two HIP runtimes in one
    process cannot both open the GPU, so when torch is installed its copy is
    loaded first; libsaigehip.so (NEEDED libamdhip64.so.7) then binds to it by
    SONAME.  Without torch (e.g. under R) the system ROCm runtime is used."""
    if os.environ.get("SAIGEHIP_SYSTEM_HIP") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except OSError:
        pass


def load():
    """Load libsaigehip.so; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C saigegds_amd/csrc` "
            "(or __graft_entry__.build()).  There is no CPU fallback.")
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    vp, sz, dp = C.c_void_p, C.c_size_t, C.c_double
    L.sgx_version.restype = C.c_char_p
    L.sgx_last_error.restype = C.c_char_p
    L.sgx_device_count.restype = C.c_int
    L.sgx_selftest.restype = C.c_int
    L.sgx_selftest.argtypes = [C.c_int]
    L.sgx_init.restype = C.c_int
    L.sgx_init.argtypes = [C.POINTER(SgxModel), C.c_int, C.POINTER(vp)]
    L.sgx_free.restype = None
    L.sgx_free.argtypes = [vp]
    L.sgx_set_thresholds.restype = C.c_int
    L.sgx_set_thresholds.argtypes = [vp, dp, dp, dp, dp]
    L.sgx_score_layout.restype = C.c_int
    L.sgx_score_layout.argtypes = [vp, vp, C.c_int32, C.POINTER(C.c_int32)]
    L.sgx_scan_2bit.restype = C.c_int
    L.sgx_scan_2bit.argtypes = [vp, vp, sz, sz, vp, vp]
    L.sgx_scan_2bit_dev.restype = C.c_int
    L.sgx_scan_2bit_dev.argtypes = [vp, vp, sz, sz, vp, vp]
    L.sgx_block_bytes.restype = sz
    L.sgx_block_bytes.argtypes = [C.c_int32, sz]
    L.sgx_block_create.restype = C.c_int
    L.sgx_block_create.argtypes = [C.c_int32, sz, C.c_int, C.POINTER(vp)]
    L.sgx_block_create_ex.restype = C.c_int
    L.sgx_block_create_ex.argtypes = [C.c_int32, sz, C.c_int, C.c_longlong, C.POINTER(vp)]
    L.sgx_block_free.restype = None
    L.sgx_block_free.argtypes = [vp]
    L.sgx_block_load_dev.restype = C.c_int
    L.sgx_block_load_dev.argtypes = [vp, vp, vp, sz, sz]
    L.sgx_block_load.restype = C.c_int
    L.sgx_block_load.argtypes = [vp, vp, vp, sz, sz]
    L.sgx_block_variants.restype = sz
    L.sgx_block_variants.argtypes = [vp]
    L.sgx_scan_block.restype = C.c_int
    L.sgx_scan_block.argtypes = [vp, vp, vp, vp]
    L.sgx_scan_u8.restype = C.c_int
    L.sgx_scan_u8.argtypes = [vp, vp, sz, vp, vp]
    L.sgx_scan_f64.restype = C.c_int
    L.sgx_scan_f64.argtypes = [vp, vp, sz, vp, vp]
    L.sgx_host_alloc.restype = vp
    L.sgx_host_alloc.argtypes = [sz]
    L.sgx_host_free.restype = None
    L.sgx_host_free.argtypes = [vp]
    L.sgx_scan_i32.restype = C.c_int
    L.sgx_scan_i32.argtypes = [vp, vp, sz, vp, vp]
    L.sgx_burden_2bit.restype = C.c_int
    L.sgx_burden_2bit.argtypes = [vp, vp, sz, sz, sz, vp, vp, vp, vp, vp]
    L.sgx_decode_dbit2.restype = C.c_int
    L.sgx_decode_dbit2.argtypes = [vp, sz, C.c_int32, sz, vp, C.c_int32, vp, sz, C.c_int]
    L.sgx_geno_stats_2bit.restype = C.c_int
    L.sgx_geno_stats_2bit.argtypes = [vp, sz, C.c_int32, sz, C.c_int, vp, vp]
    L.sgx_set_option.restype = C.c_int
    L.sgx_set_option.argtypes = [vp, C.c_char_p, C.c_longlong]
    L.sgx_sync.restype = C.c_int
    L.sgx_sync.argtypes = [vp]
    L.sgx_get_stats.restype = C.c_int
    L.sgx_get_stats.argtypes = [vp, C.POINTER(SgxStats)]
    L.sgx_get_stats_total.restype = C.c_int
    L.sgx_get_stats_total.argtypes = [vp, C.POINTER(SgxStats), C.POINTER(C.c_uint64), C.c_int]
    L.sgx_row_stride.restype = sz
    L.sgx_row_stride.argtypes = [C.c_int32]
    L.sgx_synth_2bit_dev.restype = C.c_int
    L.sgx_synth_2bit_dev.argtypes = [vp, vp, sz, C.c_int32, sz, C.c_uint64, C.c_uint64, vp]
    L.sgx_grm_init.restype = C.c_int
    L.sgx_grm_init.argtypes = [vp, sz, C.c_int32, sz, C.c_int, C.POINTER(vp)]
    L.sgx_grm_init_dev.restype = C.c_int
    L.sgx_grm_init_dev.argtypes = [vp, sz, C.c_int32, sz, C.c_int, C.POINTER(vp)]
    L.sgx_grm_crossprod_dev.restype = C.c_int
    L.sgx_grm_crossprod_dev.argtypes = [vp, vp, vp]
    L.sgx_grm_sync.restype = C.c_int
    L.sgx_grm_sync.argtypes = [vp]
    L.sgx_grm_free.restype = None
    L.sgx_grm_free.argtypes = [vp]
    L.sgx_grm_diag.restype = C.c_int
    L.sgx_grm_diag.argtypes = [vp, vp]
    L.sgx_grm_crossprod.restype = C.c_int
    L.sgx_grm_crossprod.argtypes = [vp, vp, vp]
    L.sgx_grm_pcg.restype = C.c_int
    L.sgx_grm_pcg.argtypes = [vp, vp, vp, vp, C.c_int, dp, vp, C.POINTER(C.c_int)]
    _lib = L
    return L


def check(rc: int):
    if rc != 0:
        raise SgxError(rc, load().sgx_last_error().decode("utf-8", "replace"))


class Scanner:
    """One model resident on one GPU (an ``sgx_handle``)."""

    def __init__(self, sm, device: int = 0):
        L = load()
        self._L = L
        self.n, self.k, self.quant = sm.n, sm.k, sm.quant
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
        self._keep = dict(y=f(sm.y), mu=f(sm.mu), y_mu=f(sm.y_mu), mu2=f(sm.mu2),
                          t_XXVX_inv=f(sm.t_XXVX_inv), XV=f(sm.XV), t_XVX_inv_XV=f(sm.t_XVX_inv_XV),
                          XVX=f(sm.XVX), t_X=f(sm.t_X), S_a=f(sm.S_a))
        m = SgxModel()
        m.n_samp, m.n_coeff = sm.n, sm.k
        m.trait = 1 if sm.quant else 0
        m.tau[0], m.tau[1] = float(sm.tau[0]), float(sm.tau[1])
        m.var_ratio = sm.var_ratio
        m.maf, m.mac, m.missing, m.spa_pval = sm.maf, sm.mac, sm.missing, sm.spa_pval
        for k, a in self._keep.items():
            setattr(m, k, a.ctypes.data)
        h = C.c_void_p()
        check(L.sgx_init(C.byref(m), int(device), C.byref(h)))
        self._h = h
        self.device = device

    # -- lifetime --------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.sgx_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- host-buffer scans -------------------------------------------------
    def _out(self, m):
        return np.empty((m, 8), dtype=np.float64), np.zeros(m, dtype=np.uint8)

    def scan_2bit(self, packed: np.ndarray):
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        if packed.ndim != 2:
            raise ValueError("packed genotypes must be [n_variants, bytes_per_variant]")
        m, bpv = packed.shape
        out, valid = self._out(m)
        check(self._L.sgx_scan_2bit(self._h, packed.ctypes.data, bpv, m, out.ctypes.data,
                                    valid.ctypes.data))
        return out, valid

    def scan_u8(self, dosage: np.ndarray):
        dosage = np.ascontiguousarray(dosage, dtype=np.uint8)
        if dosage.ndim != 2 or dosage.shape[1] != self.n:
            raise ValueError(f"Invalid length of dosages: {dosage.shape[-1]}.")
        out, valid = self._out(dosage.shape[0])
        check(self._L.sgx_scan_u8(self._h, dosage.ctypes.data, dosage.shape[0], out.ctypes.data,
                                  valid.ctypes.data))
        return out, valid

    def scan_i32(self, dosage: np.ndarray):
        """INTEGER dosages, NA_INTEGER (-2147483648) = missing (the INTSXP branch of get_ds)."""
        dosage = np.ascontiguousarray(dosage, dtype=np.int32)
        if dosage.ndim != 2 or dosage.shape[1] != self.n:
            raise ValueError(f"Invalid length of dosages: {dosage.shape[-1]}.")
        out, valid = self._out(dosage.shape[0])
        check(self._L.sgx_scan_i32(self._h, dosage.ctypes.data, dosage.shape[0], out.ctypes.data,
                                   valid.ctypes.data))
        return out, valid

    def scan_f64(self, dosage: np.ndarray):
        dosage = np.ascontiguousarray(dosage, dtype=np.float64)
        if dosage.ndim != 2 or dosage.shape[1] != self.n:
            raise ValueError(f"Invalid length of dosages: {dosage.shape[-1]}.")
        out, valid = self._out(dosage.shape[0])
        check(self._L.sgx_scan_f64(self._h, dosage.ctypes.data, dosage.shape[0], out.ctypes.data,
                                   valid.ctypes.data))
        return out, valid

    def burden_2bit(self, packed: np.ndarray, row_ptr, var_idx, lut):
        """Burden rows (CSR over the rows of ``packed``, one 4-entry table per entry), then the
        single-variant test on each row (``sgx_burden_2bit``)."""
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
        var_idx = np.ascontiguousarray(var_idx, dtype=np.int32)
        lut = np.ascontiguousarray(lut, dtype=np.float64)
        n_rows = row_ptr.size - 1
        if packed.ndim != 2 or lut.shape != (var_idx.size, 4) or n_rows < 0 or (n_rows >= 0 and row_ptr[-1] != var_idx.size):
            raise ValueError("burden_2bit: inconsistent CSR / table shapes")
        out, valid = self._out(n_rows)
        check(self._L.sgx_burden_2bit(self._h, packed.ctypes.data, packed.shape[1], packed.shape[0], n_rows,
                                      row_ptr.ctypes.data, var_idx.ctypes.data, lut.ctypes.data,
                                      out.ctypes.data, valid.ctypes.data))
        return out, valid

    # -- device-resident scans (pointers are raw device addresses) ---------
    def row_stride(self) -> int:
        return int(self._L.sgx_row_stride(self.n))

    def scan_2bit_dev(self, packed_ptr: int, bpv: int, m: int, out_ptr: int, valid_ptr: int):
        check(self._L.sgx_scan_2bit_dev(self._h, packed_ptr, bpv, m, out_ptr, valid_ptr))

    # -- genotype blocks (rows + their sparse side resident on the device; Block below) ---
    def load_block_dev(self, block: "Block", packed_ptr: int, bpv: int, m: int):
        """Rows already in this GPU's memory -> block (asynchronous on the handle's stream)."""
        check(self._L.sgx_block_load_dev(self._h, block._b, packed_ptr, bpv, m))

    def load_block(self, block: "Block", packed: np.ndarray):
        """Rows in host memory -> block, through the pinned pipeline."""
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        if packed.ndim != 2:
            raise ValueError("packed genotypes must be [n_variants, bytes_per_variant]")
        check(self._L.sgx_block_load(self._h, block._b, packed.ctypes.data, packed.shape[1], packed.shape[0]))

    def scan_block(self, block: "Block", out_ptr: int, valid_ptr: int):
        """Scan of a loaded block into device buffers (asynchronous; sync() before reading)."""
        check(self._L.sgx_scan_block(self._h, block._b, out_ptr, valid_ptr))

    def synth_2bit_dev(self, packed_ptr: int, bpv: int, m: int, first_variant: int, seed: int,
                       thr_ptr: int):
        check(self._L.sgx_synth_2bit_dev(self._h, packed_ptr, bpv, self.n, m, first_variant, seed,
                                         thr_ptr))

    def sync(self):
        check(self._L.sgx_sync(self._h))

    def stats(self) -> dict:
        st = SgxStats()
        check(self._L.sgx_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    def stats_total(self, reset: bool = True):
        """Sums over the calls completed since the last reset -> (dict, n_calls); syncs."""
        st, n = SgxStats(), C.c_uint64(0)
        check(self._L.sgx_get_stats_total(self._h, C.byref(st), C.byref(n), 1 if reset else 0))
        return st.as_dict(), int(n.value)

    def set_option(self, name: str, value: int):
        check(self._L.sgx_set_option(self._h, name.encode(), int(value)))

    def score_layout(self):
        """(limb counts of the score columns c'(K), e(K), s, w; number of column groups, 0 = FP64 path)."""
        p = 2 * self.k + 2
        limbs = np.zeros(p, dtype=np.int32)
        ng = C.c_int32(0)
        check(self._L.sgx_score_layout(self._h, limbs.ctypes.data, p, C.byref(ng)))
        return limbs, int(ng.value)

    def set_thresholds(self, maf, mac, missing, spa_pval):
        check(self._L.sgx_set_thresholds(self._h, maf, mac, missing, spa_pval))


class Block:
    """A block of variants resident on the device (``sgx_block``: the 2-bit rows, the positions of their missing
    genotypes, the carrier lists of the rare variants): depends on the number of samples only, so one loaded
    block can be scanned with any number of models.  ``clist_avg`` (test hook): room of the carrier lists in
    entries per variant."""

    def __init__(self, n_samp: int, max_variants: int, device: int = 0, clist_avg: int = None):
        self._L = load()
        b = C.c_void_p()
        if clist_avg is None:
            check(self._L.sgx_block_create(int(n_samp), int(max_variants), int(device), C.byref(b)))
        else:
            check(self._L.sgx_block_create_ex(int(n_samp), int(max_variants), int(device), int(clist_avg), C.byref(b)))
        self._b = b
        self.n, self.cap = int(n_samp), int(max_variants)

    @staticmethod
    def nbytes(n_samp: int, max_variants: int) -> int:
        return int(load().sgx_block_bytes(int(n_samp), int(max_variants)))

    @property
    def n_variants(self) -> int:
        return int(self._L.sgx_block_variants(self._b))

    def close(self):
        if getattr(self, "_b", None):
            self._L.sgx_block_free(self._b)
            self._b = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class PinnedBuffer:
    """Page-locked host memory (``sgx_host_alloc``) as a numpy array: block buffers filled by the GDS
    decoder and handed to the host-buffer scans cross PCIe at the full link rate."""

    def __init__(self, shape, dtype=np.uint8):
        self._L = load()
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._p = self._L.sgx_host_alloc(max(1, self.nbytes))
        if not self._p:
            raise MemoryError(self._L.sgx_last_error().decode())
        buf = (C.c_uint8 * max(1, self.nbytes)).from_address(self._p)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def close(self):
        if self._p:
            self.array = None
            self._L.sgx_host_free(self._p)
            self._p = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def decode_dbit2(alleles, bit0: int, n_samp: int, m: int, out: np.ndarray, sel: np.ndarray = None, threads: int = 0):
    """SeqArray allele codes (bytes of genotype/data from bit ``bit0`` on) -> 2-bit dosage rows in ``out``
    ([>= m, >= ceil(n / 4)] uint8, C-contiguous rows); ``sgx_decode_dbit2``, host threads, no GPU needed."""
    L = load()
    a = np.frombuffer(alleles, dtype=np.uint8) if not isinstance(alleles, np.ndarray) else alleles
    if out.dtype != np.uint8 or out.ndim != 2 or out.strides[1] != 1 or out.shape[0] < m:
        raise ValueError("decode_dbit2: out must be a [>= m, stride] uint8 array with contiguous rows")
    if out.strides[0] != out.shape[1]:
        # (the library zeroes a row's bytes from the last code up to its stride: a column slice of a wider buffer
        # would have the neighbouring columns cleared, and the last row would be written past the allocation)
        raise ValueError("decode_dbit2: out must own whole rows (a column slice of a wider buffer is not accepted)")
    need = (bit0 + m * n_samp * 4 + 7) // 8
    if a.size < need:
        raise ValueError("decode_dbit2: allele buffer too short")
    sp, ns = None, 0
    if sel is not None:
        sel = np.ascontiguousarray(sel, dtype=np.int64)
        sp, ns = sel.ctypes.data, int(sel.size)
    check(L.sgx_decode_dbit2(a.ctypes.data, int(bit0), int(n_samp), int(m), sp, ns, out.ctypes.data, int(out.strides[0]), int(threads)))
    return out


def geno_stats_2bit(packed: np.ndarray, n_samp: int, device: int = 0):
    """Per-variant (n_valid, allele_sum) of a host 2-bit matrix, counted on the GPU."""
    L = load()
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    m = packed.shape[0]
    nv, sm = np.empty(m, dtype=np.int32), np.empty(m, dtype=np.int32)
    check(L.sgx_geno_stats_2bit(packed.ctypes.data, packed.shape[1], int(n_samp), m, int(device),
                                nv.ctypes.data, sm.ctypes.data))
    return nv, sm


class GrmOperator:
    """Implicit GRM of the null-model fit on one GPU (``sgx_grm``): the operator
    behind ``saige_store_2b_geno`` / ``get_crossprod_b_grm`` / ``PCG_diag_sigma``
    (reference src/saige_fitnull.cpp:159-230, 435-536, 581-614)."""

    def __init__(self, packed, n_samp: int, device: int = 0, dev_ptr: int = 0, n_markers: int = 0,
                 bytes_per_marker: int = 0):
        """packed: numpy [n_markers, bytes_per_marker]; or dev_ptr/n_markers/bytes_per_marker
        for a matrix already in this GPU's memory."""
        L = load()
        self._L = L
        h = C.c_void_p()
        self.n = int(n_samp)
        if dev_ptr:
            self.m = int(n_markers)
            check(L.sgx_grm_init_dev(dev_ptr, int(bytes_per_marker), self.n, self.m, int(device), C.byref(h)))
        else:
            packed = np.ascontiguousarray(packed, dtype=np.uint8)
            if packed.ndim != 2:
                raise ValueError("packed genotypes must be [n_markers, bytes_per_marker]")
            self.m = int(packed.shape[0])
            check(L.sgx_grm_init(packed.ctypes.data, packed.shape[1], self.n, self.m, int(device), C.byref(h)))
        self._h = h

    def crossprod_dev(self, b_ptr: int, out_ptr: int):
        check(self._L.sgx_grm_crossprod_dev(self._h, b_ptr, out_ptr))

    def sync(self):
        check(self._L.sgx_grm_sync(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._L.sgx_grm_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def diag(self) -> np.ndarray:
        out = np.empty(self.n, dtype=np.float64)
        check(self._L.sgx_grm_diag(self._h, out.ctypes.data))
        return out

    def crossprod(self, b: np.ndarray) -> np.ndarray:
        b = np.ascontiguousarray(b, dtype=np.float64)
        if b.shape != (self.n,):
            raise ValueError("b must have one entry per sample")
        out = np.empty(self.n, dtype=np.float64)
        check(self._L.sgx_grm_crossprod(self._h, b.ctypes.data, out.ctypes.data))
        return out

    def pcg(self, w: np.ndarray, tau, b: np.ndarray, maxiter: int = 500, tol: float = 1e-5):
        w = np.ascontiguousarray(w, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        tau = np.ascontiguousarray(tau, dtype=np.float64)
        x = np.empty(self.n, dtype=np.float64)
        it = C.c_int(0)
        check(self._L.sgx_grm_pcg(self._h, w.ctypes.data, tau.ctypes.data, b.ctypes.data, int(maxiter),
                                  float(tol), x.ctypes.data, C.byref(it)))
        return x, int(it.value)
