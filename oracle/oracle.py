"""ctypes binding of oracle/liboracle.so (test infrastructure only)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_SO_LD = os.path.join(_HERE, "liboracle_ld.so")
_lib = None
_lib_ld = None


class OracleTrace(C.Structure):
    _fields_ = [(n, C.c_long) for n in (
        "sparse_path", "dense_path", "flipped", "spa_entered", "cutoff_exit", "spa_done",
        "cutoff_doubled", "root_inf", "bisect", "not_converged", "newton_iters")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build_oracle(force: bool = False) -> str:
    mt = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("saige_oracle.c", "grm_oracle.c", "saige_oracle.h"))
    for so in (_SO, _SO_LD):
        if force or not os.path.exists(so) or os.path.getmtime(so) < mt:
            subprocess.check_call(["make", "-C", _HERE, "-B", os.path.basename(so)],
                                  stdout=subprocess.DEVNULL)
    return _SO


def _load(long_double: bool = False):
    """liboracle.so, or its long-double-accumulator twin (same code, ORC_ACC)."""
    global _lib, _lib_ld
    if long_double:
        if _lib_ld is None:
            build_oracle()
            _lib_ld = _bind(C.CDLL(_SO_LD))
        return _lib_ld
    if _lib is None:
        build_oracle()
        _lib = _bind(C.CDLL(_SO))
    return _lib


def _bind(L):
    dp = C.POINTER(C.c_double)
    L.orc_model_new.restype = C.c_void_p
    L.orc_model_new.argtypes = [C.c_int, C.c_int, C.c_int] + [dp] * 11 + [C.c_double] * 5
    L.orc_model_free.argtypes = [C.c_void_p]
    for nm in ("orc_scan_f64", "orc_scan_u8"):
        f = getattr(L, nm)
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, dp, C.POINTER(C.c_uint8),
                      C.POINTER(OracleTrace)]
    L.orc_scan_2bit.restype = C.c_int
    L.orc_scan_2bit.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, dp,
                                C.POINTER(C.c_uint8), C.POINTER(OracleTrace)]
    for nm in ("orc_pchisq1_upper", "orc_qnorm"):
        getattr(L, nm).restype = C.c_double
        getattr(L, nm).argtypes = [C.c_double]
    L.orc_pnorm.restype = C.c_double
    L.orc_pnorm.argtypes = [C.c_double, C.c_int]
    L.orc_saddle_prob_fast.restype = C.c_double
    L.orc_saddle_prob_fast.argtypes = [
        C.c_double, C.c_double, C.c_double, C.c_size_t, dp, dp, C.c_size_t,
        C.POINTER(C.c_int), C.c_double, C.POINTER(C.c_int), dp, dp, C.POINTER(OracleTrace)]
    L.orc_grm_new.restype = C.c_void_p
    L.orc_grm_new.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_size_t]
    L.orc_grm_free.argtypes = [C.c_void_p]
    L.orc_grm_diag.argtypes = [C.c_void_p, dp]
    L.orc_grm_crossprod.argtypes = [C.c_void_p, dp, dp]
    L.orc_grm_pcg.restype = C.c_int
    L.orc_grm_pcg.argtypes = [C.c_void_p, dp, dp, dp, C.c_int, C.c_double, dp]
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def pchisq1_upper(x: float) -> float:
    return _load().orc_pchisq1_upper(float(x))


def pnorm(z: float, lower: bool = True) -> float:
    return _load().orc_pnorm(float(z), int(lower))


def qnorm(p: float) -> float:
    return _load().orc_qnorm(float(p))


def saddle_prob_fast(q, m1, var1, mu, g, nonzero_idx, cutoff=2.0):
    """Saddle_Prob_Fast on explicit vectors -> (pval, converged, trace dict)."""
    L = _load()
    mu = np.ascontiguousarray(mu, dtype=np.float64)
    g = np.ascontiguousarray(g, dtype=np.float64)
    idx = np.ascontiguousarray(nonzero_idx, dtype=np.int32)
    buf = np.empty(2 * max(1, idx.size), dtype=np.float64)
    conv = C.c_int(0)
    tr = OracleTrace()
    p = L.orc_saddle_prob_fast(float(q), float(m1), float(var1), mu.size, _dp(mu), _dp(g),
                               idx.size, idx.ctypes.data_as(C.POINTER(C.c_int)), float(cutoff),
                               C.byref(conv), _dp(buf), None, C.byref(tr))
    return p, bool(conv.value), tr.as_dict()


class Oracle:
    """One flattened model (fields of saigegds_amd.nullmod.ScanModel)."""

    def __init__(self, sm, long_double: bool = False):
        L = _load(long_double)
        self._L = L
        self.n, self.k = sm.n, sm.k
        keep = [np.ascontiguousarray(a, dtype=np.float64) for a in (
            sm.tau, sm.y, sm.mu, sm.y_mu, sm.mu2, sm.t_XXVX_inv, sm.XV,
            sm.t_XVX_inv_XV, sm.t_X, sm.XVX, sm.S_a)]
        self._h = L.orc_model_new(sm.n, sm.k, int(sm.quant), *[_dp(a) for a in keep],
                                  sm.var_ratio, sm.maf, sm.mac, sm.missing, sm.spa_pval)
        self.trace = OracleTrace()

    def close(self):
        if self._h:
            self._L.orc_model_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _out(self, m):
        return np.empty((m, 8), dtype=np.float64), np.zeros(m, dtype=np.uint8)

    def scan_2bit(self, packed: np.ndarray):
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        m, bpv = packed.shape
        assert bpv >= (self.n + 3) // 4
        out, valid = self._out(m)
        self._L.orc_scan_2bit(self._h, packed.ctypes.data, bpv, m, _dp(out),
                              valid.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(self.trace))
        return out, valid

    def scan_f64(self, dosage: np.ndarray):
        dosage = np.ascontiguousarray(dosage, dtype=np.float64)
        m, n = dosage.shape
        assert n == self.n
        out, valid = self._out(m)
        self._L.orc_scan_f64(self._h, dosage.ctypes.data, m, _dp(out),
                             valid.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(self.trace))
        return out, valid

    def scan_u8(self, dosage: np.ndarray):
        dosage = np.ascontiguousarray(dosage, dtype=np.uint8)
        m, n = dosage.shape
        assert n == self.n
        out, valid = self._out(m)
        self._L.orc_scan_u8(self._h, dosage.ctypes.data, m, _dp(out),
                            valid.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(self.trace))
        return out, valid


class OracleScanner(Oracle):
    """The scan oracle behind the interface of ``saigegds_amd._lib.Scanner`` that the aggregate
    driver uses (tests of its host logic without a GPU): ``burden_2bit`` collapses in numpy the
    way ``burden_collapse_kernel`` does and hands the rows to ``scan_f64``."""

    def burden_2bit(self, packed, row_ptr, var_idx, lut):
        from saigegds_amd.gds import unpack_dosage_2bit
        codes = unpack_dosage_2bit(np.ascontiguousarray(packed, dtype=np.uint8), self.n)
        rows = np.zeros((len(row_ptr) - 1, self.n))
        for r in range(len(row_ptr) - 1):
            for e in range(int(row_ptr[r]), int(row_ptr[r + 1])):
                rows[r] += np.asarray(lut[e])[codes[var_idx[e]]]
        return self.scan_f64(rows)

    def close(self):
        Oracle.close(self)


class GrmOracle:
    """Implicit GRM of the null-model fit, CPU restatement (grm_oracle.c)."""

    def __init__(self, packed: np.ndarray, n_samp: int):
        self._L = _load()
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        self.n, self.m = int(n_samp), packed.shape[0]
        self._h = self._L.orc_grm_new(packed.ctypes.data, packed.shape[1], self.n, self.m)

    def close(self):
        if self._h:
            self._L.orc_grm_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def diag(self):
        out = np.empty(self.n)
        self._L.orc_grm_diag(self._h, _dp(out))
        return out

    def crossprod(self, b):
        b = np.ascontiguousarray(b, dtype=np.float64)
        out = np.empty(self.n)
        self._L.orc_grm_crossprod(self._h, _dp(b), _dp(out))
        return out

    def pcg(self, w, tau, b, maxiter=500, tol=1e-5):
        w = np.ascontiguousarray(w, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        tau = np.ascontiguousarray(tau, dtype=np.float64)
        x = np.empty(self.n)
        it = self._L.orc_grm_pcg(self._h, _dp(w), _dp(tau), _dp(b), int(maxiter), float(tol), _dp(x))
        return x, int(it)
