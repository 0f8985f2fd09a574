"""Result-file writers of ``seqAssocGLMM_SPA(res.savefn=)``
(reference R/assoc_single.r:310-328): ``.rds`` (saveRDS of the data.frame) and
``.rda/.RData`` (``save(.res)``), R serialisation format version 2 (XDR).
The GDS ``SAIGE_OUTPUT`` container (R/assoc_single.r:243-284) is not written by
this build."""
from __future__ import annotations

import gzip
import lzma
import re
import struct
from typing import Any, Dict, List

import numpy as np

NA_INT = -2147483648


def _i(v: int) -> bytes:
    return struct.pack(">i", v)


def _charsxp(s) -> bytes:
    if s is None:
        return _i(9) + _i(-1)
    b = str(s).encode("utf-8")
    flag = 0x00040009 if all(c < 128 for c in b) else 0x00008009
    return _i(flag) + _i(len(b)) + b


def _strsxp(vals, attr: bytes = b"") -> bytes:
    return _i(16 | (0x200 if attr else 0)) + _i(len(vals)) + b"".join(_charsxp(v) for v in vals) + attr


def _vector(col) -> bytes:
    if isinstance(col, (list, tuple)):
        return _strsxp(col)
    a = np.asarray(col)
    if a.dtype.kind in "US":
        return _strsxp([str(v) for v in a])
    if a.dtype.kind == "b":
        return _i(10) + _i(a.size) + a.astype(">i4").tobytes()
    if a.dtype.kind in "iu":
        return _i(13) + _i(a.size) + a.astype(">i4").tobytes()
    if a.dtype.kind == "f":
        return _i(14) + _i(a.size) + a.astype(">f8").tobytes()
    raise TypeError(f"unsupported column dtype {a.dtype}")


def _sym(name: str) -> bytes:
    return _i(1) + _charsxp(name)


def _pairlist(items) -> bytes:
    out = b""
    for name, payload in items:
        out += _i(0x402) + _sym(name) + payload
    return out + _i(254)


def _data_frame(cols: Dict[str, Any]) -> bytes:
    names = list(cols)
    n = len(next(iter(cols.values()))) if cols else 0
    body = b"".join(_vector(cols[k]) for k in names)
    attr = _pairlist([
        ("names", _strsxp(names)),
        ("class", _strsxp(["data.frame"])),
        ("row.names", _i(13) + _i(2) + _i(NA_INT) + _i(-n)),
    ])
    return _i(19 | 0x100 | 0x200) + _i(len(names)) + body + attr


_HEADER = b"X\n" + _i(2) + _i(0x00040005) + _i(0x00020300)


def serialize_data_frame(cols: Dict[str, Any]) -> bytes:
    return _HEADER + _data_frame(cols)


def _compress(raw: bytes, res_compress: str) -> bytes:
    # cm <- switch(res.compress, LZMA="xz", LZMA_RA="xz", ZIP="gzip", ZIP_RA="gzip", TRUE)
    if res_compress in ("LZMA", "LZMA_RA"):
        return lzma.compress(raw, format=lzma.FORMAT_XZ, preset=6)
    return gzip.compress(raw)


def save_result(ans: Dict[str, Any], fn: str, res_compress: str = "LZMA", sample_id: List[str] = None):
    if re.search(r"\.gds$", fn, re.I):            # R/assoc_single.r:243-286
        from .gds_write import write_saige_output
        write_saige_output(fn, ans, sample_id or [], res_compress)
        return
    if re.search(r"\.(rda|RData)$", fn, re.I):
        raw = b"RDX2\n" + _HEADER + _pairlist([(".res", _data_frame(ans))])
    elif re.search(r"\.rds$", fn, re.I):
        raw = serialize_data_frame(ans)
    else:
        raise ValueError("Unknown format of the output file, and it should be RData, RDS or gds.")
    with open(fn, "wb") as f:
        f.write(_compress(raw, res_compress))


# ---------------------------------------------------------------------------
# model file of seqFitNullGLMM_SPA(model.savefn=)  (reference R/saige_main.r:630-643)


def _with_attr(payload: bytes, attrs) -> bytes:
    """Set the has-attribute flag of a serialised vector and append its pairlist."""
    if not attrs:
        return payload
    flags = struct.unpack(">i", payload[:4])[0] | 0x200
    return _i(flags) + payload[4:] + _pairlist(attrs)


def _num(a, names=None) -> bytes:
    a = np.asarray(a)
    attrs = []
    if a.ndim == 2:
        attrs.append(("dim", _i(13) + _i(2) + _i(a.shape[0]) + _i(a.shape[1])))
        a = a.T        # column-major
    if names is not None:
        attrs.append(("names", _strsxp(list(names))))
    return _with_attr(_vector(np.ascontiguousarray(a).ravel()), attrs)


def _rlist(items, rclass=None, data_frame_rows=None) -> bytes:
    names = [k for k, _ in items]
    attrs = [("names", _strsxp(names))]
    if data_frame_rows is not None:
        attrs.append(("row.names", _i(13) + _i(2) + _i(NA_INT) + _i(-int(data_frame_rows))))
    if rclass:
        attrs.append(("class", _strsxp([rclass])))
    flags = 19 | 0x200 | (0x100 if rclass else 0)
    return _i(flags) + _i(len(items)) + b"".join(p for _, p in items) + _pairlist(attrs)


def serialize_model(m) -> bytes:
    """The ``ClassSAIGE_NullModel`` list (man/seqFitNullGLMM_SPA.Rd:75-93)."""
    vr = m.var_ratio_table
    nk = _rlist([("y", _num(m.y)), ("mu", _num(m.mu_noK)), ("res", _num(m.res_noK)), ("V", _num(m.V)),
                 ("X1", _num(m.X1)), ("XV", _num(m.XV)), ("XXVX_inv", _num(m.XXVX_inv))], rclass="SA_NULL")
    vrl = _rlist([(k, _num(np.asarray(vr[k], dtype=np.float64 if k != "id" else np.int32)))
                  for k in ("id", "maf", "mac", "var1", "var2", "ratio")],
                 rclass="data.frame", data_frame_rows=len(vr["ratio"]))
    items = [
        ("coefficients", _num(m.coefficients, names=getattr(m, "coef_names", None))),
        ("tau", _num(m.tau, names=["Sigma_E", "Sigma_G"])),
        ("linear.predictors", _num(m.linear_predictors)),
        ("fitted.values", _num(m.fitted_values)),
        ("residuals", _num(m.residuals)),
        ("cov", _num(m.cov)),
        ("converged", _vector(np.array([bool(m.converged)]))),
        ("obj.noK", nk),
        ("var.ratio", vrl),
        ("trait.type", _strsxp([m.trait_type])),
        ("sample.id", _vector(list(m.sample_id))),
        ("variant.id", _vector(np.asarray(m.variant_id))),
    ]
    return _HEADER + _rlist(items, rclass="ClassSAIGE_NullModel")


def save_model(m, fn: str):
    if re.search(r"\.(rda|RData)$", fn, re.I):
        raw = b"RDX2\n" + _HEADER + _pairlist([(".glmm", serialize_model(m)[len(_HEADER):])])
    elif re.search(r"\.rds$", fn, re.I):
        raw = serialize_model(m)
    else:
        raise ValueError("Unknown format of the output file, and it should be RData or RDS.")
    with open(fn, "wb") as f:
        f.write(gzip.compress(raw))
