"""Minimal reader for R's RDS serialisation (XDR, format version 2/3).

The scan consumes a ``ClassSAIGE_NullModel`` object written by
``seqFitNullGLMM_SPA(..., model.savefn="*.rds")`` (reference
R/saige_main.r:630-643) and the reference's golden result tables
(inst/unitTests/saige_pval*.rds).  There is no R on the GPU box, so the host
side decodes the container itself.  Only the node types those files contain are
supported; anything else raises ``RdsError``.

R objects map to Python as
    NULL -> None, logical/integer/real vector -> numpy array (NA_integer_ and
    NA_logical_ -> masked by ``RObj.na``), character vector -> list[str|None],
    list -> RList (ordered, name lookup), pairlist -> RList,
    symbol -> RSym, language objects -> RLang, environments -> REnv (opaque).
Attributes (names, dim, class, ...) live in ``.attr`` of the wrapper.
"""
from __future__ import annotations

import bz2
import gzip
import lzma
import struct
from typing import Any, Dict, List, Optional

import numpy as np

NA_INTEGER = -2147483648


class RdsError(ValueError):
    pass


class RSym(str):
    """An R symbol."""


class REnv:
    """An R environment; contents are parsed and dropped."""

    def __init__(self, kind: str = "env"):
        self.kind = kind

    def __repr__(self):
        return f"<REnv {self.kind}>"


class RLang:
    def __init__(self, items, attr=None):
        self.items = items
        self.attr = attr or {}

    def __repr__(self):
        return f"<RLang {self.items!r}>"


class RList:
    """Ordered R list with optional names (``x$name`` -> ``x["name"]``)."""

    def __init__(self, values: List[Any], names: Optional[List[Optional[str]]] = None,
                 attr: Optional[Dict[str, Any]] = None):
        self.values = values
        self.names = names
        self.attr = attr or {}

    def __len__(self):
        return len(self.values)

    def __iter__(self):
        return iter(self.values)

    def keys(self):
        return list(self.names or [])

    def __contains__(self, key):
        return self.names is not None and key in self.names

    def __getitem__(self, key):
        if isinstance(key, str):
            if self.names is None or key not in self.names:
                raise KeyError(key)
            return self.values[self.names.index(key)]
        return self.values[key]

    def get(self, key, default=None):
        try:
            return self[key]
        except (KeyError, IndexError):
            return default

    def rclass(self):
        c = self.attr.get("class")
        return list(c) if c is not None else []

    def __repr__(self):
        return f"<RList n={len(self.values)} names={self.names}>"


class RArray(np.ndarray):
    """numpy array carrying R attributes (dim already applied, column-major)."""

    def __new__(cls, arr, attr=None):
        obj = np.asarray(arr).view(cls)
        obj.attr = attr or {}
        return obj

    def __array_finalize__(self, obj):
        self.attr = getattr(obj, "attr", {})


def _decompress(raw: bytes) -> bytes:
    if raw[:6] == b"\xfd7zXZ\x00":
        return lzma.decompress(raw)
    if raw[:2] == b"\x1f\x8b":
        return gzip.decompress(raw)
    if raw[:3] == b"BZh":
        return bz2.decompress(raw)
    return raw


class _Reader:
    def __init__(self, buf: bytes):
        self.b = buf
        self.p = 0
        self.refs: List[Any] = []

    def int(self) -> int:
        v = struct.unpack_from(">i", self.b, self.p)[0]
        self.p += 4
        return v

    def length(self) -> int:
        n = self.int()
        if n == -1:
            hi, lo = self.int(), self.int()
            n = (hi << 32) + (lo & 0xFFFFFFFF)
        return n

    def bytes(self, n: int) -> bytes:
        v = self.b[self.p:self.p + n]
        if len(v) != n:
            raise RdsError("truncated RDS stream")
        self.p += n
        return v

    # -- items ---------------------------------------------------------
    def item(self) -> Any:
        flags = self.int()
        t = flags & 0xFF
        has_attr = bool(flags & 0x200)
        has_tag = bool(flags & 0x400)

        if t == 254:  # NILVALUE_SXP
            return None
        if t == 253:
            return REnv("global")
        if t == 242:
            return REnv("empty")
        if t == 241:
            return REnv("base")
        if t == 248:
            return REnv("basenamespace")
        if t in (251, 252):  # missing arg / unbound value
            return None
        if t == 255:  # REFSXP
            idx = flags >> 8
            if idx == 0:
                idx = self.int()
            return self.refs[idx - 1]
        if t in (249, 250, 247):  # NAMESPACESXP / PACKAGESXP / PERSISTSXP
            self.int()  # always 0
            n = self.int()
            s = [self.item() for _ in range(n)]
            e = REnv("namespace:" + str(s[0] if s else ""))
            self.refs.append(e)
            return e
        if t == 1:  # SYMSXP
            s = RSym(self.item() or "")
            self.refs.append(s)
            return s
        if t == 4:  # ENVSXP
            e = REnv()
            self.refs.append(e)
            self.int()  # locked
            self.item()  # enclos
            self.item()  # frame
            self.item()  # hashtab
            self.item()  # attrib
            return e
        if t in (2, 6, 3, 5, 17, 239, 240):  # pairlist-like
            return self.pairlist(t, has_attr, has_tag)
        if t == 9:  # CHARSXP
            n = self.int()
            if n == -1:
                return None
            return self.bytes(n).decode("utf-8", errors="replace")
        if t in (10, 13):  # LGLSXP / INTSXP
            n = self.length()
            a = np.frombuffer(self.bytes(4 * n), dtype=">i4").astype(np.int32)
            return self.finish_vector(a, has_attr, logical=(t == 10))
        if t == 14:  # REALSXP
            n = self.length()
            a = np.frombuffer(self.bytes(8 * n), dtype=">f8").astype(np.float64)
            return self.finish_vector(a, has_attr)
        if t == 15:  # CPLXSXP
            n = self.length()
            a = np.frombuffer(self.bytes(16 * n), dtype=">c16").astype(np.complex128)
            return self.finish_vector(a, has_attr)
        if t == 24:  # RAWSXP
            n = self.length()
            a = np.frombuffer(self.bytes(n), dtype=np.uint8).copy()
            return self.finish_vector(a, has_attr)
        if t == 16:  # STRSXP
            n = self.length()
            vals = [self.item() for _ in range(n)]
            attr = self.attributes() if has_attr else {}
            if attr:
                return RList(vals, _names_of(attr), attr) if False else _StrVec(vals, attr)
            return vals
        if t in (19, 20):  # VECSXP / EXPRSXP
            n = self.length()
            vals = [self.item() for _ in range(n)]
            attr = self.attributes() if has_attr else {}
            return RList(vals, _names_of(attr), attr)
        if t == 238:  # ALTREP_SXP
            info = self.item()
            state = self.item()
            attr_pl = self.item()
            attr = _pairlist_to_attr(attr_pl)
            return self.altrep(info, state, attr)
        if t == 25:  # S4SXP
            attr = self.attributes() if has_attr else {}
            return RList([], None, attr)
        if t == 22:  # EXTPTRSXP
            e = REnv("extptr")
            self.refs.append(e)
            self.item()
            self.item()
            if has_attr:
                self.attributes()
            return e
        if t == 8 or t == 7:  # BUILTINSXP / SPECIALSXP
            n = self.int()
            return RSym(self.bytes(n).decode())
        raise RdsError(f"unsupported SEXP type {t} at offset {self.p - 4}")

    def attributes(self) -> Dict[str, Any]:
        return _pairlist_to_attr(self.item())

    def pairlist(self, t, has_attr, has_tag):
        vals, names = [], []
        attr = {}
        first = True
        while True:
            if not first:
                flags = self.int()
                t2 = flags & 0xFF
                if t2 == 254:
                    break
                if t2 not in (2, 6, 3, 5, 17, 239, 240):
                    # dotted tail (CDR is not a pairlist): rewind and read as item
                    self.p -= 4
                    vals.append(self.item())
                    names.append(None)
                    break
                has_attr = bool(flags & 0x200)
                has_tag = bool(flags & 0x400)
                t_cur = t2
            else:
                t_cur = t
            first = False
            if has_attr or t_cur in (239, 240):
                a = self.attributes()
                if not attr:
                    attr = a
            tag = self.item() if has_tag else None
            vals.append(self.item())
            names.append(str(tag) if tag is not None else None)
        if t in (6, 240):
            return RLang(vals, attr)
        nm = names if any(n is not None for n in names) else None
        return RList(vals, nm, attr)

    def finish_vector(self, a, has_attr, logical=False):
        attr = self.attributes() if has_attr else {}
        if logical:
            na = a == NA_INTEGER
            out = RArray(a.astype(bool), attr)
            attr["__na__"] = na if na.any() else None
        else:
            out = RArray(a, attr)
        dim = attr.get("dim")
        if dim is not None and int(np.prod(dim)) == out.size:
            shp = tuple(int(d) for d in np.asarray(dim))
            out = RArray(np.asarray(out).reshape(shp, order="F"), attr)
        return out

    def altrep(self, info, state, attr):
        cls = str(info[0]) if len(info) else ""
        if cls in ("wrap_real", "wrap_integer", "wrap_logical", "wrap_string",
                   "wrap_complex", "wrap_raw", "wrap_list"):
            x = state[0]
            return _with_attr(x, attr)
        if cls in ("compact_intseq", "compact_realseq"):
            n, start, step = (float(v) for v in np.asarray(state)[:3])
            dt = np.int32 if cls == "compact_intseq" else np.float64
            a = (start + step * np.arange(int(n))).astype(dt)
            return _with_attr(RArray(a), attr)
        if cls == "deferred_string":
            arg = state[0]
            vals = np.asarray(arg)
            if vals.dtype.kind == "i":
                out = [None if v == NA_INTEGER else str(int(v)) for v in vals]
            else:
                out = [None if np.isnan(v) else _r_num_to_str(float(v)) for v in vals]
            return _with_attr(out, attr)
        raise RdsError(f"unsupported ALTREP class {cls!r}")


class _StrVec(list):
    """character vector with attributes"""

    def __init__(self, vals, attr):
        super().__init__(vals)
        self.attr = attr


def _r_num_to_str(v: float) -> str:
    if v == int(v) and abs(v) < 1e15:
        return str(int(v))
    return repr(v)


def _with_attr(x, attr):
    if not attr:
        return x
    if isinstance(x, np.ndarray):
        a = dict(getattr(x, "attr", {}))
        a.update(attr)
        out = RArray(np.asarray(x), a)
        dim = a.get("dim")
        if dim is not None and out.ndim == 1 and int(np.prod(dim)) == out.size:
            out = RArray(np.asarray(out).reshape(tuple(int(d) for d in dim), order="F"), a)
        return out
    if isinstance(x, RList):
        a = dict(x.attr)
        a.update(attr)
        return RList(x.values, _names_of(a) or x.names, a)
    if isinstance(x, list):
        return _StrVec(x, attr)
    return x


def _names_of(attr):
    nm = attr.get("names") if attr else None
    if nm is None:
        return None
    return [n for n in nm]


def _pairlist_to_attr(pl) -> Dict[str, Any]:
    if pl is None:
        return {}
    if isinstance(pl, RList):
        out = {}
        for k, v in zip(pl.names or [None] * len(pl.values), pl.values):
            if k is not None:
                out[k] = v
        return out
    return {}


def read_rds(path: str) -> Any:
    """Parse an ``.rds`` file (xz/gzip/bzip2/uncompressed XDR)."""
    with open(path, "rb") as f:
        raw = f.read()
    return loads(_decompress(raw))


def read_rdata(path: str) -> Dict[str, Any]:
    """Parse an ``.rda`` / ``.RData`` file written by ``save()``: the magic line "RDX2" / "RDX3", then one
    serialised pairlist whose tags are the object names -> {name: object} in file order."""
    with open(path, "rb") as f:
        raw = _decompress(f.read())
    if raw[:5] not in (b"RDX2\n", b"RDX3\n"):
        raise RdsError("not an RData file (header %r): only the XDR form written by save() is supported" % raw[:5])
    pl = loads(raw[5:])
    if isinstance(pl, RList):
        return {str(k): v for k, v in zip(pl.names or [], pl.values)}
    if isinstance(pl, dict):
        return {str(k): v for k, v in pl.items()}
    raise RdsError("RData: the top-level object is not a pairlist")


def loads(buf: bytes) -> Any:
    if buf[:2] != b"X\n":
        raise RdsError("only XDR-format RDS is supported (header %r)" % buf[:2])
    r = _Reader(buf)
    r.p = 2
    version = r.int()
    r.int()  # writer version
    r.int()  # min reader version
    if version == 3:
        n = r.int()
        r.bytes(n)  # native encoding
    elif version != 2:
        raise RdsError(f"unsupported RDS version {version}")
    return r.item()


def data_frame_columns(df: RList) -> Dict[str, Any]:
    """Columns of an R data.frame as {name: array|list}."""
    return {k: v for k, v in zip(df.names, df.values)}
