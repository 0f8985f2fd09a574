"""Narrow reader for CoreArray GDS containers written by SeqArray/gdsfmt.

The reference reads genotypes through SeqArray (``seqOpen``/``seqGetData``/
``seqApply``; R/assoc_single.r:116-144,202-221), a Bioconductor dependency
that is not part of the reference tree.  This module decodes just what the
single-variant scan consumes:

* the directory tree (folder streams),
* array nodes of class dBit2 (``genotype/data``), dInt32, dStr8/dStr16,
  dFloat32/64, dPackedReal8[U] (``annotation/format/DS/data``),
* stored uncompressed or as ``LZMA_RA`` (independent XZ blocks).

Container layout (decoded from the files themselves):
  file   := "COREARRAYx0A" u8[2]version u32 root_id  block*
  block  := u48 size (bit47 = head) u48 next  [u32 id u48 stream_size]  payload
Property records inside node streams are matched by their byte signatures.
Anything outside this subset raises ``GdsError``.
"""
from __future__ import annotations

import lzma
import re
import struct
from typing import Dict, List, Optional, Tuple

import numpy as np


class GdsError(ValueError):
    pass


_MAGIC = b"COREARRAYx0A"
_XZ = b"\xfd7zXZ\x00"

# property signatures (type byte + packed property name)
_P_ENTRY_ID = b"\t\x02\xf5\x00"
_P_ENTRY_FLAG = b"\tDt1\x12"
_P_ENTRY_NAME = b"\x15D\xc6`\x10"
_P_PIPE = b"\x15\xc4Fm\x10"
_P_PIPE_SIZE = b"\x02\xc9FmP\xe0Q%\x04\x10"
_P_DCNT = b"\x07\xc4\xe3d\x1f"
_P_DIM = b"\x02\xc3Ca"
_P_DATA = b"\t\xc4\xc3|\x0c"
_P_OFFSET = b"\x13\x86\x16E\x1e\xf4\x01"
_P_SCALE = b"\x13\x85\xe70\x17\x04"


def _u48(b: bytes, o: int) -> int:
    return int.from_bytes(b[o:o + 6], "little")


def _unpack7(b: bytes, n: int) -> str:
    """n characters packed 7 bits each, LSB first."""
    v = int.from_bytes(b, "little")
    return "".join(chr((v >> (7 * i)) & 0x7F) for i in range(n))


class GdsNode:
    def __init__(self, name: str, block_id: int, is_folder: bool):
        self.name = name
        self.block_id = block_id
        self.is_folder = is_folder
        self.children: Dict[str, "GdsNode"] = {}
        # array properties
        self.cls: Optional[str] = None
        self.dims: Optional[Tuple[int, ...]] = None
        self.data_id: Optional[int] = None
        self.pipe: str = ""
        self.raw_size: Optional[int] = None
        self.offset = 0.0
        self.scale = 1.0


class GdsFile:
    """Read-only view of a GDS file.  The file is memory-mapped; array nodes can be read whole
    (``read``) or by row ranges (``read_rows``, ``dosage_alt_packed_range``), which decode only
    the LZMA_RA blocks that hold the range -- the scan never has the genotype node in RAM."""

    def __init__(self, path: str):
        import mmap
        self.path = path
        self._fh = open(path, "rb")
        import threading
        self.buf = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
        self._tls = threading.local()
        if self.buf[:len(_MAGIC)] != _MAGIC:
            raise GdsError(f"{path}: not a CoreArray GDS file")
        self.root_id = struct.unpack_from("<I", self.buf, 14)[0]
        self._streams: Dict[int, List[Tuple[int, int]]] = {}
        self._walk_blocks()
        self.root = GdsNode("", self.root_id, True)
        self._load_folder(self.root)

    # ------------------------------------------------------------------
    def _walk_blocks(self):
        b, pos, end = self.buf, 18, len(self.buf)
        heads: Dict[int, Tuple[int, int, int]] = {}
        cont: Dict[int, Tuple[int, int]] = {}
        while pos < end:
            sz = _u48(b, pos)
            head = bool(sz >> 47)
            sz &= (1 << 47) - 1
            if sz < 12:
                raise GdsError("corrupt block list")
            nxt = _u48(b, pos + 6)
            if head:
                bid = struct.unpack_from("<I", b, pos + 12)[0]
                heads[bid] = (pos + 22, sz - 22, nxt)
            else:
                cont[pos] = (sz - 12, nxt)
            pos += sz
        for bid, (p, n, nxt) in heads.items():
            segs = [(p, n)]
            while nxt:
                if nxt not in cont:
                    raise GdsError("dangling block chain")
                n2, nxt2 = cont[nxt]
                segs.append((nxt + 12, n2))
                nxt = nxt2
            self._streams[bid] = segs

    def stream(self, bid: int) -> bytes:
        if bid not in self._streams:
            raise GdsError(f"no stream with id {bid}")
        return b"".join(self.buf[p:p + n] for p, n in self._streams[bid])

    def stream_size(self, bid: int) -> int:
        return sum(n for _, n in self._streams[bid])

    def stream_read(self, bid: int, off: int, n: int) -> bytes:
        """Bytes [off, off + n) of a stream without materialising the rest of it."""
        out, pos = [], 0
        for p, ln in self._streams[bid]:
            lo, hi = max(off, pos), min(off + n, pos + ln)
            if lo < hi:
                out.append(self.buf[p + lo - pos:p + hi - pos])
            pos += ln
            if pos >= off + n:
                break
        return b"".join(out)

    def _ra_index(self, nd: "GdsNode"):
        """Block table of an LZMA_RA stream: (stream offset, compressed size, raw offset, raw size).
        Stream layout: "XZ_RA" u8[3] u32 nblk u48 index_offset(after the 18-byte header) xz-blocks
        index[nblk] of (u24 compressed size, u32 raw size)."""
        if getattr(nd, "_ra", None) is None:
            head = self.stream_read(nd.data_id, 0, 18)
            if head[:5] != b"XZ_RA":
                raise GdsError("bad LZMA_RA header")
            nblk = struct.unpack_from("<I", head, 8)[0]
            ioff = _u48(head, 12) + 18
            idx = self.stream_read(nd.data_id, ioff, 7 * nblk)
            tab, so, ro = [], 18, 0
            for k in range(nblk):
                cs = int.from_bytes(idx[7 * k:7 * k + 3], "little")
                rs = struct.unpack_from("<I", idx, 7 * k + 3)[0]
                tab.append((so, cs, ro, rs))
                so += cs
                ro += rs
            nd._ra = tab
        return nd._ra

    def raw_range(self, path: str, lo: int, hi: int) -> bytes:
        """Bytes [lo, hi) of the decompressed payload of an array node."""
        nd = self.node(path)
        if nd.is_folder or nd.data_id is None:
            raise GdsError(f"{path}: not an array node")
        pipe = nd.pipe.upper()
        if pipe == "":
            return self.stream_read(nd.data_id, lo, hi - lo)
        if not pipe.startswith("LZMA_RA"):
            return self.raw(path)[lo:hi]              # one XZ stream: no random access
        out = []
        cache = self._tls.__dict__.setdefault("ra", {})     # last decoded block per node, per thread
        for k, (so, cs, ro, rs) in enumerate(self._ra_index(nd)):
            if ro + rs <= lo or ro >= hi:
                continue
            if cache.get(path, (-1, b""))[0] != k:
                blk = lzma.LZMADecompressor(format=lzma.FORMAT_XZ).decompress(self.stream_read(nd.data_id, so, cs))
                if len(blk) != rs:
                    raise GdsError("LZMA_RA block size mismatch")
                cache[path] = (k, blk)
            blk = cache[path][1]
            out.append(blk[max(lo, ro) - ro:min(hi, ro + rs) - ro])
        return b"".join(out)

    def read_rows(self, path: str, r0: int, r1: int) -> np.ndarray:
        """Rows [r0, r1) of a fixed-width array node (first dimension), decoding only what is needed."""
        nd = self.node(path)
        cls, dims = nd.cls or "", tuple(nd.dims or ())
        width = {"dInt32": 4, "dUInt32": 4, "dInt8": 1, "dUInt8": 1, "dInt16": 2, "dUInt16": 2, "dFloat32": 4,
                 "dFloat64": 8, "dPackedReal8U": 1, "dPackedReal8": 1}.get(cls)
        if width is None or not dims:
            raise GdsError(f"{path}: read_rows needs a fixed-width array, not {cls!r}")
        per = int(np.prod(dims[1:])) if len(dims) > 1 else 1
        data = self.raw_range(path, r0 * per * width, r1 * per * width)
        shape = (r1 - r0,) + dims[1:]
        if cls in ("dPackedReal8U", "dPackedReal8"):
            by = np.frombuffer(data, dtype=np.uint8)
            if cls == "dPackedReal8U":
                out = by.astype(np.float64) * nd.scale + nd.offset
                out[by == 0xFF] = np.nan
            else:
                sb = by.view(np.int8)
                out = sb.astype(np.float64) * nd.scale + nd.offset
                out[sb == -128] = np.nan
            return out.reshape(shape)
        dt = {"dInt32": "<i4", "dUInt32": "<u4", "dInt8": np.int8, "dUInt8": np.uint8, "dInt16": "<i2",
              "dUInt16": "<u2", "dFloat32": "<f4", "dFloat64": "<f8"}[cls]
        return np.frombuffer(data, dtype=dt).reshape(shape)

    # ------------------------------------------------------------------
    def _load_folder(self, node: GdsNode):
        s = self.stream(node.block_id)
        for m in re.finditer(re.escape(_P_ENTRY_ID), s):
            o = m.end()
            bid = struct.unpack_from("<I", s, o)[0]
            o += 4
            if s[o:o + 5] != _P_ENTRY_FLAG:
                continue
            flag = struct.unpack_from("<I", s, o + 5)[0]
            o += 9
            if s[o:o + 5] != _P_ENTRY_NAME:
                continue
            ln = s[o + 5]
            name = s[o + 6:o + 6 + ln].decode("utf-8")
            child = GdsNode(name, bid, False)
            node.children[name] = child
            cs = self.stream(bid)
            if flag & 0x2 and cs.find(_P_DATA) < 0:
                child.is_folder = True
                self._load_folder(child)
            else:
                self._load_array(child, cs)

    def _load_array(self, node: GdsNode, s: bytes):
        # class name: u48 size, u16 nprop, 0x00 0x01, n, 7-bit packed chars
        if s[8:10] == b"\x00\x01":
            n = s[10]
            nb = (7 * n + 7) // 8
            node.cls = _unpack7(s[11:11 + nb], n)
        i = s.find(_P_PIPE)
        if i >= 0:
            ln = s[i + 5]
            node.pipe = s[i + 6:i + 6 + ln].decode()
        i = s.find(_P_PIPE_SIZE)
        if i >= 0:
            node.raw_size = struct.unpack_from("<q", s, i + len(_P_PIPE_SIZE))[0]
        i = s.find(_P_DIM)
        if i >= 0:
            nb = s[i + 4]
            node.dims = struct.unpack_from("<%di" % (nb // 4), s, i + 5)
        i = s.find(_P_DATA)
        if i >= 0:
            node.data_id = struct.unpack_from("<I", s, i + 5)[0]
        i = s.find(_P_OFFSET)
        if i >= 0:
            node.offset = struct.unpack_from("<d", s, i + len(_P_OFFSET))[0]
        i = s.find(_P_SCALE)
        if i >= 0:
            node.scale = struct.unpack_from("<d", s, i + len(_P_SCALE))[0]

    # ------------------------------------------------------------------
    def node(self, path: str, silent: bool = False) -> Optional[GdsNode]:
        cur = self.root
        for part in [p for p in path.split("/") if p]:
            if part not in cur.children:
                if silent:
                    return None
                raise GdsError(f"No such GDS node \"{path}\"!")
            cur = cur.children[part]
        return cur

    def ls(self, path: str = "") -> List[str]:
        return list(self.node(path).children)

    def raw(self, path: str) -> bytes:
        """Decompressed payload bytes of an array node."""
        nd = self.node(path)
        if nd.is_folder or nd.data_id is None:
            raise GdsError(f"{path}: not an array node")
        s = self.stream(nd.data_id)
        pipe = nd.pipe.upper()
        if pipe == "":
            return s
        if pipe.startswith("LZMA_RA") or pipe.startswith("LZMA"):
            return _decode_xz(s, ra=pipe.startswith("LZMA_RA"))
        raise GdsError(f"{path}: unsupported compression {nd.pipe!r}")

    def read(self, path: str):
        """Array node -> numpy array (C order over the stored dims) or list[str]."""
        nd = self.node(path)
        data = self.raw(path)
        cls = nd.cls or ""
        dims = tuple(nd.dims or ())
        n = int(np.prod(dims)) if dims else 0
        if cls.startswith("dStr") or cls.startswith("dVStr") or cls.startswith("dCStr"):
            return _decode_strings(data, n)
        if cls in ("dInt32", "dUInt32"):
            return np.frombuffer(data, dtype="<i4" if cls == "dInt32" else "<u4", count=n).reshape(dims)
        if cls in ("dInt8", "dUInt8"):
            return np.frombuffer(data, dtype=np.int8 if cls == "dInt8" else np.uint8, count=n).reshape(dims)
        if cls in ("dInt16", "dUInt16"):
            return np.frombuffer(data, dtype="<i2" if cls == "dInt16" else "<u2", count=n).reshape(dims)
        if cls == "dFloat32":
            return np.frombuffer(data, dtype="<f4", count=n).reshape(dims)
        if cls == "dFloat64":
            return np.frombuffer(data, dtype="<f8", count=n).reshape(dims)
        if cls == "dBit2":
            by = np.frombuffer(data, dtype=np.uint8)
            v = np.empty((by.size, 4), dtype=np.uint8)
            for k in range(4):
                v[:, k] = (by >> (2 * k)) & 3
            return v.reshape(-1)[:n].reshape(dims)
        if cls in ("dPackedReal8U", "dPackedReal8"):
            by = np.frombuffer(data, dtype=np.uint8, count=n)
            if cls == "dPackedReal8U":
                out = by.astype(np.float64) * nd.scale + nd.offset
                out[by == 0xFF] = np.nan
            else:
                sb = by.view(np.int8)
                out = sb.astype(np.float64) * nd.scale + nd.offset
                out[sb == -128] = np.nan
            return out.reshape(dims)
        raise GdsError(f"{path}: unsupported array class {cls!r}")

    # ------------------------------------------------------------------
    # SeqArray-level accessors used by the scan
    def sample_id(self) -> List[str]:
        v = self.read("sample.id")
        return [str(x) for x in (v if isinstance(v, list) else v.tolist())]

    def n_variant(self) -> int:
        return int(self.node("variant.id").dims[0])

    def has_genotype(self) -> bool:
        nd = self.node("genotype/data", silent=True)
        return nd is not None

    def genotype_dims(self) -> Tuple[int, int]:
        """(variants, samples) of genotype/data after the checks of ``dosage_alt_packed``."""
        nd = self.node("genotype/data")
        if nd.cls != "dBit2" or len(nd.dims) != 3:
            raise GdsError("genotype/data: expected dBit2 [variant, sample, ploidy]")
        M, N, P = nd.dims
        if P != 2:
            raise GdsError("only diploid genotypes are supported")
        if not getattr(nd, "_reps_ok", False):
            if self.node("genotype/@data", silent=True) is not None:
                reps = np.asarray(self.read("genotype/@data")).reshape(-1)
                if reps.size and not np.all(reps == 1):
                    raise GdsError("multi-allelic (>2 bits) genotype storage is not supported")
            nd._reps_ok = True
        return M, N

    # per byte of genotype/data (two samples x two 2-bit allele codes) -> the two samples' dosage codes in
    # the low 4 bits.  $dosage_alt counts every non-reference allele (R/assoc_single.r:69-85 reads SeqArray's
    # $dosage_alt): allele codes 1 and 2 both count, 3 = missing makes the sample missing (code 3).
    _NIB = None

    @classmethod
    def _nibble_lut(cls):
        if cls._NIB is None:
            lut = np.zeros(256, dtype=np.uint8)
            for b in range(256):
                v = 0
                for smp in range(2):
                    a0, a1 = (b >> (4 * smp)) & 3, (b >> (4 * smp + 2)) & 3
                    d = 3 if (a0 == 3 or a1 == 3) else int(a0 != 0) + int(a1 != 0)
                    v |= d << (2 * smp)
                lut[b] = v
            cls._NIB = lut
        return cls._NIB

    def dosage_alt_packed_range(self, v0: int, v1: int, sample_sel: Optional[np.ndarray] = None) -> np.ndarray:
        """``$dosage_alt`` of variants [v0, v1) as 2-bit codes, 4 samples per byte: code = number of
        non-reference alleles, 3 = missing (any allele missing).  ``sample_sel``: sample indices to
        keep, in the order wanted.  Only the LZMA_RA blocks that hold the range are decoded."""
        M, N = self.genotype_dims()
        v0, v1 = max(0, v0), min(M, v1)
        m = v1 - v0
        bits0, bits1 = v0 * N * 4, v1 * N * 4
        data = np.frombuffer(self.raw_range("genotype/data", bits0 // 8, (bits1 + 7) // 8), dtype=np.uint8)
        if N % 2 == 0 and sample_sel is None:
            nib = self._nibble_lut()[data.reshape(m, N // 2)]          # rows are whole bytes
            if (N // 2) % 2:
                nib = np.concatenate([nib, np.zeros((m, 1), np.uint8)], axis=1)
            return (nib[:, 0::2] | (nib[:, 1::2] << 4)).astype(np.uint8)
        v = np.empty((data.size, 4), dtype=np.uint8)
        for k in range(4):
            v[:, k] = (data >> (2 * k)) & 3
        al = v.reshape(-1)[(bits0 % 8) // 2:][:m * N * 2].reshape(m, N, 2)
        ds = ((al != 0) & (al != 3)).sum(axis=2).astype(np.uint8)
        ds[(al == 3).any(axis=2)] = 3
        if sample_sel is not None:
            ds = ds[:, sample_sel]
        return pack_dosage_2bit(ds)

    def dosage_alt_packed(self) -> Tuple[np.ndarray, int, int]:
        """``$dosage_alt`` of every variant (see ``dosage_alt_packed_range``) -> (packed, N, M)."""
        M, N = self.genotype_dims()
        return self.dosage_alt_packed_range(0, M), N, M

    def dosage_real_range(self, path: str, v0: int, v1: int) -> np.ndarray:
        """Rows [v0, v1) of a real-valued dosage node [M, N] (NaN = missing)."""
        nd = self.node(path + "/data")
        if not getattr(nd, "_reps_ok", False):
            if self.node(path + "/@data", silent=True) is not None:
                reps = np.asarray(self.read(path + "/@data")).reshape(-1)
                if reps.size and not np.all(reps == 1):
                    raise GdsError(f"{path}: more than one value per variant")
            nd._reps_ok = True
        return np.asarray(self.read_rows(path + "/data", v0, v1), dtype=np.float64)

    def dosage_real(self, path: str = "annotation/format/DS") -> np.ndarray:
        """Real-valued dosages [M, N] (NaN = missing) from a format node."""
        at = self.node(path + "/@data", silent=True)
        if at is not None:
            reps = np.asarray(self.read(path + "/@data")).reshape(-1)
            if reps.size and not np.all(reps == 1):
                raise GdsError(f"{path}: more than one value per variant")
        return np.asarray(self.read(path + "/data"), dtype=np.float64)

    def alleles(self) -> Tuple[List[str], List[str]]:
        a = self.read("allele")
        ref, alt = [], []
        for s in a:
            r, _, t = s.partition(",")
            ref.append(r)
            alt.append(t)
        return ref, alt


def _decode_strings(data: bytes, n: int) -> List[str]:
    """dStr8 payload: per string a 7-bit varint byte length, then the bytes."""
    out, p = [], 0
    for _ in range(n):
        ln, sh = 0, 0
        while True:
            c = data[p]
            p += 1
            ln |= (c & 0x7F) << sh
            sh += 7
            if not c & 0x80:
                break
        out.append(data[p:p + ln].decode("utf-8", errors="replace"))
        p += ln
    return out


def _decode_xz(s: bytes, ra: bool) -> bytes:
    out = []
    pos = 0
    nblk = None
    if ra:
        if s[:5] != b"XZ_RA":
            raise GdsError("bad LZMA_RA header")
        nblk = struct.unpack_from("<I", s, 8)[0]
        pos = 18
    done = 0
    while pos < len(s) and (nblk is None or done < nblk):
        j = s.find(_XZ, pos)
        if j < 0:
            break
        d = lzma.LZMADecompressor(format=lzma.FORMAT_XZ)
        out.append(d.decompress(s[j:]))
        if not d.eof:
            raise GdsError("truncated XZ block")
        pos = len(s) - len(d.unused_data)
        done += 1
    return b"".join(out)


def pack_dosage_2bit(ds: np.ndarray) -> np.ndarray:
    """[M, N] codes 0..3 -> [M, ceil(N/4)] bytes, sample 4b+k in bits 2k..2k+1."""
    ds = np.ascontiguousarray(ds, dtype=np.uint8)
    M, N = ds.shape
    nb = (N + 3) // 4
    pad = np.zeros((M, nb * 4), dtype=np.uint8)
    pad[:, :N] = ds & 3
    q = pad.reshape(M, nb, 4)
    return (q[:, :, 0] | (q[:, :, 1] << 2) | (q[:, :, 2] << 4) | (q[:, :, 3] << 6)).astype(np.uint8)


def unpack_dosage_2bit(packed: np.ndarray, n_samp: int) -> np.ndarray:
    """Inverse of :func:`pack_dosage_2bit` -> [M, N] uint8 codes."""
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    M, nb = packed.shape
    out = np.empty((M, nb, 4), dtype=np.uint8)
    for k in range(4):
        out[:, :, k] = (packed >> (2 * k)) & 3
    return out.reshape(M, nb * 4)[:, :n_samp]
