"""Narrow reader for CoreArray GDS containers written by SeqArray/gdsfmt.

The reference reads genotypes through SeqArray (``seqOpen``/``seqGetData``/
``seqApply``; R/assoc_single.r:116-144,202-221), a Bioconductor dependency
that is not part of the reference tree.  This module decodes just what the
single-variant scan consumes:

* the directory tree (folder streams),
* array nodes of class dBit2 (``genotype/data``), dInt32, dStr8/dStr16,
  dFloat32/64, dPackedReal8[U] / dPackedReal16[U] (``annotation/format/DS/data``),
* stored uncompressed, as one compressed stream (``LZMA``, ``ZIP``) or in gdsfmt's random-access
  form: independent blocks behind an 18-byte header with a block index at the end (``LZMA_RA``:
  XZ blocks, pinned by the reference's own files; ``ZIP_RA`` zlib and ``LZ4_RA`` LZ4 blocks in the
  same framing -- no file of the reference uses them, so their framing is unpinned: the reader
  accepts zlib / raw-deflate / gzip and LZ4 frame / raw LZ4 block payloads alike).

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
import zlib
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
        self._lock = threading.Lock()
        self._whole: Dict[str, object] = {}      # decoded payload of the last node without random access
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

    def stream_view(self, bid: int, off: int, n: int):
        """As ``stream_read``, but without a copy where the range lies inside one segment of the stream: a
        memoryview of the mapped file (the block decoders and ``numpy.frombuffer`` read from it directly, and
        the page faults of a large uncompressed node are then taken by the decoder's threads, not here)."""
        pos = 0
        for p, ln in self._streams[bid]:
            if pos <= off and off + n <= pos + ln:
                return memoryview(self.buf)[p + off - pos:p + off - pos + n]
            pos += ln
            if pos >= off + n:
                break
        return self.stream_read(bid, off, n)

    _RA_MAGIC = {b"XZ_RA": "xz", b"ZIP_RA": "zip", b"LZ4_RA": "lz4"}

    def _ra_index(self, nd: "GdsNode"):
        """Block table of a random-access stream: (stream offset, compressed size, raw offset, raw size).
        Stream layout: magic + version (7 bytes: "XZ_RA" v v | "ZIP_RA" v | "LZ4_RA" v) u8 block-size code
        u32 nblk u48 index_offset(after the 18-byte header) blocks index[nblk] of (u24 compressed size, u32 raw size)."""
        if getattr(nd, "_ra", None) is None:
            head = self.stream_read(nd.data_id, 0, 18)
            kind = next((k for m, k in self._RA_MAGIC.items() if head.startswith(m)), None)
            if kind is None:
                raise GdsError("bad random-access stream header")
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
            nd._ra, nd._ra_kind = tab, kind
        return nd._ra

    def raw_range(self, path: str, lo: int, hi: int) -> bytes:
        """Bytes [lo, hi) of the decompressed payload of an array node."""
        nd = self.node(path)
        if nd.is_folder or nd.data_id is None:
            raise GdsError(f"{path}: not an array node")
        pipe = nd.pipe.upper()
        if pipe == "":
            return self.stream_view(nd.data_id, lo, hi - lo)
        if not pipe.endswith("_RA") and "_RA:" not in pipe and "_RA" not in pipe:
            # one compressed stream: no random access -- decode once per file object, not once per block
            with self._lock:
                if self._whole.get("path") != path:
                    self._whole = {"path": path, "data": self.raw(path)}
                return self._whole["data"][lo:hi]
        out = []
        cache = self._tls.__dict__.setdefault("ra", {})     # last decoded block per node, per thread
        tab = self._ra_index(nd)
        for k, (so, cs, ro, rs) in enumerate(tab):
            if ro + rs <= lo or ro >= hi:
                continue
            if cache.get(path, (-1, b""))[0] != k:
                blk = _decode_block(nd._ra_kind, self.stream_view(nd.data_id, so, cs), rs)
                if len(blk) != rs:
                    raise GdsError("random-access block size mismatch")
                cache[path] = (k, blk)
            blk = cache[path][1]
            out.append(blk[max(lo, ro) - ro:min(hi, ro + rs) - ro])
        return b"".join(out)

    def read_rows(self, path: str, r0: int, r1: int) -> np.ndarray:
        """Rows [r0, r1) of a fixed-width array node (first dimension), decoding only what is needed."""
        nd = self.node(path)
        cls, dims = nd.cls or "", tuple(nd.dims or ())
        width = {"dInt32": 4, "dUInt32": 4, "dInt8": 1, "dUInt8": 1, "dInt16": 2, "dUInt16": 2, "dFloat32": 4,
                 "dFloat64": 8, "dPackedReal8U": 1, "dPackedReal8": 1, "dPackedReal16U": 2, "dPackedReal16": 2}.get(cls)
        if width is None or not dims:
            raise GdsError(f"{path}: read_rows needs a fixed-width array, not {cls!r}")
        per = int(np.prod(dims[1:])) if len(dims) > 1 else 1
        data = self.raw_range(path, r0 * per * width, r1 * per * width)
        shape = (r1 - r0,) + dims[1:]
        if cls in ("dPackedReal16U", "dPackedReal16"):
            return _packed_real16(np.frombuffer(data, dtype="<u2"), cls, nd.scale, nd.offset).reshape(shape)
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
        if "_RA" in pipe:
            tab = self._ra_index(nd)
            return b"".join(_decode_block(nd._ra_kind, s[so:so + cs], rs) for so, cs, _, rs in tab)
        if pipe.startswith("LZMA"):
            return _decode_xz(s, ra=False)
        if pipe.startswith("ZIP"):
            return _decode_block("zip", s, None)
        if pipe.startswith("LZ4"):
            return _decode_block("lz4", s, None)
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
        if cls in ("dPackedReal16U", "dPackedReal16"):
            return _packed_real16(np.frombuffer(data, dtype="<u2", count=n), cls, nd.scale, nd.offset).reshape(dims)
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
        """(variants, samples) of genotype/data.  A site with more than three alleles takes more than one row
        of 2-bit codes (SeqArray: ``genotype/@data`` holds the number of rows per variant; the rows are the
        base-4 digits of the allele index, least significant first, all bits set = missing), so the number of
        variants is the length of ``@data`` and the row offsets are its running sum (``_geno_rows``)."""
        nd = self.node("genotype/data")
        if nd.cls != "dBit2" or len(nd.dims) != 3:
            raise GdsError("genotype/data: expected dBit2 [variant, sample, ploidy]")
        R, N, P = nd.dims
        if P != 2:
            raise GdsError("only diploid genotypes are supported")
        if getattr(nd, "_row0", None) is None:
            if self.node("genotype/@data", silent=True) is not None:
                reps = np.asarray(self.read("genotype/@data")).reshape(-1).astype(np.int64)
                if reps.size and (reps.min() < 1 or int(reps.sum()) != R):
                    raise GdsError("genotype/@data does not add up to the rows of genotype/data")
            else:
                reps = np.ones(R, dtype=np.int64)
            nd._row0 = np.concatenate([[0], np.cumsum(reps)])      # rows of variant v: [_row0[v], _row0[v + 1])
            nd._multi = bool(reps.size and reps.max() > 1)
        return int(nd._row0.size - 1), N

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

    def dosage_alt_packed_range(self, v0: int, v1: int, sample_sel: Optional[np.ndarray] = None,
                                out: Optional[np.ndarray] = None, chunk_bytes: int = 64 << 20) -> np.ndarray:
        """``$dosage_alt`` of variants [v0, v1) as 2-bit codes, 4 samples per byte: code = number of
        non-reference alleles, 3 = missing (any allele missing).  ``sample_sel``: sample indices to
        keep, in the order wanted.  ``out``: a [>= v1 - v0, >= ceil(n / 4)] uint8 buffer to fill (a
        pinned block buffer, say; bytes beyond a row's codes are zeroed) -- else a new array.
        Only the compressed blocks that hold the range are decoded, and the work is done in pieces of
        about ``chunk_bytes`` of temporaries, whatever the block size: at N = 430 000 a block of 50 000
        variants is 10.75 GB of allele codes, which never exist at once (the reference's seqApply works
        variant by variant)."""
        M, N = self.genotype_dims()
        v0, v1 = max(0, v0), min(M, v1)
        m = v1 - v0
        nd = self.node("genotype/data")
        if nd._multi and m > 0 and nd._row0[v1] - nd._row0[v0] != m:
            return self._dosage_alt_multirow(v0, v1, sample_sel, out, chunk_bytes)
        r0 = int(nd._row0[v0]) if m > 0 else 0                 # (a range of single-row variants: rows r0 .. r0 + m)
        sel = None if sample_sel is None else np.asarray(sample_sel, dtype=np.int64)
        n_out = N if sel is None else int(sel.size)
        nb = (n_out + 3) // 4
        if out is None:
            out = np.zeros((m, nb), dtype=np.uint8)
        elif out.shape[0] < m or out.shape[1] < nb or out.dtype != np.uint8:
            raise ValueError("dosage_alt_packed_range: `out` too small")
        res = out[:m]
        if res.shape[1] > nb:
            res[:, nb:] = 0
        # variants per piece: the allele bytes of a piece (N / 2 each) and, for a sample subset, one byte per
        # selected (variant, sample) stay under chunk_bytes
        per = N // 2 + 1 + (2 * n_out if sel is not None else 0) + nb
        step = max(1, min(m, chunk_bytes // per)) if m else 1
        lut = self._nibble_lut()
        native = _native_decoder() if res.strides[1] == 1 else None
        for a in range(0, m, step):
            b = min(m, a + step)
            k = b - a
            bits0, bits1 = (r0 + a) * N * 4, (r0 + b) * N * 4
            data = np.frombuffer(self.raw_range("genotype/data", bits0 // 8, (bits1 + 7) // 8), dtype=np.uint8)
            if native is not None:
                # libsaigehip's host decoder (sgx_decode_dbit2): rows over host threads, straight into `out`
                native(data, bits0 % 8, N, k, res[a:b], sel)
                continue
            if N % 2 == 0:
                nib = lut[data.reshape(k, N // 2)]                  # per byte: the dosage codes of two samples
                if sel is None:
                    if (N // 2) % 2:
                        nib = np.concatenate([nib, np.zeros((k, 1), np.uint8)], axis=1)
                    np.bitwise_or(nib[:, 0::2], nib[:, 1::2] << 4, out=res[a:b, :nb])
                    continue
                codes = (nib[:, sel >> 1] >> ((sel & 1) << 1).astype(np.uint8)) & 3          # [k, n_out] uint8
            else:
                # an odd number of samples: rows are not whole bytes -- through the 2-bit allele codes
                al = np.empty((data.size, 4), dtype=np.uint8)
                for q in range(4):
                    al[:, q] = (data >> (2 * q)) & 3
                al = al.reshape(-1)[(bits0 % 8) // 2:][:k * N * 2].reshape(k, N, 2)
                codes = ((al[:, :, 0] != 0) & (al[:, :, 0] != 3)).astype(np.uint8) + ((al[:, :, 1] != 0) & (al[:, :, 1] != 3))
                codes[(al[:, :, 0] == 3) | (al[:, :, 1] == 3)] = 3
                if sel is not None:
                    codes = codes[:, sel]
            res[a:b, :nb] = pack_dosage_2bit(codes)
        return res if res.shape[1] == nb else res

    def _dosage_alt_multirow(self, v0, v1, sample_sel, out, chunk_bytes):
        """The range holds sites of more than three alleles: through the allele indices (numpy; such sites are
        rare, so this path is not tuned).  Runs of single-row variants inside the range take the fast path."""
        M, N = self.genotype_dims()
        nd = self.node("genotype/data")
        m = v1 - v0
        sel = None if sample_sel is None else np.asarray(sample_sel, dtype=np.int64)
        n_out = N if sel is None else int(sel.size)
        nb = (n_out + 3) // 4
        if out is None:
            out = np.zeros((m, nb), dtype=np.uint8)
        elif out.shape[0] < m or out.shape[1] < nb or out.dtype != np.uint8:
            raise ValueError("dosage_alt_packed_range: `out` too small")
        res = out[:m]
        res[:, nb:] = 0
        reps = np.diff(nd._row0[v0:v1 + 1])
        v = v0
        while v < v1:
            if reps[v - v0] == 1:
                e = v
                while e < v1 and reps[e - v0] == 1:
                    e += 1
                self.dosage_alt_packed_range(v, e, sample_sel, res[v - v0:e - v0], chunk_bytes)
                v = e
                continue
            ra, rb = int(nd._row0[v]), int(nd._row0[v + 1])
            bits0, bits1 = ra * N * 4, rb * N * 4
            data = np.frombuffer(self.raw_range("genotype/data", bits0 // 8, (bits1 + 7) // 8), dtype=np.uint8)
            dig = np.empty((data.size, 4), dtype=np.uint8)
            for q in range(4):
                dig[:, q] = (data >> (2 * q)) & 3
            dig = dig.reshape(-1)[(bits0 % 8) // 2:][:(rb - ra) * N * 2].reshape(rb - ra, N, 2).astype(np.int64)
            idx = np.zeros((N, 2), dtype=np.int64)
            for k in range(rb - ra):
                idx |= dig[k] << (2 * k)
            miss = idx == (1 << (2 * (rb - ra))) - 1
            codes = ((idx[:, 0] != 0) & ~miss[:, 0]).astype(np.uint8) + ((idx[:, 1] != 0) & ~miss[:, 1])
            codes[miss[:, 0] | miss[:, 1]] = 3
            if sel is not None:
                codes = codes[sel]
            res[v - v0, :nb] = pack_dosage_2bit(codes[None, :])[0]
            v += 1
        return res

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


_NATIVE = [False, None]


def _native_decoder():
    """sgx_decode_dbit2 of libsaigehip.so when the library has been built (it needs no GPU); else None and
    the numpy path below does the same thing."""
    if not _NATIVE[0]:
        _NATIVE[0] = True
        try:
            from . import _lib
            _lib.load()
            _NATIVE[1] = _lib.decode_dbit2
        except Exception:       # noqa: BLE001 -- library not built: numpy path
            _NATIVE[1] = None
    return _NATIVE[1]


def _packed_real16(raw: np.ndarray, cls: str, scale: float, offset: float) -> np.ndarray:
    """dPackedReal16U: value = raw * scale + offset, 0xFFFF = NaN; dPackedReal16: signed, -32768 = NaN."""
    if cls == "dPackedReal16U":
        out = raw.astype(np.float64) * scale + offset
        out[raw == 0xFFFF] = np.nan
    else:
        sv = raw.view(np.int16)
        out = sv.astype(np.float64) * scale + offset
        out[sv == -32768] = np.nan
    return out


def lz4_block_decode(src: bytes, raw_size: Optional[int] = None) -> bytes:
    """LZ4 block format (sequences of literals + matches), pure Python.  Matches are copied by slices (an
    overlapping match -- the format's run-length trick -- by doubling); a block cut off inside a sequence is a
    GdsError.  Parity unpinned: the reference holds no LZ4_RA file (README)."""
    out = bytearray()
    i, n = 0, len(src)
    try:
        while i < n:
            tok = src[i]
            i += 1
            ll = tok >> 4
            if ll == 15:
                while True:
                    c = src[i]
                    i += 1
                    ll += c
                    if c != 255:
                        break
            if i + ll > n:
                raise GdsError("LZ4: literals run past the end of the block")
            out += src[i:i + ll]
            i += ll
            if i >= n:
                break
            off = src[i] | (src[i + 1] << 8)
            i += 2
            if off == 0:
                raise GdsError("LZ4: zero offset")
            ml = (tok & 15) + 4
            if (tok & 15) == 15:
                while True:
                    c = src[i]
                    i += 1
                    ml += c
                    if c != 255:
                        break
            st = len(out) - off
            if st < 0:
                raise GdsError("LZ4: offset before the start of the block")
            if off >= ml:
                out += out[st:st + ml]
            else:
                seg = bytes(out[st:])              # the last `off` bytes, repeated
                reps = -(-ml // off)
                out += (seg * reps)[:ml]
    except IndexError:
        raise GdsError("LZ4: block truncated inside a sequence") from None
    if raw_size is not None and len(out) != raw_size:
        raise GdsError("LZ4: block size mismatch")
    return bytes(out)


def lz4_block_encode(raw: bytes) -> bytes:
    """A valid (not a good) LZ4 block: greedy 4-byte hash matches; for the writer's test files."""
    n, i, anchor = len(raw), 0, 0
    out = bytearray()
    table: Dict[bytes, int] = {}

    def emit(lit: bytes, off: int, ml: int):
        tok_l = min(len(lit), 15)
        tok_m = min(ml - 4, 15) if ml else 0
        out.append((tok_l << 4) | tok_m)
        if len(lit) >= 15:
            r = len(lit) - 15
            while r >= 255:
                out.append(255)
                r -= 255
            out.append(r)
        out.extend(lit)
        if ml:
            out.extend((off & 0xFF, off >> 8))
            if ml - 4 >= 15:
                r = ml - 4 - 15
                while r >= 255:
                    out.append(255)
                    r -= 255
                out.append(r)

    while i + 12 < n:                       # the last 12 bytes stay literals (format rule: 5, kept simple)
        key = raw[i:i + 4]
        j = table.get(key)
        table[key] = i
        if j is not None and i - j <= 0xFFFF:
            ml = 4
            while i + ml < n - 5 and raw[j + ml] == raw[i + ml]:
                ml += 1
            emit(raw[anchor:i], i - j, ml)
            i += ml
            anchor = i
        else:
            i += 1
    emit(raw[anchor:], 0, 0)
    return bytes(out)


def _decode_block(kind: str, blk: bytes, raw_size: Optional[int]) -> bytes:
    """One block of a random-access stream (or a whole single-stream payload)."""
    if kind == "xz":
        return lzma.LZMADecompressor(format=lzma.FORMAT_XZ).decompress(blk)
    if kind == "zip":
        for wbits in (15, -15, 31):          # zlib header, raw deflate, gzip
            try:
                return zlib.decompressobj(wbits).decompress(blk)
            except zlib.error:
                continue
        raise GdsError("ZIP block: not a zlib / deflate / gzip stream")
    if kind == "lz4":
        if blk[:4] == b"\x04\x22\x4d\x18":   # LZ4 frame: descriptor, then blocks (u32 size, bit 31 = stored)
            flg = blk[4]
            p = 6 + (8 if flg & 0x08 else 0) + (4 if flg & 0x01 else 0) + 1
            out = []
            while True:
                bs = struct.unpack_from("<I", blk, p)[0]
                p += 4
                if bs == 0:
                    break
                body = blk[p:p + (bs & 0x7FFFFFFF)]
                p += bs & 0x7FFFFFFF
                out.append(body if bs >> 31 else lz4_block_decode(body))
                if flg & 0x10:
                    p += 4
            return b"".join(out)
        return lz4_block_decode(blk, raw_size)
    raise GdsError(f"unknown block coder {kind!r}")


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
