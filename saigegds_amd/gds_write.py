"""Writer for the CoreArray GDS container, as far as ``seqAssocGLMM_SPA(res.savefn="*.gds")`` needs it
(reference R/assoc_single.r:243-286: ``createfn.gds`` + ``add.gdsn`` of character, integer, double and
logical vectors under the root, two root attributes).

gdsfmt is not part of the reference tree; the byte layout below is the one read off the reference's
own files (inst/extdata/*.gds, see saigegds_amd/gds.py and SURVEY.md App. B) and every record is
emitted exactly as it appears there:

  file    := "COREARRAYx0A" 00 01  u32 root_id  block*
  block   := u48 size|bit47  u48 next(0)  u32 stream_id  u48 stream_size  payload       (one block per stream)
  node    := u48 size  u16 nprop  [class]  prop*                                         (a stream of its own)
  folder  := DIRCNT u32 n  [DIRLIST u48 size  entry*]  ATTRCNT u32 n  [ATTRDATA u48 size  attr*]
  entry   := u48 size  u16 3  ID u32  FLAG u32  NAME len bytes
  array   := class  [PIPE "LZMA_ra"  PIPE_SIZE u64 raw u64 stored  LEVEL  BLOCK]  DCNT u16  DIM u32[]  DATA u32  ATTRCNT ...
  data    := raw bytes | "XZ_RA" 10 11 ff  u32 nblk  u48 index_offset  xz-block*  (u24 stored, u32 raw)*
Strings are stored as a 7-bit varint length followed by the bytes (dStr8); logical vectors as dInt32
with the attribute ``R.logical`` (what gdsfmt's add.gdsn does for them).
"""
from __future__ import annotations

import lzma
import struct
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

_MAGIC = b"COREARRAYx0A"
# records as they stand in the reference's files (type byte + packed property name)
_R_DIRCNT = bytes.fromhex("08c643754ef601")
_R_DIRLIST = bytes.fromhex("01c7437517e57d")
_R_ENTRY_ID = b"\t\x02\xf5\x00"
_R_ENTRY_FLAG = b"\tDt1\x12"
_R_ENTRY_NAME = b"\x15D\xc6`\x10"
_R_ATTRCNT = bytes.fromhex("0807f37d9d937d")
_R_ATTRDATA = bytes.fromhex("0108f37ddd45791f")
_R_PIPE = b"\x15\xc4Fm\x10"
_R_PIPE_SIZE = b"\x02\xc9FmP\xe0Q%\x04\x10"
_R_PIPE_LEVEL = bytes.fromhex("05ca466d507041217401") + b"\x02"
_R_PIPE_BLOCK = bytes.fromhex("05cb466d50d0581e5542") + b"\x04"
_R_DCNT = b"\x07\xc4\xe3d\x1f"
_R_DIM = b"\x02\xc3Ca"
_R_DATA = b"\t\xc4\xc3|\x0c"
_RA_BLOCK = 2359296      # raw bytes per LZMA_RA block in the reference's files


def _u48(v: int) -> bytes:
    return int(v).to_bytes(6, "little")


def _pack7(s: str) -> bytes:
    """n characters, 7 bits each, LSB first (class names)."""
    v = 0
    for i, ch in enumerate(s):
        v |= (ord(ch) & 0x7F) << (7 * i)
    return bytes([len(s)]) + v.to_bytes((7 * len(s) + 7) // 8, "little")


def _varint(n: int) -> bytes:
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _attrs(attrs: Sequence[Tuple[str, Optional[str]]]) -> bytes:
    """ATTRCNT (+ ATTRDATA): value None = an attribute without a value (type 0), else a short string."""
    out = _R_ATTRCNT + struct.pack("<I", len(attrs))
    if attrs:
        body = b""
        for name, val in attrs:
            nb = name.encode()
            body += bytes([len(nb)]) + nb
            if val is None:
                body += b"\x00"
            else:
                vb = val.encode()
                if len(vb) > 255:
                    raise ValueError("attribute value too long")
                body += b"\x0e" + bytes([len(vb)]) + vb
        out += _R_ATTRDATA + _u48(6 + len(body)) + body
    return out


def _node_stream(nprop: int, body: bytes) -> bytes:
    return _u48(8 + len(body)) + struct.pack("<H", nprop) + body


def _ra_stream(raw: bytes, kind: str = "xz", block: int = _RA_BLOCK):
    """gdsfmt's random-access stream: 18-byte header, independent blocks, then the block index.  kind: "xz"
    (LZMA_RA: the framing of the reference's own files), "zip" (zlib blocks), "lz4" (LZ4 blocks) -- the last
    two in the same framing by analogy; no file of the reference pins them."""
    import zlib
    from .gds import lz4_block_encode
    magic = {"xz": b"XZ_RA\x10\x11", "zip": b"ZIP_RA\x10", "lz4": b"LZ4_RA\x10"}[kind]
    blocks, index = [], b""
    for off in range(0, max(len(raw), 1), block):
        piece = raw[off:off + block]
        if kind == "xz":
            z = lzma.compress(piece, format=lzma.FORMAT_XZ, check=lzma.CHECK_CRC32, preset=6)
        elif kind == "zip":
            z = zlib.compress(piece, 6)
        else:
            z = lz4_block_encode(piece)
        if len(z) >= 1 << 24:
            raise ValueError("random-access block does not fit its 24-bit size field")
        blocks.append(z)
        index += len(z).to_bytes(3, "little") + struct.pack("<I", len(piece))
    body = b"".join(blocks)
    return magic + b"\xff" + struct.pack("<I", len(blocks)) + _u48(len(body)) + body + index, len(blocks)


def _lzma_ra(raw: bytes):
    return _ra_stream(raw, "xz")


_P_OFFSET = b"\x13\x86\x16E\x1e\xf4\x01"
_P_SCALE = b"\x13\x85\xe70\x17\x04"


class GdsWriter:
    """GDS file: array nodes in folders ("genotype/data"), string attributes on the root."""

    def __init__(self, path: str):
        self.path = path
        # tree: name -> ("array", node stream with the data id to be patched, data stream) | ("folder", children)
        self._tree: Dict[str, Any] = {}
        self._root_attrs: List[Tuple[str, Optional[str]]] = []

    def put_attr(self, name: str, value: str):
        self._root_attrs.append((name, value))

    def _slot(self, name: str):
        parts = [p for p in name.split("/") if p]
        cur = self._tree
        for p in parts[:-1]:
            cur = cur.setdefault(p, ("folder", {}))[1]
        return cur, parts[-1]

    def add(self, name: str, values: Any, compress: str = "LZMA_RA", cls: Optional[str] = None,
            dims: Optional[Sequence[int]] = None, scale: Optional[float] = None, offset: Optional[float] = None,
            ra_block: int = _RA_BLOCK):
        """add.gdsn(node, name, values, compress=, closezip=TRUE).  cls / dims: store ``values`` (raw bytes or a
        uint8 array) as that class with those dimensions -- dBit2 allele codes, dPackedReal8U/16U dosages
        (with scale / offset)."""
        attrs: List[Tuple[str, Optional[str]]] = []
        if cls is not None:
            raw = values if isinstance(values, (bytes, bytearray)) else np.ascontiguousarray(values).tobytes()
            dims = tuple(int(d) for d in dims)
        elif isinstance(values, (list, tuple)) and (len(values) == 0 or isinstance(values[0], str)):
            cls, dims = "dStr8", (len(values),)
            raw = b"".join(_varint(len(b)) + b for b in (str(v).encode("utf-8") for v in values))
        else:
            a = np.asarray(values)
            dims = (a.size,)
            if a.dtype == np.bool_:
                cls, raw = "dInt32", a.astype("<i4").tobytes()
                attrs.append(("R.logical", None))
            elif np.issubdtype(a.dtype, np.integer):
                cls, raw = "dInt32", a.astype("<i4").tobytes()
            elif np.issubdtype(a.dtype, np.floating):
                cls, raw = "dFloat64", a.astype("<f8").tobytes()
            else:
                raise TypeError(f"{name}: cannot store {a.dtype}")
        body = b"\x00\x01" + _pack7(cls)
        nprop = 0
        comp = (compress or "none").upper()
        if comp != "NONE":
            # LZMA and LZMA_RA are written as LZMA_ra (the coder whose stream layout is known from the
            # reference's files); ZIP_RA / LZ4_RA in the same framing with their own block coder
            kind = "zip" if comp.startswith("ZIP") else "lz4" if comp.startswith("LZ4") else "xz"
            data, nblk = _ra_stream(raw, kind, ra_block)
            pname = {"xz": b"LZMA_ra", "zip": b"ZIP_ra", "lz4": b"LZ4_ra"}[kind]
            body += _R_PIPE + bytes([len(pname)]) + pname + _R_PIPE_SIZE + struct.pack("<qq", len(raw), len(data) - 7 * nblk)
            body += _R_PIPE_LEVEL + _R_PIPE_BLOCK
            nprop += 4
        else:
            data = raw
        body += _R_DCNT + struct.pack("<H", len(dims)) + _R_DIM + bytes([4 * len(dims)]) + struct.pack("<%di" % len(dims), *dims)
        body += _R_DATA + b"\xff\xff\xff\xff"               # data stream id, patched in close()
        nprop += 3
        if offset is not None:
            body += _P_OFFSET + struct.pack("<d", float(offset))
            nprop += 1
        if scale is not None:
            body += _P_SCALE + struct.pack("<d", float(scale))
            nprop += 1
        body += _attrs(attrs)
        nprop += 1 + (1 if attrs else 0)
        cur, leaf = self._slot(name)
        cur[leaf] = ("array", _node_stream(nprop, body), data)

    def close(self):
        streams: Dict[int, bytes] = {}
        counter = [2]

        def folder_stream(children: Dict[str, Any], root_attrs) -> bytes:
            entries = b""
            for name, node in children.items():
                nid = counter[0]
                counter[0] += 1
                if node[0] == "array":
                    did = counter[0]
                    counter[0] += 1
                    ns = node[1]
                    k = ns.index(_R_DATA + b"\xff\xff\xff\xff")
                    streams[nid] = ns[:k + len(_R_DATA)] + struct.pack("<I", did) + ns[k + len(_R_DATA) + 4:]
                    streams[did] = node[2]
                    flag = 0
                else:
                    streams[nid] = folder_stream(node[1], None)
                    flag = 2
                nb = name.encode()
                rec = (_R_ENTRY_ID + struct.pack("<I", nid) + _R_ENTRY_FLAG + struct.pack("<I", flag) + _R_ENTRY_NAME
                       + bytes([len(nb)]) + nb)
                entries += _u48(8 + len(rec)) + struct.pack("<H", 3) + rec
            body = _R_DIRCNT + struct.pack("<I", len(children))
            nprop = 1
            if children:
                body += _R_DIRLIST + _u48(6 + len(entries)) + entries
                nprop += 1
            body += _attrs(root_attrs or [])
            nprop += 1 + (1 if root_attrs else 0)
            return _node_stream(nprop, body)

        streams[1] = folder_stream(self._tree, self._root_attrs)
        with open(self.path, "wb") as f:
            f.write(_MAGIC + b"\x00\x01" + struct.pack("<I", 1))
            for sid in sorted(streams):
                s = streams[sid]
                f.write(_u48((22 + len(s)) | (1 << 47)) + _u48(0) + struct.pack("<I", sid) + _u48(len(s)) + s)


def write_seqarray_alleles(fn: str, alleles: np.ndarray, sample_id: Optional[Sequence[str]] = None,
                           compress: str = "none", ra_block: int = _RA_BLOCK):
    """A SeqArray-style genotype file from allele indices ``alleles`` [M, N, 2] (0 = reference, -1 = missing):
    a site whose largest index does not fit below the all-ones missing code of n rows takes n + 1 rows of
    2-bit codes (base-4 digits, least significant row first); ``genotype/@data`` holds the rows per variant.
    Test generator for sites with more than three alleles."""
    al = np.asarray(alleles, dtype=np.int64)
    M, N, _ = al.shape
    rows, reps = [], []
    for v in range(M):
        top = int(al[v].max())
        n = 1
        while top >= 4 ** n - 1:
            n += 1
        a = np.where(al[v] < 0, 4 ** n - 1, al[v])
        for k in range(n):
            rows.append(((a >> (2 * k)) & 3).astype(np.uint8).reshape(-1))          # [N * 2] codes of row k
        reps.append(n)
    codes = np.concatenate(rows)
    pad = (-codes.size) % 4
    codes = np.concatenate([codes, np.zeros(pad, np.uint8)]).reshape(-1, 4)
    raw = (codes[:, 0] | (codes[:, 1] << 2) | (codes[:, 2] << 4) | (codes[:, 3] << 6)).astype(np.uint8).tobytes()
    R = int(np.sum(reps))
    w = GdsWriter(fn)
    w.put_attr("FileFormat", "SEQ_ARRAY")
    sid = [f"s{i + 1}" for i in range(N)] if sample_id is None else [str(x) for x in sample_id]
    w.add("sample.id", sid, "none")
    w.add("variant.id", np.arange(1, M + 1), "none")
    w.add("position", np.arange(1, M + 1) * 100, "none")
    w.add("chromosome", ["1"] * M, "none")
    w.add("allele", [",".join("ACGTN"[:1] + "CGTAN"[:1]) if r == 1 else "A,C,G,T,AA,CC" for r in reps], "none")
    w.add("genotype/data", raw, compress, cls="dBit2", dims=(R, N, 2), ra_block=ra_block)
    w.add("genotype/@data", np.asarray(reps, dtype=np.uint8).tobytes(), "none", cls="dUInt8", dims=(M,))
    w.close()


def write_seqarray_genotypes(fn: str, packed: np.ndarray, n_samp: int, sample_id: Optional[Sequence[str]] = None,
                             compress: str = "none", chromosome: str = "1", ra_block: int = _RA_BLOCK):
    """A SeqArray-style genotype file from 2-bit dosage rows (``packed``: [M, >= ceil(N / 4)], code = alt-allele
    count, 3 = missing): sample.id, variant.id, position, chromosome, allele, genotype/data (dBit2
    [variant, sample, ploidy]) and genotype/@data -- what seqAssocGLMM_SPA reads (R/assoc_single.r:116-144).
    The test and benchmark generator; dosage d is stored as alleles (1, 0) / (1, 1), missing as (3, 3)."""
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    M = packed.shape[0]
    nb = (n_samp + 3) // 4
    # one packed byte (4 dosage codes) -> 16 bits (4 samples x 2 alleles x 2 bits)
    nib = np.array([0b0000, 0b0001, 0b0101, 0b1111], dtype=np.uint16)
    lut = np.zeros(256, dtype=np.uint16)
    for b in range(256):
        lut[b] = sum(int(nib[(b >> (2 * k)) & 3]) << (4 * k) for k in range(4))
    al = lut[packed[:, :nb]].astype("<u2")                      # [M, nb] -> 2 bytes per 4 samples
    rows = al.view(np.uint8).reshape(M, 2 * nb)
    row_bits = n_samp * 4
    if row_bits % 8 == 0:
        raw = np.ascontiguousarray(rows[:, :row_bits // 8]).tobytes()
    else:
        # rows are not whole bytes: through a bit stream
        bits = np.unpackbits(rows, axis=1, bitorder="little")[:, :row_bits].reshape(-1)
        raw = np.packbits(bits, bitorder="little").tobytes()
    w = GdsWriter(fn)
    w.put_attr("FileFormat", "SEQ_ARRAY")
    sid = [f"s{i + 1}" for i in range(n_samp)] if sample_id is None else [str(x) for x in sample_id]
    w.add("sample.id", sid, "none")
    w.add("variant.id", np.arange(1, M + 1), "none")
    w.add("position", np.arange(1, M + 1) * 100, "none")
    w.add("chromosome", [chromosome] * M, "none")
    w.add("allele", ["A,C"] * M, "none")
    w.add("genotype/data", raw, compress, cls="dBit2", dims=(M, n_samp, 2), ra_block=ra_block)
    w.add("genotype/@data", np.ones(M, dtype=np.uint8).tobytes(), "none", cls="dUInt8", dims=(M,))
    w.close()


def write_saige_output(fn: str, ans: Dict[str, Any], sample_id: Sequence[str], compress: str = "LZMA",
                       version: str = "SAIGEgds_amd"):
    """The SAIGE_OUTPUT file of R/assoc_single.r:249-280, same nodes in the same order."""
    w = GdsWriter(fn)
    w.put_attr("FileFormat", "SAIGE_OUTPUT")
    w.put_attr("Version", version)
    w.add("sample.id", [str(s) for s in sample_id], compress)
    for name in ("id", "chr", "pos", "rs.id", "ref", "alt", "AF.alt", "mac", "num", "beta", "SE", "pval",
                 "p.norm", "converged"):
        if name in ans:
            v = ans[name]
            w.add(name, list(v) if isinstance(v, list) else np.asarray(v), compress)
    w.close()
