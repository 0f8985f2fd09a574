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


def _lzma_ra(raw: bytes) -> bytes:
    blocks, index = [], b""
    for off in range(0, max(len(raw), 1), _RA_BLOCK):
        piece = raw[off:off + _RA_BLOCK]
        xz = lzma.compress(piece, format=lzma.FORMAT_XZ, check=lzma.CHECK_CRC32, preset=6)
        if len(xz) >= 1 << 24:
            raise ValueError("LZMA_RA block does not fit its 24-bit size field")
        blocks.append(xz)
        index += len(xz).to_bytes(3, "little") + struct.pack("<I", len(piece))
    body = b"".join(blocks)
    return b"XZ_RA\x10\x11\xff" + struct.pack("<I", len(blocks)) + _u48(len(body)) + body + index, len(blocks)


class GdsWriter:
    """Flat GDS file: array nodes under the root, string attributes on the root."""

    def __init__(self, path: str):
        self.path = path
        self._nodes: List[Tuple[str, bytes, bytes]] = []     # name, node stream (data id patched at close), data stream
        self._root_attrs: List[Tuple[str, Optional[str]]] = []

    def put_attr(self, name: str, value: str):
        self._root_attrs.append((name, value))

    def add(self, name: str, values: Any, compress: str = "LZMA_RA"):
        """add.gdsn(root, name, values, compress=, closezip=TRUE)"""
        attrs: List[Tuple[str, Optional[str]]] = []
        if isinstance(values, (list, tuple)) and (len(values) == 0 or isinstance(values[0], str)):
            cls, n = "dStr8", len(values)
            raw = b"".join(_varint(len(b)) + b for b in (str(v).encode("utf-8") for v in values))
        else:
            a = np.asarray(values)
            n = a.size
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
        if compress and compress.lower() != "none":
            # every compressed flavour is written as LZMA_ra (the coder whose stream layout is known from
            # the reference's files); only the file size depends on res.compress
            data, nblk = _lzma_ra(raw)
            body += _R_PIPE + b"\x07LZMA_ra" + _R_PIPE_SIZE + struct.pack("<qq", len(raw), len(data) - 7 * nblk)
            body += _R_PIPE_LEVEL + _R_PIPE_BLOCK
            nprop += 4
        else:
            data = raw
        body += _R_DCNT + struct.pack("<H", 1) + _R_DIM + b"\x04" + struct.pack("<i", n)
        body += _R_DATA + b"\xff\xff\xff\xff"               # data stream id, patched in close()
        body += _attrs(attrs)
        nprop += 4 + (1 if attrs else 0)
        self._nodes.append((name, _node_stream(nprop, body), data))

    def close(self):
        streams: Dict[int, bytes] = {}
        entries = b""
        next_id = 2
        for name, node, data in self._nodes:
            nid, did = next_id, next_id + 1
            next_id += 2
            k = node.index(_R_DATA + b"\xff\xff\xff\xff")
            streams[nid] = node[:k + len(_R_DATA)] + struct.pack("<I", did) + node[k + len(_R_DATA) + 4:]
            streams[did] = data
            nb = name.encode()
            rec = _R_ENTRY_ID + struct.pack("<I", nid) + _R_ENTRY_FLAG + struct.pack("<I", 0) + _R_ENTRY_NAME + bytes([len(nb)]) + nb
            entries += _u48(8 + len(rec)) + struct.pack("<H", 3) + rec
        body = _R_DIRCNT + struct.pack("<I", len(self._nodes))
        nprop = 1
        if self._nodes:
            body += _R_DIRLIST + _u48(6 + len(entries)) + entries
            nprop += 1
        body += _attrs(self._root_attrs)
        nprop += 1 + (1 if self._root_attrs else 0)
        streams[1] = _node_stream(nprop, body)
        with open(self.path, "wb") as f:
            f.write(_MAGIC + b"\x00\x01" + struct.pack("<I", 1))
            for sid in sorted(streams):
                s = streams[sid]
                f.write(_u48((22 + len(s)) | (1 << 47)) + _u48(0) + struct.pack("<I", sid) + _u48(len(s)) + s)


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
