#!/usr/bin/env python3
"""Registers, scratch and static LDS of the kernels in libsaigehip.so (from the gfx950 code object's metadata).
    python tools/kernel_regs.py [substring ...]
"""
import os, re, struct, subprocess, sys, tempfile
so = os.path.join(os.path.dirname(__file__), "..", "saigegds_amd", "libsaigehip.so")
d = open(so, "rb").read()
i = d.find(b"__CLANG_OFFLOAD_BUNDLE__")
n = struct.unpack_from("<Q", d, i + 24)[0]
p = i + 32
co = None
for _ in range(n):
    off, size, tl = struct.unpack_from("<QQQ", d, p); p += 24
    t = d[p:p + tl].decode(); p += tl
    if "gfx950" in t: co = d[i + off:i + off + size]
with tempfile.NamedTemporaryFile(suffix=".co") as f:
    f.write(co); f.flush()
    notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
def demangle(names):
    try: return subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    except FileNotFoundError: return names
rows = []
for e in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, e).group(1))
    rows.append((re.search(r"\.name:\s+(\S+)", e).group(1), g("vgpr_count"), int(e.split("\n")[0].strip()), g("sgpr_count"),
                 g("private_segment_fixed_size"), g("group_segment_fixed_size")))
dn = demangle([r[0] for r in rows])
print("%-70s %5s %5s %5s %8s %8s" % ("kernel", "vgpr", "agpr", "sgpr", "scratch", "lds"))
for r, name in zip(rows, dn):
    name = name.replace("void ", "").split("(")[0]
    if len(sys.argv) > 1 and not any(s in name for s in sys.argv[1:]): continue
    print("%-70s %5d %5d %5d %8d %8d" % (name[:70], r[1], r[2], r[3], r[4], r[5]))
