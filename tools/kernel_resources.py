#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel in the built library (the code object's own metadata; no GPU, no recompile).
Usage: python tools/kernel_resources.py [substring ...]   -> name, VGPRs, AGPRs, SGPRs, scratch bytes per lane, static LDS, wavefronts per SIMD"""
import os, re, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = open(os.path.join(ROOT, "dapol_amd", "libdapol_hip.so"), "rb").read()
i = so.find(b"__CLANG_OFFLOAD_BUNDLE__")
n = struct.unpack_from("<Q", so, i + 24)[0]
p, co = i + 32, None
for _ in range(n):
    off, size, tl = struct.unpack_from("<QQQ", so, p); p += 24
    triple = so[p:p + tl].decode(); p += tl
    if "gfx950" in triple:
        co = so[i + off:i + off + size]
with tempfile.NamedTemporaryFile(suffix=".co") as f:
    f.write(co); f.flush()
    txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
rows = []
for k in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
    g = lambda key: (re.search(r"\.%s:\s+(\S+)" % key, k) or [None, "?"])[1]
    rows.append((g("name"), int(k.split("\n")[0].strip()), int(g("vgpr_count")), int(g("sgpr_count")), int(g("private_segment_fixed_size")), int(g("group_segment_fixed_size"))))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
want = sys.argv[1:]
print("%-58s %5s %5s %5s %8s %7s %4s" % ("kernel", "vgpr", "agpr", "sgpr", "scratch", "lds", "occ"))
for r, nm in sorted(zip(rows, names), key=lambda x: x[1]):
    nm = re.sub(r"^(void )?dapol::", "", nm.split("(")[0])
    if want and not any(w in nm for w in want):
        continue
    tot = r[2] + r[1] if r[1] else r[2]                # unified register file: 512 per SIMD lane
    occ = min(8, 512 // max(8, (tot + 7) // 8 * 8))
    print("%-58s %5d %5d %5d %8d %7d %4d" % (nm[:58], r[2], r[1], r[3], r[4], r[5], occ))
