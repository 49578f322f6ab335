#!/usr/bin/env python3
"""usage: scripts/regs.py lib.so [name filter]: VGPRs / spills / LDS of the gfx950 kernels in a built library
(reads the offload bundle inside the .so and the code object's metadata notes; compiles nothing)"""
import re, struct, subprocess, sys, tempfile
data = open(sys.argv[1], "rb").read()
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
magic = b"__CLANG_OFFLOAD_BUNDLE__"
at = data.find(magic)
assert at >= 0, "no uncompressed offload bundle in this file"
n, = struct.unpack_from("<Q", data, at + 24)
p = at + 32
for _ in range(n):
    off, size, tl = struct.unpack_from("<QQQ", data, p); p += 24
    triple = data[p:p + tl].decode(); p += tl
    if "gfx950" not in triple:
        continue
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(data[at + off:at + off + size]); f.flush()
        notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
    cur = {}
    for line in notes.splitlines():
        m = re.match(r"\s+-?\s*\.(\w+):\s+(\S+)", line)
        if not m:
            continue
        k, v = m.groups()
        if k == "group_segment_fixed_size" and "name" in cur:   # a new kernel record starts (fields come sorted by key)
            cur = {}
        cur[k] = v
        if k == "vgpr_spill_count":
            if pat.search(cur.get("name", "")):
                print(f"{cur.get('name'):70s} vgpr {cur.get('vgpr_count'):>4s} spill {v:>3s} sgpr_spill {cur.get('sgpr_spill_count'):>3s} lds {cur.get('group_segment_fixed_size')}")
