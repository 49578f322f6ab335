"""Guards on the SHIPPED gfx950 code object (CPU test: reads libhf.so, launches nothing).

Round 3 measured a variant that made the traversal of a batch a real (noinline) call; its parity run aborted and its
full-size launch ended in `Memory access fault ... on address (nil)` (profiles/r03_ab/r03_d/call.log).  The re-built
variant (profiles/r04_fault/) shows what a call does to these kernels: the by-value kernel argument, the ray state and
the hit record go to a 788-820 byte private frame per lane (the product: 8-16 bytes of spills), reached by the callee
through FLAT loads of  (private aperture, offset) -- at 5120 resident waves that is 268 MB of scratch per dispatch where
the product needs 5 MB, i.e. the dispatch depends on the runtime's large-scratch path and every stack access faults at
the aperture's base when that path has not mapped the memory.  The product's safety therefore rests on
everything being inlined; these tests make that a checked property of the library that ships instead of a habit:
  * no call instruction (s_swappc_b64 / s_setpc_b64) in any kernel,
  * no dynamic stack, private segment <= 64 bytes per lane in every kernel,
  * no flat_* memory instruction (every access names its address space: global / scratch / ds),
  * the spill counts of the hot kernels stay where profiles/r03_final_registers.txt recorded them (+ slack).
"""
import os
import re
import struct
import subprocess
import tempfile

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"
MAX_PRIVATE_BYTES = 64


def _code_object():
    import hf_amd
    path = hf_amd.build.LIB_PATH
    assert os.path.exists(path), "libhf.so not built (run python __graft_entry__.py)"
    data = open(path, "rb").read()
    at = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
    assert at >= 0, "no uncompressed offload bundle in libhf.so"
    n, = struct.unpack_from("<Q", data, at + 24)
    p = at + 32
    found = []
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", data, p); p += 24
        triple = data[p:p + tl].decode(); p += tl
        if "amdgcn" in triple:
            found.append((triple, data[at + off:at + off + size]))
    assert len(found) == 1 and "gfx950" in found[0][0], f"expected exactly one device code object, for gfx950: {[t for t, _ in found]}"
    return found[0][1]


@pytest.fixture(scope="module")
def code_object():
    if not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("llvm-objdump not available")
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(_code_object())
    yield f.name
    os.unlink(f.name)


def _kernels(co):
    """per-kernel metadata records of the code object's notes"""
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    recs, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s+(-\s+)?\.(\w+):\s+(\S+)", line)
        if not m:
            continue
        first, k, v = m.groups()
        if k == "agpr_count" or (first and k in ("args",)):   # first key of a kernel record (keys come sorted; .args first when present)
            pass
        if k in ("args", "agpr_count") and (cur is None or k in cur or "name" in cur):
            cur = {}
            recs.append(cur)
        if cur is not None and k not in ("offset", "size", "value_kind", "address_space", "actual_access", "is_const"):
            cur.setdefault(k, v)
    return [r for r in recs if "name" in r and "vgpr_count" in r]


def test_no_calls_and_no_flat_accesses(code_object):
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", code_object], capture_output=True, text=True, check=True).stdout
    assert dis.count("s_endpgm") >= 20, "disassembly looks empty"
    for ins in ("s_swappc_b64", "s_setpc_b64", "s_call_b64"):
        assert len(re.findall(rf"\b{ins}\b", dis)) == 0, f"{ins} in the shipped kernels: a device function was not inlined"
    flat = re.findall(r"\bflat_(?:load|store|atomic)\w*", dis)
    assert not flat, f"flat memory instructions in the shipped kernels: {sorted(set(flat))}"
    assert "v_mfma" not in dis   # nothing on this path is a contraction


def test_private_segment_and_stack_of_every_kernel(code_object):
    ks = _kernels(code_object)
    names = [k["name"] for k in ks]
    assert len(ks) >= 20 and any("hf_trace_kernel" in n for n in names) and any("hf_adjoint_kernel" in n for n in names), names
    for k in ks:
        assert k.get("uses_dynamic_stack", "false") == "false", f"{k['name']}: dynamic stack"
        priv = int(k["private_segment_fixed_size"])
        assert priv <= MAX_PRIVATE_BYTES, f"{k['name']}: {priv} bytes of private segment per lane (a by-reference call or a spill storm)"


def test_spills_of_the_hot_kernels(code_object):
    """register allocation is the fragile part of the traversal kernels: a source change that pushes the
    hot kernels into spilling shows up here before it shows up in a profile"""
    ks = {k["name"]: k for k in _kernels(code_object)}
    bounds = {"hf_trace_kernelILi0ELb0": 8, "hf_trace_kernelILi1ELb0": 8, "hf_trace_kernelILi2ELb0": 8, "hf_trace_kernelILi2ELb1": 16,
              "hf_adjoint_kernelILb0": 0, "hf_si_kernel": 0}
    for frag, cap in bounds.items():
        hit = [k for n, k in ks.items() if frag in n]
        assert hit, f"kernel {frag} not found in {sorted(ks)}"
        for k in hit:
            assert int(k["vgpr_spill_count"]) <= cap, f"{k['name']}: {k['vgpr_spill_count']} spilled VGPRs (cap {cap})"
            assert int(k["vgpr_count"]) <= 128, k["name"]
