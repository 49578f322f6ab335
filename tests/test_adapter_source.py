"""Row f4: the Mitsuba adapter plugin ships as source (adapters/mitsuba3/heightfield.cpp); Mitsuba 3.3 + Dr.Jit
0.4.2 cannot be built here, so what is checked is its consistency with the boundary it binds: every hf_* entry
point it calls is declared in include/hf.h and exported by libhf.so, it is called with the declared number of
arguments, the struct fields it fills exist, and the file has no placeholder left in it."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "adapters", "mitsuba3", "heightfield.cpp")


def _strip_comments(txt):
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return re.sub(r"//[^\n]*", "", txt)


def _header_decls():
    txt = _strip_comments(open(os.path.join(ROOT, "include", "hf.h")).read())
    decls = {}
    for m in re.finditer(r"\b(hf_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        args = m.group(2).strip()
        decls[m.group(1)] = 0 if args in ("", "void") else len(_split_args(args))
    return decls


def _split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{<":
            depth += 1
        elif ch in ")]}>":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def _calls(txt):
    """(name, n_args) of every hf_*( ... ) call in the adapter"""
    res = []
    for m in re.finditer(r"\b(hf_[a-z_0-9]+)\s*\(", txt):
        name, i, depth = m.group(1), m.end(), 1
        j = i
        while depth and j < len(txt):
            depth += txt[j] in "([{"
            depth -= txt[j] in ")]}"
            j += 1
        args = txt[i:j - 1].strip()
        res.append((name, 0 if not args else len(_split_args(args))))
    return res


def test_adapter_calls_match_the_c_abi():
    import hf_amd
    src = _strip_comments(open(SRC).read())
    decls = _header_decls()
    lib = C.CDLL(hf_amd.build.LIB_PATH)
    calls = [(n, k) for n, k in _calls(src) if n not in ("hf_check",)]
    assert calls, "no hf_* calls found"
    used = {n for n, _ in calls}
    # the plugin must cover the whole hot path of SURVEY 8a/8b
    for need in ("hf_create", "hf_destroy", "hf_set_heights_host", "hf_set_transform", "hf_bbox",
                 "hf_ray_intersect_preliminary", "hf_ray_test", "hf_compute_surface_interaction", "hf_adjoint",
                 "hf_ray_intersect_preliminary_packet", "hf_ray_test_packet", "hf_last_error_string"):
        assert need in used, f"adapter does not call {need}"
    for name, nargs in calls:
        assert name in decls, f"adapter calls {name}, which include/hf.h does not declare"
        assert hasattr(lib, name), f"{name} not exported by libhf.so"
        assert nargs == decls[name], f"{name}: called with {nargs} arguments, declared with {decls[name]}"


def test_adapter_struct_fields_and_registration():
    src = open(SRC).read()
    hdr = _strip_comments(open(os.path.join(ROOT, "include", "hf.h")).read())
    for field in re.findall(r"\bdesc\.([a-z_]+)", src):
        assert re.search(rf"\b{field}\b", hdr[hdr.index("typedef struct hf_desc"):hdr.index("} hf_desc_t")]), field
    # plugin / class registration of the reference (class.h:195-211), both forms of the ray methods, the AD node
    for token in ("MI_IMPLEMENT_CLASS_VARIANT(Heightfield, Shape)", "MI_EXPORT_PLUGIN(Heightfield",
                  "MI_SHAPE_DEFINE_RAY_INTERSECT_METHODS()", "MI_DECLARE_CLASS()", "dr::CustomOp<",
                  "void traverse(TraversalCallback", "void parameters_changed(", "bool parameters_grad_enabled()",
                  "ScalarBoundingBox3f bbox()"):
        assert token in src, token
    # complete source: no ellipsis placeholders, no TODO markers
    code = _strip_comments(src)
    assert "..." not in code.replace("Input...", "").replace("typename...", "").replace("Args...", "")
    assert "TODO" not in src and "FIXME" not in src
    frag = open(os.path.join(ROOT, "adapters", "mitsuba3", "CMakeLists.fragment.txt")).read()
    assert "add_plugin(heightfield heightfield.cpp)" in frag and "libhf.so" in frag
