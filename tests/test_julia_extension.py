"""Boundary check of the Julia side without Julia: every `ccall((:ocn_..., lib), Ret, (argtypes...), args...)` in
julia/ext/OceananigansHIPShimExt.jl is parsed and compared with the declaration of that symbol in include/ocn_hip.h -- name, arity,
return kind and the kind (pointer / 32-bit integer / 64-bit integer / size / double) of every argument -- and the Julia mirror structs
are compared with the C structs field by field.  (Pattern of the extension: ext/OceananigansMetalExt.jl:11-35 of the reference.)"""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXT = os.path.join(ROOT, "julia", "ext", "OceananigansHIPShimExt.jl")
HDR = os.path.join(ROOT, "include", "ocn_hip.h")

# symbols the hot path cannot do without: the extension must bind each of them
REQUIRED = {
    "ocn_last_error", "ocn_malloc", "ocn_free", "ocn_memcpy_h2d", "ocn_memcpy_d2h", "ocn_sync", "ocn_set_device",
    "ocn_fill_halo_regions", "ocn_compute_momentum_tendencies", "ocn_compute_tracer_tendency", "ocn_rk3_substep", "ocn_ab2_step",
    "ocn_cache_previous_tendencies", "ocn_poisson_create", "ocn_poisson_destroy", "ocn_solve_for_pressure",
    "ocn_pressure_correct_velocities", "ocn_rk3_driver_create", "ocn_rk3_driver_time_step", "ocn_rk3_driver_flush",
    "ocn_rk3_driver_create_distributed", "ocn_rk3_driver_configure", "ocn_model_driver_create", "ocn_model_driver_time_step", "ocn_model_driver_flush",
    "ocn_comm_unique_id", "ocn_comm_init", "ocn_halo_exchange_begin", "ocn_halo_exchange_end", "ocn_dist_poisson_create_global",
    "ocn_dist_poisson_exchange", "ocn_compute_momentum_tendencies_terms", "ocn_compute_tracer_tendency_terms",
}


def c_kind(ctype):
    t = " ".join(ctype.replace("const", " ").split())
    if "*" in t or t.endswith("_t") and t not in ("int32_t", "int64_t", "size_t", "uint8_t"):
        return "ptr"  # data pointers and the opaque handle typedefs (ocn_poisson_t = struct ocn_poisson *)
    return {"int": "i32", "int32_t": "i32", "int64_t": "i64", "long long": "i64", "size_t": "size", "double": "f64", "void": "void"}[t]


def header_declarations():
    src = re.sub(r"/\*.*?\*/", "", open(HDR).read(), flags=re.S)
    out = {}
    for ret, name, args in re.findall(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(ocn_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        kinds = []
        for a in args.split(","):
            a = " ".join(a.split())
            if a == "void":
                continue
            m = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)$", a)  # strip the parameter name
            kinds.append(c_kind(m.group(1).strip()))
        out[name] = (c_kind(ret.strip()), kinds)
    return out


def julia_kind(jt):
    jt = jt.strip()
    if jt.startswith(("Ptr{", "Ref{")) or jt == "Cstring":
        return "ptr"
    return {"Cint": "i32", "Int32": "i32", "Cuint": "i32", "Int64": "i64", "Clonglong": "i64", "Csize_t": "size",
            "Float64": "f64", "Cdouble": "f64", "Cvoid": "void"}[jt]


def strip_julia_comments(src):
    out, in_str, i = [], False, 0
    while i < len(src):
        ch = src[i]
        if in_str:
            out.append(ch)
            if ch == "\\":
                out.append(src[i + 1])
                i += 1
            elif ch == '"':
                in_str = False
        elif ch == '"':
            in_str = True
            out.append(ch)
        elif ch == "#":
            while i < len(src) and src[i] != "\n":
                i += 1
            continue
        else:
            out.append(ch)
        i += 1
    return "".join(out)


def split_top_level(s):
    parts, depth, cur = [], 0, []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append("".join(cur).strip())
            cur = []
        else:
            cur.append(ch)
    if "".join(cur).strip():
        parts.append("".join(cur).strip())
    return parts


def julia_ccalls():
    src = strip_julia_comments(open(EXT).read())
    calls = []
    for m in re.finditer(r"\bccall\(", src):
        depth, j = 1, m.end()
        while depth:
            depth += {"(": 1, ")": -1}.get(src[j], 0)
            j += 1
        parts = split_top_level(src[m.end():j - 1])
        sym = re.match(r"\(\s*:([A-Za-z0-9_]+)\s*,\s*lib\s*\)", parts[0])
        assert sym, f"ccall without a (:symbol, lib) target: {parts[0]}"
        argtypes = parts[2].strip()
        assert argtypes.startswith("(") and argtypes.endswith(")")
        types = split_top_level(argtypes[1:-1])
        calls.append((sym.group(1), parts[1].strip(), types, parts[3:]))
    return calls


def test_every_ccall_matches_the_header():
    decl = header_declarations()
    calls = julia_ccalls()
    assert len(calls) >= 50
    for sym, ret, types, args in calls:
        assert sym in decl, f"{sym}: not declared in include/ocn_hip.h"
        cret, ckinds = decl[sym]
        assert julia_kind(ret) == cret, f"{sym}: return {ret} vs C {cret}"
        assert len(types) == len(ckinds), f"{sym}: {len(types)} argument types in the ccall, {len(ckinds)} parameters in the header"
        assert len(args) == len(types), f"{sym}: {len(args)} argument values for {len(types)} argument types"
        for n, (jt, ck) in enumerate(zip(types, ckinds)):
            assert julia_kind(jt) == ck, f"{sym}: argument {n + 1} is {jt} in the ccall, {ck} in the header"


def test_extension_binds_the_hot_path():
    bound = {c[0] for c in julia_ccalls()}
    missing = sorted(REQUIRED - bound)
    assert not missing, f"julia/ext/OceananigansHIPShimExt.jl lacks bindings for {missing}"


def _c_struct_fields(name):
    src = re.sub(r"/\*.*?\*/", "", open(HDR).read(), flags=re.S)
    body = re.search(r"typedef struct " + name + r"\s*\{(.*?)\}\s*" + name + r"\s*;", src, flags=re.S).group(1)
    fields = []
    for stmt in body.split(";"):
        stmt = " ".join(stmt.split())
        if not stmt:
            continue
        m = re.match(r"((?:const )?[A-Za-z0-9_]+(?: \*)?)\s*(.*)$", stmt)
        ctype, names = m.group(1), m.group(2)
        for nm in names.split(","):
            nm = nm.strip()
            ptr = "*" in ctype or nm.startswith("*")
            nm = nm.lstrip("* ")
            count = None
            am = re.match(r"([A-Za-z0-9_]+)\[(.*)\]$", nm)  # fixed arrays: name[expr] with the header's #define constants
            if am:
                nm = am.group(1)
                expr = am.group(2)
                for k, v in re.findall(r"#define\s+([A-Z0-9_]+)\s+(\d+)", src):
                    expr = re.sub(r"\b" + k + r"\b", v, expr)
                count = int(eval(expr, {"__builtins__": {}}))
            base = ctype.replace("const ", "").replace("*", "").strip()
            kind = "ptr" if ptr else ("struct:" + base if base.startswith("ocn_") and not base.endswith("_t") else c_kind(ctype))
            fields.append((nm, kind if count is None else f"{count}x{kind}"))
    return fields


def _julia_struct_fields(name):
    src = strip_julia_comments(open(EXT).read())
    body = re.search(r"\bstruct " + name + r"\b(.*?)\bend\b", src, flags=re.S).group(1)
    out = []
    for n, t in re.findall(r"([A-Za-z_][A-Za-z0-9_]*)::([A-Za-z0-9_{}, ]+?)(?=;|\n|$)", body):
        t = t.strip()
        m = re.match(r"NTuple\{(\d+), *(.*)\}$", t)
        if m:
            out.append((n, f"{m.group(1)}x{_julia_field_kind(m.group(2))}"))
        else:
            out.append((n, _julia_field_kind(t)))
    return out


JULIA_STRUCTS = {"OcnGrid": "ocn_grid", "OcnModelTerms": "ocn_model_terms", "OcnBc": "ocn_bc", "OcnFieldBcs": "ocn_field_bcs",
                 "OcnModelDriverDesc": "ocn_model_driver_desc"}


def _julia_field_kind(t):
    return "struct:" + JULIA_STRUCTS[t] if t in JULIA_STRUCTS else julia_kind(t)


@pytest.mark.parametrize("jname,cname", sorted(JULIA_STRUCTS.items()))
def test_julia_mirror_structs_match_the_c_structs(cname, jname):
    """field names, order, scalar / pointer kinds, nested structs and fixed-array lengths (NTuple{N, T} <-> T name[N])"""
    assert _julia_struct_fields(jname) == _c_struct_fields(cname)


def test_extension_follows_the_metal_extension_pattern():
    src = open(EXT).read()
    for needle in ("module OceananigansHIPShimExt", "const HIPGPU = GPU{HIPShim}", "architecture(::HIPShimArray)",
                   "on_architecture(::HIPGPU, a::Array", "on_architecture(::CPU, d::HIPShimArray", "end # module"):
        assert needle in src, needle
