"""The Julia binding cannot run here (no Julia toolchain).  What CAN be checked without one: every ccall in
julia/CFMMRouterAMD.jl names a function include/cfmm_amd.h declares, with the same number of arguments and
argument / return types that map onto the C declaration (Ptr{Float64} <-> double*, Int32 <-> int32_t, ...), and the
RouteInfo struct mirrors cfmm_route_info field by field."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C2J = {   # C parameter type (const and spaces stripped) -> acceptable Julia ccall types
    "cfmm_ctx*": {"Ptr{Cvoid}"},
    "cfmm_ctx**": {"Ref{Ptr{Cvoid}}", "Ptr{Ptr{Cvoid}}"},
    "int32_t": {"Int32", "Cint"},
    "int": {"Cint", "Int32"},
    "int64_t": {"Int64"},
    "uint64_t": {"UInt64"},
    "double": {"Float64", "Cdouble"},
    "double*": {"Ptr{Float64}", "Ref{Float64}"},
    "int32_t*": {"Ptr{Int32}", "Ref{Int32}"},
    "int64_t*": {"Ptr{Int64}", "Ref{Int64}"},
    "char*": {"Cstring", "Ptr{UInt8}"},
    "unsignedchar*": {"Ptr{UInt8}"},
    "void*": {"Ptr{Cvoid}"},
    "void": {"Cvoid"},
    "cfmm_route_info*": {"Ref{RouteInfo}", "Ptr{RouteInfo}"},
    "cfmm_polish_info*": {"Ref{PolishInfo}", "Ptr{PolishInfo}"},
}


def c_declarations():
    text = open(os.path.join(ROOT, "include", "cfmm_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    decls = {}
    for m in re.finditer(r"\b(int|void|const char\*|int32_t|int64_t)\s+(cfmm_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        plist = []
        for p in [q.strip() for q in params.replace("\n", " ").split(",") if q.strip() and q.strip() != "void"]:
            p = re.sub(r"\bconst\b", "", p).strip()
            p = re.sub(r"^(.*?)(\w+)\s*\[[^\]]*\]$", r"\1* \2", p)      # `unsigned char id[N]` is `unsigned char* id`
            mm = re.match(r"^(.*?)(\w+)?$", p)            # strip the parameter name
            ty = mm.group(1).strip() if mm.group(1).strip() else p
            ty = ty.replace(" ", "")
            plist.append(ty)
        decls[name] = (re.sub(r"\bconst\b", "", ret).replace(" ", ""), plist)
    return decls


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "{(":
            depth += 1
        elif ch in "})":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def julia_ccalls():
    text = open(os.path.join(ROOT, "julia", "CFMMRouterAMD.jl")).read()
    text = re.sub(r"#[^\n]*", "", text)
    calls = []
    for m in re.finditer(r"ccall\(\(:(\w+),\s*LIB\),\s*(\w+),\s*\(", text):
        i = m.end()
        depth, j = 1, i
        while depth:
            depth += {"(": 1, ")": -1}.get(text[j], 0)
            j += 1
        calls.append((m.group(1), m.group(2), split_top(text[i:j - 1])))
    return calls


def test_every_ccall_matches_the_header():
    decls = c_declarations()
    calls = julia_ccalls()
    assert len(calls) >= 15
    for name, ret, args in calls:
        assert name in decls, f"{name} is not declared in include/cfmm_amd.h"
        cret, cargs = decls[name]
        assert ret in C2J[cret], f"{name}: return {ret} vs C {cret}"
        assert len(args) == len(cargs), f"{name}: {len(args)} Julia arguments vs {len(cargs)} in the header"
        for k, (ja, ca) in enumerate(zip(args, cargs)):
            assert ca in C2J, f"{name}: unmapped C type {ca}"
            assert ja in C2J[ca], f"{name} argument {k}: Julia {ja} vs C {ca}"


import pytest


@pytest.mark.parametrize("cname,jname", [("cfmm_route_info", "RouteInfo"), ("cfmm_polish_info", "PolishInfo")])
def test_info_structs_mirror_the_header(cname, jname):
    h = open(os.path.join(ROOT, "include", "cfmm_amd.h")).read()
    body = re.search(r"typedef struct[^{]*\{([^}]*)\}\s*" + cname, h, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", " ", body, flags=re.S)
    cfields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ty, names = decl.split(None, 1)
        for nm in names.split(","):
            cfields.append((nm.strip(), ty))
    j = open(os.path.join(ROOT, "julia", "CFMMRouterAMD.jl")).read()
    jbody = re.search(r"struct " + jname + r"\n(.*?)\nend", j, flags=re.S).group(1)
    jfields = [tuple(x.strip() for x in line.split("#")[0].split("::")) for line in jbody.splitlines() if "::" in line]
    tmap = {"double": "Float64", "int32_t": "Int32", "int64_t": "Int64", "int": "Int32"}
    assert [(n, tmap[t]) for n, t in cfields] == jfields
