"""The Julia binding cannot run here (no Julia toolchain).  What CAN be checked without one: every ccall in
julia/src/CFMMRouterAMD.jl names a function include/cfmm_amd.h declares, with the same number of arguments and
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
    text = open(os.path.join(ROOT, "julia", "src", "CFMMRouterAMD.jl")).read()
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
    j = open(os.path.join(ROOT, "julia", "src", "CFMMRouterAMD.jl")).read()
    jbody = re.search(r"struct " + jname + r"\n(.*?)\nend", j, flags=re.S).group(1)
    jfields = [tuple(x.strip() for x in line.split("#")[0].split("::")) for line in jbody.splitlines() if "::" in line]
    tmap = {"double": "Float64", "int32_t": "Int32", "int64_t": "Int64", "int": "Int32"}
    assert [(n, tmap[t]) for n, t in cfields] == jfields


def test_julia_package_and_its_test_suite_are_consistent():
    """julia/ is a package (Project.toml + src/ + test/runtests.jl) a maintainer with Julia and an MI355X
    runs with ONE command; without Julia, what can be checked: the test file only calls verbs the module defines (with the
    keyword arguments it passes), reads only fields RouteInfo has, mirrors the reference's test files it says it mirrors, and
    the project file names the module and its two dependencies."""
    proj = open(os.path.join(ROOT, "julia", "Project.toml")).read()
    assert 'name = "CFMMRouterAMD"' in proj and "CFMMRouter =" in proj and "LBFGSB =" in proj and "[targets]" in proj
    src = open(os.path.join(ROOT, "julia", "src", "CFMMRouterAMD.jl")).read()
    tst = open(os.path.join(ROOT, "julia", "test", "runtests.jl")).read()
    code = re.sub(r"#[^\n]*", "", tst)
    assert "using CFMMRouterAMD" in code and "using CFMMRouter" in code and "@testset" in code
    defined = set(re.findall(r"^function (\w+!?)\(", src, flags=re.M)) | set(re.findall(r"^(\w+!?)\(r::AMDRouter", src, flags=re.M))
    exported = set(re.search(r"^export (.*)$", src, flags=re.M).group(1).replace(" ", "").split(","))
    imported = set(re.search(r"^import CFMMRouter: (.*)$", src, flags=re.M).group(1).replace(" ", "").split(","))
    # verbs the test applies to an AMDRouter
    for verb in ("AMDRouter", "find_arb!", "route!", "route_native!", "netflows", "update_reserves!"):
        assert verb in code, verb
        assert verb in defined, f"{verb} is used by julia/test/runtests.jl but not defined in the module"
        assert verb in exported or verb in imported, f"{verb} is neither exported nor an extension of a CFMMRouter verb"
    for qualified in set(re.findall(r"CFMMRouterAMD\.(\w+!?)", code)):
        assert qualified in defined, qualified
    # keyword arguments passed to the module's verbs exist there
    assert re.search(r"function netflows\(r::AMDRouter; exact::Bool=true\)", src) and "exact=false" in code
    assert re.search(r"function netflows!\(ψ, r::AMDRouter; exact::Bool=true\)", src)
    jbody = re.search(r"struct RouteInfo\n(.*?)\nend", src, flags=re.S).group(1)
    fields = {line.split("::")[0].strip() for line in jbody.splitlines() if "::" in line}
    for f in set(re.findall(r"\binfo\.(\w+)", code)):
        assert f in fields, f
    # the scenarios it says it mirrors: the UniV3 fixture and its seven prices per fee, the README market, both objectives
    for needle in ("[30.0, 20, 10, 5]", "[1.0, 2.0, 1.5, 0.0]", "16.0, 14.0, 25.0, 7.5, 4.0, 35.0", "[1e6, 1e6]", "[1e3, 2e3]",
                   "BasketLiquidation(1,", "LinearNonnegative(", "flows .== netflows(r)"):
        assert needle in tst, needle
    # and the one command is documented where a maintainer looks
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "julia --project=julia -e 'using Pkg; Pkg.test()'" in integ
