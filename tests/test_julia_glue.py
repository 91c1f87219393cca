"""The Julia glue cannot be executed here (no Julia toolchain in the container or on the GPU box), so what CAN be checked
statically is: every `ccall` of `enlsip.jl_amd/julia/EnlsipHIP.jl` names an entry point that `include/enlsip_gn.h` declares, with
the declared number of arguments, a Julia type of the right width and kind for every one of them, and the declared return type;
and the two structs that cross the boundary (`Opts`, `Info`) have the header's fields in the header's order.  A prototype that
changes in the header without its ccall breaks this test instead of a maintainer's first run."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
HEADER = (ROOT / "include" / "enlsip_gn.h").read_text()
GLUE = (ROOT / "enlsip.jl_amd" / "julia" / "EnlsipHIP.jl").read_text()

# C parameter type (qualifiers stripped) -> the Julia ccall types that may carry it
ALLOWED = {
    "enlsip_gn_handle": {"Ptr{Cvoid}"},
    "enlsip_gn_handle*": {"Ref{Ptr{Cvoid}}", "Ptr{Ptr{Cvoid}}"},
    "enlsip_gn_opts*": {"Ref{Opts}", "Ptr{Opts}"},
    "enlsip_gn_info*": {"Ref{Info}", "Ptr{Info}"},
    "int": {"Cint"},
    "int*": {"Ref{Cint}", "Ptr{Cint}"},
    "int64_t": {"Int64"},
    "int64_t*": {"Ref{Int64}", "Ptr{Int64}"},
    "uint64_t*": {"Ref{UInt64}", "Ptr{UInt64}"},
    "double": {"Float64"},
    "double*": {"Ref{Float64}", "Ptr{Float64}"},
    "void*": {"Ptr{Cvoid}", "Ptr{UInt8}"},
}
RETURNS = {"int": {"Cint"}, "const char*": {"Cstring", "Ptr{UInt8}"}}


def header_prototypes():
    text = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|const char\*)\s+(enlsip_gn_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = []
        for a in re.sub(r"\s+", " ", m.group(3)).split(","):
            a = a.strip()
            if a == "void" or not a:
                continue
            a = re.sub(r"\bconst\b", "", a).strip()
            ty = re.sub(r"\s*\w+$", "", a).replace(" ", "") if not a.endswith("*") else a.replace(" ", "")
            args.append(ty)
        out[m.group(2)] = (m.group(1), args)
    return out


def glue_ccalls():
    calls = []
    for m in re.finditer(r"ccall\(\(:(enlsip_gn_\w+), LIB\),\s*([\w{}]+),\s*\(", GLUE):
        i = m.end() - 1
        depth, k = 0, i
        while True:
            depth += GLUE[k] == "("
            depth -= GLUE[k] == ")"
            if depth == 0:
                break
            k += 1
        types = [t.strip() for t in re.sub(r"\s+", " ", GLUE[i + 1:k]).split(",") if t.strip()]
        calls.append((m.group(1), m.group(2), types))
    return calls


def test_every_ccall_matches_its_prototype():
    protos = header_prototypes()
    calls = glue_ccalls()
    assert len(calls) >= 25
    for name, ret, types in calls:
        assert name in protos, f"{name}: not declared in include/enlsip_gn.h"
        cret, cargs = protos[name]
        assert ret in RETURNS[cret], f"{name}: returns {cret}, ccall says {ret}"
        assert len(types) == len(cargs), f"{name}: {len(cargs)} parameters in the header, {len(types)} in the ccall"
        for pos, (c, j) in enumerate(zip(cargs, types)):
            assert c in ALLOWED, f"{name}: parameter type {c!r} has no Julia mapping in this test"
            assert j in ALLOWED[c], f"{name}: parameter {pos} is {c} in the header, {j} in the ccall"


def c_struct_fields(name):
    body = re.search(r"typedef struct " + name + r"\s*\{(.*?)\}\s*" + name + r"\s*;", HEADER, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    return [(re.sub(r"\s+", "", t), f) for t, f in re.findall(r"([\w\*\s]+?)\s+(\w+)\s*;", body)]


def julia_struct_fields(name):
    body = re.search(r"^struct " + name + r"\n(.*?)^end", GLUE, flags=re.S | re.M).group(1)
    return [(t, f) for f, t in re.findall(r"^\s*(\w+)::([\w{}]+)", body, flags=re.M)]


def test_structs_that_cross_the_boundary_have_the_headers_layout():
    width = {"int32_t": "Int32", "int64_t": "Int64", "void*": "Ptr{Cvoid}"}
    for cname, jname in (("enlsip_gn_opts", "Opts"), ("enlsip_gn_info", "Info")):
        cf, jf = c_struct_fields(cname), julia_struct_fields(jname)
        assert [f for _, f in cf] == [f for _, f in jf], f"{jname}: field names / order differ from {cname}"
        assert [width[t] for t, _ in cf] == [t for t, _ in jf], f"{jname}: field types differ from {cname}"


def test_factor_selectors_match_the_header():
    for cname, jname in (("ENLSIP_GN_FACTOR_A", "FACTOR_A"), ("ENLSIP_GN_FACTOR_L11", "FACTOR_L11"), ("ENLSIP_GN_FACTOR_J2", "FACTOR_J2")):
        c = int(re.search(cname + r"\s*=\s*(\d+)", HEADER).group(1))
        j = int(re.search(r"const " + jname + r"\s*=\s*Cint\((\d+)\)", GLUE).group(1))
        assert c == j


def test_glue_is_balanced():
    """Cheap syntax net: every block opener of the file has its `end`, brackets are balanced outside strings and comments."""
    code = re.sub(r'"""(.*?)"""', '""', GLUE, flags=re.S)
    lines = [re.sub(r'"(\\.|[^"\\])*"', '""', ln) for ln in code.splitlines()]
    lines = [ln.split("#")[0] for ln in lines]
    text = "\n".join(lines)
    for o, c in ("()", "[]", "{}"):
        assert text.count(o) == text.count(c), f"unbalanced {o}{c}"
    # `end` inside an index expression (a[1:end]) is not a block end
    indexed = len(re.findall(r"\[[^\[\]\n]*\bend\b[^\[\]\n]*\]", text))
    openers = len(re.findall(r"^\s*(?:mutable struct|struct|module|function|for|while|if|begin)\b", text, flags=re.M))
    openers += len(re.findall(r"\bbegin\s*$", text, flags=re.M)) - len(re.findall(r"^\s*begin\s*$", text, flags=re.M))
    ends = len(re.findall(r"\bend\b", text)) - indexed
    assert openers == ends, f"{openers} block openers, {ends} ends"
