"""Julia is not in the build image, so the ccall module cannot be executed here.  The only guard available: every
`ccall((:pmk_xxx, libpmk), Ret, (Arg...), ...)` tuple in julia/PatchMixtureKriging/src/PatchMixtureKriging.jl is parsed
and compared, argument by argument, with the prototype of the same symbol in include/pmk.h (C-type category: 32-bit int,
64-bit int, double, pointer)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "julia", "PatchMixtureKriging", "src", "PatchMixtureKriging.jl")

JULIA = {"Cint": "i32", "Int32": "i32", "Int64": "i64", "Float64": "f64", "Cstring": "ptr", "Cvoid": "void"}


def jl_cat(t):
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")):
        return "ptr"
    return JULIA[t]


def c_cat(t):
    t = re.sub(r"\bconst\b", "", t).strip()
    if "*" in t:
        return "ptr"
    base = t.split()[0] if t.split() else t
    if base in ("int", "int32_t"):
        return "i32"
    if base == "int64_t":
        return "i64"
    if base == "double":
        return "f64"
    if base == "void":
        return "void"
    raise ValueError(t)


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({":
            depth += 1
        if ch in ")}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out]


def header_prototypes():
    txt = open(os.path.join(ROOT, "include", "pmk.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(pmk_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argl = [] if args.strip() in ("", "void") else split_top(args)
        cats = []
        for a in argl:
            a = re.sub(r"\b[A-Za-z_]\w*$", "", a.strip()) if not a.strip().endswith("*") else a     # drop the parameter name
            cats.append(c_cat(a))
        protos[name] = (c_cat(ret), cats)
    return protos


def julia_ccalls():
    txt = open(JL).read()
    calls = []
    for m in re.finditer(r"ccall\(\(:(pmk_[a-z0-9_]+),\s*libpmk\),\s*([A-Za-z0-9{}]+),\s*\(", txt):
        name, ret = m.group(1), m.group(2)
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(txt[i], 0)
            i += 1
        calls.append((name, jl_cat(ret), [jl_cat(a) for a in split_top(txt[m.end():i - 1])], txt[:m.start()].count("\n") + 1))
    # every ccall must carry a literal tuple of argument types (a variable is not valid Julia)
    assert len(calls) == len(re.findall(r"ccall\(\(:pmk_", txt)), "a ccall without a literal argument-type tuple"
    return calls


def test_every_ccall_matches_the_header():
    protos = header_prototypes()
    calls = julia_ccalls()
    assert len(calls) >= 35 and len(protos) >= 50
    seen = set()
    for name, ret, args, line in calls:
        assert name in protos, "%s (line %d) is not declared in include/pmk.h" % (name, line)
        cret, cargs = protos[name]
        assert ret == cret, "%s line %d: return %s vs %s" % (name, line, ret, cret)
        assert args == cargs, "%s line %d:\n  julia %s\n  C     %s" % (name, line, args, cargs)
        seen.add(name)
    # the entry points the two example scripts need are all bound
    for need in ("pmk_bsp_build", "pmk_bsp_assign", "pmk_bsp_arrays", "pmk_fit_batched", "pmk_model_get", "pmk_query_plan",
                 "pmk_query_items", "pmk_query_mix", "pmk_query_fetch", "pmk_kernel_matrix", "pmk_query_mean",
                 "pmk_query_predict_sharded", "pmk_comm_create"):
        assert need in seen, need
