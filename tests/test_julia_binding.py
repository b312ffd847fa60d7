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


# ---------------------------------------------------------------------------------------------------------------------
# The API surface of the reference module (tests/golden/julia_api_surface.json: struct names and field order, the export
# list, the names its example scripts call qualified -- names only, written by tests/golden/make_julia_surface.py) against
# the drop-in module, and a block-balance check of the file: what can be said about Julia source without a Julia.
import json


def _surface():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "julia_api_surface.json"), encoding="utf-8"))


def _julia_code():
    """the module's text with doc strings, strings, character literals and comments blanked out"""
    txt = open(JL, encoding="utf-8").read()
    txt = re.sub(r'"""(?:.|\n)*?"""', lambda m: "\n" * m.group(0).count("\n"), txt)
    txt = re.sub(r'"(?:\\.|[^"\\\n])*"', '""', txt)
    txt = re.sub(r"'(?:\\.|[^'\\\n])'", "' '", txt)
    return "\n".join(ln.split("#")[0] for ln in txt.split("\n"))


def _julia_structs():
    out, cur = {}, None
    for ln in _julia_code().split("\n"):
        code = ln.strip()
        m = re.match(r"(?:mutable\s+)?struct\s+([^\W\d]\w*)", code)
        if m:
            cur = m.group(1)
            out[cur] = []
            code = code[m.end():]
            code = code.split(";", 1)[1] if ";" in code else ""       # one-line structs: `struct A; x; end`
        if cur is None:
            continue
        for part in code.split(";"):
            part = part.strip()
            if part == "end" or part.startswith("end"):
                cur = None
                break
            f = re.match(r"([^\s:=({]+)\s*::", part)
            if f and cur is not None:
                out[cur].append(f.group(1))
    return out


def test_struct_names_and_field_order_follow_the_reference():
    ref, mine = _surface()["structs"], _julia_structs()
    # closure-carrying / latently broken types of the reference stay out
    out_of_scope = {"AdaptiveModulatedSqExpKernelType", "BrownianBridge20CompactDomain", "BrownianBridgeCompactDomain",
                    "DoubleProductKernelType", "KRWarpKernelType", "ModulatedMultivariateSqExpKernelType"}
    for s in ref:
        if s["name"] in out_of_scope:
            continue
        assert s["name"] in mine, "struct %s (%s) is missing" % (s["name"], s["at"])
        got = mine[s["name"]]
        # the reference's fields first, in its order (the drop-in may append its own: device handles)
        assert got[:len(s["fields"])] == s["fields"], "%s: fields %s vs reference %s (%s)" % (s["name"], got, s["fields"], s["at"])


def test_exports_and_the_names_the_examples_use_exist():
    sur, code = _surface(), _julia_code()
    m = re.search(r"^export\s+((?:.|\n)*?)\n\s*\n", code, flags=re.M)
    exported = set(re.findall(r"[^\W\d][\w]*!?", m.group(1)))
    assert set(sur["exports"]) <= exported, sorted(set(sur["exports"]) - exported)
    defined = set(re.findall(r"^\s*function\s+([^\W\d]\w*!?)", code, flags=re.M))
    defined |= set(re.findall(r"^([^\W\d]\w*!?)\(.*\)(?:\s*where\s+[^=]*)?\s*=(?!=)", code, flags=re.M))      # short-form methods
    defined |= set(re.findall(r"^(?:mutable\s+)?struct\s+([^\W\d]\w*)", code, flags=re.M))
    defined |= set(re.findall(r"^const\s+([^\W\d]\w*)", code, flags=re.M))
    for ex, names in sur["qualified"].items():
        for n in names:
            assert n in defined, "%s calls PatchMixtureKriging.%s, which the module does not define" % (ex, n)
    for n in sur["exports"]:
        assert n in defined, "exported name %s is not defined" % n


def test_blocks_balance():
    code = _julia_code()
    openers = ("function", "struct", "if", "for", "while", "let", "try", "begin", "do", "module", "quote", "macro")
    depth_sq = depth_par = 0
    opened = closed = 0
    prev_word = ""
    for m in re.finditer(r"[\[\]()]|[^\W\d]\w*", code):
        tok = m.group(0)
        if tok == "[":
            depth_sq += 1
        elif tok == "]":
            depth_sq -= 1
        elif tok == "(":
            depth_par += 1
        elif tok == ")":
            depth_par -= 1
        elif tok == "end":
            if depth_sq == 0:                      # inside [...] `end` is the last index
                closed += 1
        elif tok == "type" and prev_word in ("abstract", "primitive") and depth_sq == 0 and depth_par == 0:
            opened += 1                            # `abstract type X end`
        elif tok in openers and depth_sq == 0 and depth_par == 0:
            opened += 1                            # (`mutable struct` opens one block: counted at `struct`)
        assert depth_sq >= 0 and depth_par >= 0, "unbalanced bracket near offset %d" % m.start()
        if re.match(r"[^\W\d]", tok):
            prev_word = tok
    assert depth_sq == 0 and depth_par == 0
    assert opened == closed, "%d block openers against %d `end`s" % (opened, closed)
