#!/usr/bin/env python3
"""The API surface of the reference's Julia module that the drop-in module must reproduce, as DATA (names only):
struct names with their field names in order, the export list, and the names the example scripts call qualified.
Read from /root/reference (available in the build container only); the result is committed as
tests/golden/julia_api_surface.json and tests/test_julia_binding.py checks julia/PatchMixtureKriging against it.

    python tests/golden/make_julia_surface.py
"""
import json
import os
import re

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def structs(path, lo, hi):
    lines = open(os.path.join(REF, path), encoding="utf-8").read().split("\n")[lo - 1:hi]
    out, cur = [], None
    for ln in lines:
        code = ln.split("#")[0].strip()
        m = re.match(r"(?:mutable\s+)?struct\s+([A-Za-z_]\w*)", code)
        if m:
            cur = {"name": m.group(1), "mutable": code.startswith("mutable"), "fields": [], "at": "%s:%d" % (path, lo + lines.index(ln))}
            continue
        if cur is not None:
            if code == "end" or code.startswith("function ") or re.match(r"[A-Za-z_]\w*\(", code):
                if code == "end" or cur["fields"]:
                    out.append(cur)
                    cur = None
                continue
            f = re.match(r"([^\s:=(]+)\s*::", code)
            if f:
                cur["fields"].append(f.group(1))
    return out


def main():
    surface = {"source": "RoyCCWang/PatchMixtureKriging v0.1.4 (names only)", "structs": [], "exports": [], "qualified": {}}
    for path, lo, hi in (("src/misc/declarations.jl", 18, 111), ("src/misc/declarations.jl", 226, 231),
                         ("src/RKHS/mixtureGP.jl", 5, 52), ("src/patchwork/partition.jl", 3, 29)):
        surface["structs"] += structs(path, lo, hi)
    txt = open(os.path.join(REF, "src/PatchMixtureKriging.jl"), encoding="utf-8").read()
    exp = txt[txt.index("export"):txt.index("end # module")]
    for ln in exp.split("\n"):
        code = ln.split("#")[0]
        surface["exports"] += [t for t in re.findall(r"[A-Za-z_]\w*!?", code) if t != "export"]
    for ex in ("examples/mixGP.jl", "examples/IBB1D.jl", "examples/patchGP_partitioning.jl", "examples/splines1D.jl"):
        t = "\n".join(ln.split("#")[0] for ln in open(os.path.join(REF, ex), encoding="utf-8").read().split("\n"))   # code, not comments
        names = sorted(set(re.findall(r"PatchMixtureKriging\.([A-Za-z_]\w*!?)", t)) - {"jl"})
        surface["qualified"][ex] = names
    json.dump(surface, open(os.path.join(HERE, "julia_api_surface.json"), "w", encoding="utf-8"), indent=1, ensure_ascii=False)
    print(len(surface["structs"]), "structs,", len(surface["exports"]), "exports")


if __name__ == "__main__":
    main()
