#!/usr/bin/env python3
"""Conformance of tests/fortran_scaffold/scaffold_modules.f90 with the reference's declarations (build container only).

The same-name drop-in modules (dmft-lanc-ed_amd/fortran/dropin/) are compiled against a scaffold that re-declares the host
variables, types and interfaces they USE.  This script parses the declarations of exactly those names in the reference
(/root/reference: ED_INPUT_VARS.f90, ED_VARS_GLOBAL.f90, ED_HAMILTONIAN_COMMON.f90 -- read as text, nothing is copied or
compiled) and in the scaffold, and fails when type, kind, rank or the allocatable/pointer attribute differ, when a derived
type lacks a component the drop-ins read, or when the dd_sparse_HxV interface has another argument list.

    python scripts/check_scaffold_conformance.py [--reference /root/reference]      # exit 0 = conformant
"""
import argparse
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCAFFOLD = os.path.join(ROOT, "tests", "fortran_scaffold", "scaffold_modules.f90")

# host names the drop-in modules read, per reference file
NAMES = {
    "ED_INPUT_VARS.f90": ["Norb", "Nspin", "Nbath", "Nph", "Uloc", "Ust", "Jh", "Jx", "Jp", "xmu", "g_ph", "w0_ph", "HFmode",
                          "ed_sparse_H", "ed_total_ud", "bath_type"],
    "ED_VARS_GLOBAL.f90": ["Ns", "Nsectors", "Ns_orb", "Ns_ud", "DimPh", "impHloc", "getDim", "dmft_bath", "spH0d", "spH0nd",
                           "spH0ups", "spH0dws", "spH0_ph", "spH0e_eph", "spH0ph_eph", "spHtimesV_p", "Jhflag", "MpiComm",
                           "MpiComm_Global", "MpiStatus", "MpiRank", "MpiSize", "H_basis"],
    "ED_HAMILTONIAN_COMMON.f90": ["Dim", "DimUp", "DimDw", "DimUps", "DimDws", "Hsector", "Hstatus", "Hs", "Nfoo", "diag_hybr",
                                  "bath_diag", "iter"],
}
TYPES = {"ED_VARS_GLOBAL.f90": {"H_operator": ["O"], "effective_bath_component": ["v", "lambda"],
                                "effective_bath": ["e", "v", "item"], "sector_map": ["map"]}}
INTERFACES = {"ED_VARS_GLOBAL.f90": ["dd_sparse_HxV"]}


def logical_lines(text):
    """free-form source -> statements: comments and preprocessor lines dropped, continuations joined, lower case"""
    out, cur = [], ""
    for raw in text.splitlines():
        line = raw.split("!")[0].rstrip()
        if not line.strip() or line.lstrip().startswith("#"):
            continue
        line = line.strip()
        if line.startswith("&"):
            line = line[1:]
        if line.endswith("&"):
            cur += line[:-1] + " "
            continue
        out.append((cur + line).lower())
        cur = ""
    return out


def split_top(s, sep=","):
    parts, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([":
            depth += 1
        elif ch in ")]":
            depth -= 1
        if ch == sep and depth == 0:
            parts.append(cur); cur = ""
        else:
            cur += ch
    parts.append(cur)
    return [p.strip() for p in parts if p.strip()]


def rank_of(spec):
    return len(split_top(spec)) if spec else 0


def norm_type(t):
    t = t.replace(" ", "")
    t = re.sub(r"^real\((kind=)?8\)|^real\*8|^doubleprecision", "real8", t)
    t = re.sub(r"^character\(len=(\d+)\)", r"character\1", t)
    return t


def parse_decl(stmt):
    """'integer,dimension(:),allocatable :: a,b(3)=0' -> {name: (type, rank, attrs)}"""
    if "::" not in stmt:
        return {}
    left, right = stmt.split("::", 1)
    parts = split_top(left)
    if not parts:
        return {}
    typ = norm_type(parts[0])
    if not re.match(r"(integer|real8|real|logical|character|complex|type\(|procedure\()", typ):
        return {}
    rank, attrs = 0, set()
    for a in parts[1:]:
        a = a.replace(" ", "")
        m = re.match(r"dimension\((.*)\)$", a)
        if m:
            rank = rank_of(m.group(1))
        elif a in ("allocatable", "pointer"):
            attrs.add(a)
    out = {}
    for ent in split_top(right):
        ent = ent.split("=")[0].strip()
        m = re.match(r"(\w+)\s*(\((.*)\))?$", ent)
        if not m:
            continue
        r = rank_of(m.group(3)) if m.group(3) else rank
        out[m.group(1)] = (typ, r, frozenset(attrs))
    return out


def parse_file(path):
    stmts = logical_lines(open(path).read())
    variables, types, interfaces = {}, {}, {}
    i = 0
    while i < len(stmts):
        s = stmts[i]
        m = re.match(r"type\s*(,\s*\w+\s*)*(::)?\s*(\w+)$", s)
        if m and not s.startswith("type("):
            name, comps = m.group(3), {}
            i += 1
            while i < len(stmts) and not re.match(r"end\s*type", stmts[i]):
                comps.update(parse_decl(stmts[i])); i += 1
            types[name] = comps
        elif re.match(r"abstract\s+interface", s):
            i += 1
            while i < len(stmts) and not re.match(r"end\s*interface", stmts[i]):
                m2 = re.match(r"subroutine\s+(\w+)\s*\((.*)\)", stmts[i])
                if m2:
                    name, args, decls = m2.group(1), [a.strip() for a in m2.group(2).split(",")], {}
                    i += 1
                    while not re.match(r"end\s*subroutine", stmts[i]):
                        decls.update(parse_decl(stmts[i])); i += 1
                    interfaces[name] = (args, {a: decls.get(a) for a in args})
                i += 1
        else:
            for k, v in parse_decl(s).items():
                variables.setdefault(k, v)
        i += 1
    return variables, types, interfaces


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    a = ap.parse_args()
    if not os.path.isdir(a.reference):
        print("reference tree not present (%s): nothing to check against" % a.reference)
        return 2
    sv, st, si = parse_file(SCAFFOLD)
    bad, n = [], 0
    for fn, names in NAMES.items():
        rv, rt, ri = parse_file(os.path.join(a.reference, fn))
        for nm in names:
            k = nm.lower(); n += 1
            if k not in rv:
                bad.append("%s: %s not found in the reference" % (fn, nm)); continue
            if k not in sv:
                bad.append("%s: %s missing in the scaffold" % (fn, nm)); continue
            if rv[k] != sv[k]:
                bad.append("%s: %s reference %s / scaffold %s" % (fn, nm, rv[k], sv[k]))
        for tn, comps in TYPES.get(fn, {}).items():
            for c in comps:
                n += 1
                r = rt.get(tn.lower(), {}).get(c.lower()); s = st.get(tn.lower(), {}).get(c.lower())
                if r is None or s is None or r != s:
                    bad.append("%s: type %s component %s reference %s / scaffold %s" % (fn, tn, c, r, s))
        for itf in INTERFACES.get(fn, []):
            n += 1
            if ri.get(itf.lower()) != si.get(itf.lower()) or itf.lower() not in ri:
                bad.append("%s: interface %s reference %s / scaffold %s" % (fn, itf, ri.get(itf.lower()), si.get(itf.lower())))
    for b in bad:
        print("MISMATCH", b)
    print("%d declarations checked, %d mismatches" % (n, len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
