"""The reference-side binding (julia/LPVSpectralAMD.jl) cannot be executed here (no julia in the image).  What CAN be
checked mechanically is checked: every ``@ccall`` in the file is parsed and compared -- symbol, arity, the C type of every
argument and the return type -- with the prototypes of include/lpvspectral.h and with the ctypes table the parity tests
call through; the wrapper must bind every entry point the drop-in API needs; converted arrays must be protected by
GC.@preserve under the name that is passed.  CPU only."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "julia", "LPVSpectralAMD.jl")
HEADER = os.path.join(ROOT, "include", "lpvspectral.h")


def _split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [a.strip() for a in out]


def _ccalls():
    src = open(JL).read()
    src = re.sub(r"#[^\n]*", "", src)                       # comments
    calls = []
    for m in re.finditer(r"@ccall\s+LIB\.(\w+)\(", src):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        args = _split_top(src[m.end():i - 1])
        ret = re.match(r"::\s*([A-Za-z0-9_{}]+)", src[i:]).group(1)
        types, names = [], []
        for a in args:
            mm = re.search(r"::\s*([A-Za-z0-9_{}]+)\s*$", a)
            assert mm, f"{m.group(1)}: argument without a type annotation: {a!r}"
            types.append(mm.group(1)); names.append(a[:mm.start()].strip())
        # the GC.@preserve list guarding this call (same statement), if any
        stmt_start = src.rfind("\n", 0, m.start())
        prev = src[max(0, src.rfind("GC.@preserve", 0, m.start())):m.start()] if "GC.@preserve" in src[stmt_start - 400:m.start()] else ""
        calls.append(dict(name=m.group(1), types=types, args=names, ret=ret, preserve=prev))
    return calls


def _prototypes():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int32_t)\s*(lpvs_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        alist = [] if args in ("", "void") else [re.sub(r"/\*.*?\*/", "", a).strip() for a in args.split(",")]
        protos[name] = (ret, alist)
    return protos


def _c_class(decl):
    """C parameter declaration -> (kind, element)"""
    d = re.sub(r"\bconst\b", "", decl).strip()
    stars = d.count("*")
    base = re.match(r"([A-Za-z_0-9]+)", d).group(1)
    if stars == 0:
        return {"int64_t": ("i64", None), "int32_t": ("i32", None), "double": ("f64", None)}[base]
    if stars == 1:
        return ("ptr", {"double": "Float64", "float": "Float32", "int64_t": "Int64", "int32_t": "Int32", "lpvs_problem": "Cvoid", "char": "char"}[base])
    return ("ptrptr", {"lpvs_problem": "Cvoid", "double": "Float64"}[base])


def _jl_class(t):
    if t in ("Int64",): return ("i64", None)
    if t in ("Int32",): return ("i32", None)
    if t in ("Float64",): return ("f64", None)
    if t == "Cstring": return ("ptr", "char")
    m = re.match(r"(Ptr|Ref)\{(Ptr)\{(\w+)\}\}$", t)
    if m: return ("ptrptr", m.group(3))
    m = re.match(r"(Ptr|Ref)\{(\w+)\}$", t)
    assert m, f"unrecognised Julia ccall type {t}"
    return ("ptr", m.group(2))


def _ctypes_class(t):
    if t is C.c_int64: return "i64"
    if t is C.c_int32: return "i32"
    if t is C.c_double: return "f64"
    return "ptr"          # c_void_p, c_char_p, POINTER(...)


def test_every_ccall_matches_the_header_and_the_ctypes_table():
    import sys
    sys.path.insert(0, ROOT)
    from lpvspectral_jl_amd._lib import SIGNATURES
    protos = _prototypes()
    calls = _ccalls()
    assert len(calls) >= 25
    for c in calls:
        assert c["name"] in protos, f"@ccall of {c['name']}: not declared in include/lpvspectral.h"
        ret, cargs = protos[c["name"]]
        assert len(c["types"]) == len(cargs), f"{c['name']}: {len(c['types'])} arguments in the @ccall, {len(cargs)} in the header"
        assert (c["ret"] == "Cstring") == ret.startswith("const"), f"{c['name']}: return type {c['ret']} vs {ret}"
        if not ret.startswith("const"):
            assert c["ret"] == "Int32", f"{c['name']}: status return must be Int32, is {c['ret']}"
        sig = SIGNATURES[c["name"]][1]
        assert len(sig) == len(cargs)
        for k, (jt, cd, ct) in enumerate(zip(c["types"], cargs, sig)):
            jk, je = _jl_class(jt)
            ck, ce = _c_class(cd)
            assert jk.replace("ptrptr", "ptr") == _ctypes_class(ct), f"{c['name']} arg {k}: Julia {jt} vs ctypes {ct}"
            assert jk == ck, f"{c['name']} arg {k} ({cd}): Julia {jt} is {jk}, header is {ck}"
            if jk in ("ptr", "ptrptr"):
                assert je == ce or (c["args"][k] == "C_NULL"), f"{c['name']} arg {k} ({cd}): Julia element type {je}, header {ce}"


def test_wrapper_binds_the_whole_drop_in_surface():
    bound = {c["name"] for c in _ccalls()}
    needed = {
        "lpvs_check_freq_f64", "lpvs_fourier_regressor_f64", "lpvs_problem_create_fourier_f64", "lpvs_problem_create_lpv_f64",
        "lpvs_problem_create_lpv_multi_f64", "lpvs_problem_create_dense_f64", "lpvs_problem_create_gram_f64", "lpvs_problem_destroy",
        "lpvs_problem_set_prox", "lpvs_admm_init_f64", "lpvs_admm_run", "lpvs_admm_get_f64", "lpvs_admm_set_state_f64",
        "lpvs_problem_get_params_f64", "lpvs_problem_pack_params_f64", "lpvs_problem_solve_ridge_f64", "lpvs_problem_get_gram_f64",
        "lpvs_problem_get_rhs_f64", "lpvs_problem_get_inverse_f64", "lpvs_ls_spectral_f64", "lpvs_lpv_regressor_f64",
        "lpvs_window_count", "lpvs_window_offsets", "lpvs_merge_f64", "lpvs_windows_estimate_multi_f64", "lpvs_last_error",
    }
    assert needed <= bound, sorted(needed - bound)
    src = open(JL).read()
    for fn in ("ls_spectral", "tls_spectral", "ls_sparse_spectral", "ls_sparse_spectral_lpv", "ls_spectral_lpv", "ls_windowpsd", "ls_windowcsd",
               "ls_cohere", "ls_windowpsd_lpv", "Windows2", "Windows3", "mapwindows", "ADMM", "psd", "get_fourier_regressor", "check_freq"):
        assert re.search(r"function\s+%s\(|^%s\(" % (fn, fn), src, flags=re.M), f"{fn} is not defined in the Julia wrapper"
    # the reference's prox objects are the dispatch types; no same-named shadow structs
    assert "import ProximalOperators" in src and not re.search(r"^struct\s+(NormL1|NormL0|IndBallL0)\b", src, flags=re.M)
    for t in ("PO.NormL1", "PO.NormL0", "PO.IndBallL0", "PO.SlicedSeparableSum"):
        assert "proxparams(g::%s" % t in src
    # the reference's progress lines are reproduced byte for byte (src/lasso.jl:159,165-166)
    assert src.count('@printf("%d ||x-z||₂ %.10f\\n"') == 2 and '@info("||x-z||₂ ≤ tol")' in src
    # Σ of ls_spectral_lpv is returned, the unweighted ls_spectral goes through lpvs_ls_spectral_f64
    assert "SpectralExt(Y, X, V, w, Nv, λ, coulomb, normalize, prm, Σ)" in src


def test_array_arguments_are_preserved_under_the_name_that_is_passed():
    """Every pointer argument that is a Julia array variable must appear in the GC.@preserve list of its call (a temporary
    created inside the call expression cannot be protected -- the defect of the first version of this wrapper)."""
    for c in _ccalls():
        for a, t in zip(c["args"], c["types"]):
            if not t.startswith("Ptr{") or t == "Ptr{Cvoid}":
                continue
            if a in ("C_NULL",) or re.match(r"^(p|q)\.h$", a):
                continue
            assert re.match(r"^[A-Za-zΦ_][A-Za-z0-9_Φ]*$", a), f"{c['name']}: pointer argument {a!r} is an expression, not a named array"
            if a.endswith("p") and a[:-1] + "v" in c["preserve"]:
                continue                                     # a raw pointer derived from a preserved vector (Wp <- Wv, x0p <- x0v)
            assert re.search(r"\b%s\b" % re.escape(a), c["preserve"]), f"{c['name']}: {a} is passed as {t} without GC.@preserve"


# ---- the keyword surface of the reference's hot-path methods: names, order and defaults --------------------------------------------
# tests/golden/reference_signatures.json holds, for every method of the hot-path functions in src/lsfft.jl / src/lasso.jl /
# src/windows.jl, its positional names (with defaults) and its keywords (with defaults) -- extracted by
# tests/golden/make_reference_signatures.py.  A user who switches must find the same names with the same defaults.
def _reference_signatures():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "reference_signatures.json")))


def test_signature_fixture_matches_the_reference_when_it_is_present():
    import sys
    import pytest
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("/root/reference is not on this machine (GPU box): the committed fixture stands")
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_reference_signatures as mk
    import json
    assert json.loads(json.dumps(mk.extract())) == _reference_signatures(), "regenerate tests/golden/reference_signatures.json"


def test_julia_wrapper_has_the_references_signatures():
    """Every reference method has a wrapper method of the same name with the same positional names / defaults, and every reference
    keyword with the same default; the wrapper may ADD keywords (device, ngpus, storage, ...)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _jlsig import HOT_PATH_FUNCTIONS, parse_methods
    ref = _reference_signatures()
    mine = parse_methods(open(JL).read(), HOT_PATH_FUNCTIONS)
    for name, methods in ref.items():
        for m in methods:
            pos = [tuple(p) for p in m["pos"]]
            cands = [w for w in mine[name] if [p[0] for p in w["pos"]] == [p[0] for p in pos]]
            assert cands, f"{name}({', '.join(p[0] for p in pos)}) of {m['file']} has no method with these positional arguments in the wrapper"
            w = cands[0]
            assert [p[1] for p in w["pos"]] == [p[1] for p in pos], f"{name}: positional defaults {w['pos']} vs reference {pos}"
            wkw = dict(w["kw"])
            for k, d in m["kw"]:
                assert k in wkw, f"{name}: keyword {k} of the reference is missing in the wrapper (has {sorted(wkw)})"
                assert wkw[k] == d, f"{name}: keyword {k} defaults to {wkw[k]} in the wrapper, {d} in the reference"
            if m["varkw"]:
                assert w["varkw"], f"{name}: the reference forwards kwargs..., the wrapper does not"


def test_python_mirror_has_the_references_signatures():
    """The same for lpvspectral.jl_amd/api.py (the host mirror the parity tests call): positional names in order, every reference
    keyword present with the same default.  Julia defaults that are expressions in earlier arguments (`f=default_freqs(t)`,
    `proxg=NormL1(λ)`, `estimator=ls_spectral`, `n=length(y)>>3`) are `None` in Python and resolved in the body."""
    import inspect
    import sys
    sys.path.insert(0, ROOT)
    import lpvspectral_jl_amd as L
    ref = _reference_signatures()
    lit = {"false": False, "true": True, "nothing": None, "T(1)": 1.0, "T(0.05)": 0.05, "rect": L.rect}
    def value(d):
        if d in lit:
            return lit[d]
        try:
            return float(d) if any(c in d for c in ".e") else int(d)
        except ValueError:
            return None                                    # an expression in earlier arguments
    for name, methods in ref.items():
        fn = getattr(L, name)
        sig = inspect.signature(fn.__init__ if inspect.isclass(fn) else fn)
        params = [p for p in sig.parameters.values() if p.name != "self"]
        names = [p.name for p in params]
        for m in methods:
            pos = [p[0] for p in m["pos"]]
            if name == "default_freqs":                    # three Julia methods share one Python function (t_or_n, fs, n)
                continue
            assert names[:len(pos)] == pos, f"{name}: positional {names[:len(pos)]} vs reference {pos}"
            for (k, d) in m["pos"]:
                if d is not None:
                    assert sig.parameters[k].default == value(d) or value(d) is None, f"{name}: {k} defaults to {sig.parameters[k].default!r}, reference {d}"
            for k, d in m["kw"]:
                assert k in sig.parameters, f"{name}: keyword {k} of the reference is missing in api.py"
                have, want = sig.parameters[k].default, value(d)
                assert have == want or (want is None and have is None), f"{name}: keyword {k} defaults to {have!r}, reference {d}"
            if m["varkw"]:
                assert any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params), f"{name}: the reference forwards kwargs..."
