"""A small parser of Julia method signatures -- `function name(pos...; kw=default...) [where T]` and the one-line form
`name(pos...) = ...` -- used to compare the keyword surface of the reference's hot-path functions with the reference-side binding
(julia/LPVSpectralAMD.jl) and with the Python mirror (lpvspectral.jl_amd/api.py).  Test infrastructure."""
import re


def _split_top(s, sep):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    out.append(cur)
    return [a.strip() for a in out if a.strip()]


def _norm(expr):
    expr = re.sub(r"\s+", "", expr)
    expr = re.sub(r"\b(PO|ProximalOperators)\.", "", expr)
    return expr


def _param(p):
    """'name::Type = default' -> (name, default | None); 'kwargs...' -> ('kwargs...', None)"""
    parts, depth, eq = p, 0, -1
    for i, ch in enumerate(p):
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        elif ch == "=" and depth == 0 and p[i + 1:i + 2] != "=" and p[i - 1:i] not in ("=", "<", ">", "!"):
            eq = i
            break
    name, default = (p[:eq], p[eq + 1:]) if eq >= 0 else (p, None)
    name = name.strip()
    name = re.sub(r"::.*$", "", name).strip()
    return name, (None if default is None else _norm(default))


def parse_methods(src, names):
    """{name: [ {pos: [(name, default)], kw: [(name, default)], varkw: bool}, ... ]} for the wanted function names"""
    src = re.sub(r"#=.*?=#", "", src, flags=re.S)
    src = "\n".join(re.sub(r"#[^\n\"]*$", "", l) for l in src.splitlines())   # line comments (none of the signatures holds a '#' in a string)
    out = {n: [] for n in names}
    for n in names:
        for m in re.finditer(r"(?m)^[ \t]*(?:function\s+|@inline\s+function\s+)?%s\(" % re.escape(n), src):
            i, depth = m.end(), 1
            while depth:
                depth += {"(": 1, ")": -1}.get(src[i], 0)
                i += 1
            inner = src[m.end():i - 1]
            tail = src[i:i + 40]
            is_function = "function" in m.group(0)
            if not is_function and not re.match(r"\s*(where\s+\w+\s*)?=(?!=)", tail):
                continue                                   # a call at the start of a line, not a definition
            halves = _split_top(inner, ";")
            pos = [_param(p) for p in _split_top(halves[0], ",")] if halves else []
            kw = [_param(p) for p in _split_top(halves[1], ",")] if len(halves) > 1 else []
            out[n].append(dict(pos=pos, kw=[k for k in kw if not k[0].endswith("...")], varkw=any(k[0].endswith("...") for k in kw)))
    return out


HOT_PATH_FUNCTIONS = ["ls_spectral", "tls_spectral", "ls_sparse_spectral", "ls_sparse_spectral_lpv", "ls_spectral_lpv", "ls_windowpsd",
                      "ls_windowcsd", "ls_cohere", "ls_windowpsd_lpv", "ADMM", "get_fourier_regressor", "check_freq", "default_freqs",
                      "Windows2", "Windows3"]
