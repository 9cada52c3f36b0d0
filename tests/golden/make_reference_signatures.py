#!/usr/bin/env python3
"""Extracts the positional / keyword names and default expressions of the reference's hot-path methods (src/lsfft.jl, src/lasso.jl,
src/windows.jl of /root/reference) into tests/golden/reference_signatures.json -- data for the drop-in check of
tests/test_julia_binding.py (names and defaults only; no source text).  Run where /root/reference exists."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from _jlsig import HOT_PATH_FUNCTIONS, parse_methods   # noqa: E402

REF = os.environ.get("LPVS_REFERENCE", "/root/reference")


def extract():
    out = {}
    for rel in ("src/lsfft.jl", "src/lasso.jl", "src/windows.jl"):
        src = open(os.path.join(REF, rel)).read()
        for name, methods in parse_methods(src, HOT_PATH_FUNCTIONS).items():
            for m in methods:
                out.setdefault(name, []).append(dict(file=rel, pos=m["pos"], kw=m["kw"], varkw=m["varkw"]))
    return out


if __name__ == "__main__":
    with open(os.path.join(HERE, "reference_signatures.json"), "w") as fh:
        json.dump(extract(), fh, indent=1, ensure_ascii=False, sort_keys=True)
    print("wrote reference_signatures.json")
