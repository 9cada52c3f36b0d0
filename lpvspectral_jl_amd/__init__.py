"""Importable alias of the package directory ``lpvspectral.jl_amd/`` (a dot is not legal in a Python
module name).  All code lives there; this file only points the import system at it."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "lpvspectral.jl_amd"))
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
