import os
import sys

import pytest

# Several tests select kernel / schedule variants and numerics experiments through the library's EXPERIMENT knobs (LPVS_FIX_BITS, LPVS_NIB_*,
# LPVS_FACTOR_SCHEME, LPVS_MULTI_*, ...): those are inert unless the process has this master switch (csrc/lpvs_internal.h: experiment_env).
os.environ.setdefault("LPVS_EXPERIMENTS", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a clean checkout has no built library (the .so files are git-ignored): build it once, as __graft_entry__.build() does
    so = os.path.join(ROOT, "lpvspectral.jl_amd", "liblpvspectral.so")
    if not os.path.exists(so) and os.path.exists("/opt/rocm/bin/hipcc"):
        import subprocess
        subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(ROOT, "lpvspectral.jl_amd", "csrc")])


def _have_gpu():
    try:
        import lpvspectral_jl_amd as L
        from lpvspectral_jl_amd._lib import lib
        return lib().lpvs_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def L():
    """The product package (loads liblpvspectral.so; ImportError if it was not built)."""
    import lpvspectral_jl_amd as L
    return L


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.lib()
    return o


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
