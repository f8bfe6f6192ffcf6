import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle_binding

    return oracle_binding.load_oracle()


@pytest.fixture(scope="session", autouse=True)
def _short_term_variants_built(request):
    """The GPU tests of the n_points_short_term build variants (libsigmaenv_ns2.so / _ns5.so) need those libraries in the tree; __graft_entry__.build()
    makes them, and they travel with the tree like libsigmaenv.so.  On a checkout that has only the default library, build them once per GPU session
    (the product itself never compiles anything at load time: capi.load_library fails loudly instead)."""
    if request.config.getoption("-m") != "gpu":
        return
    import subprocess

    from sigmarl_amd import capi

    missing = [ns for ns in (2, 5) if not os.path.exists(capi.variant_path(ns))]
    jobs = [subprocess.Popen(["make", "-C", os.path.join(ROOT, "sigmarl_amd", "csrc"), f"NS={ns}"], stdout=subprocess.DEVNULL) for ns in missing]
    for j in jobs:
        j.wait()
