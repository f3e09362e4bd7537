import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the library reads its measurement switches (RSASA_OCCLUSION_KERNEL, RSASA_SUB_ATOMS, ...: kernel variants and paths the
# tests force) only when the process says RSASA_TUNING=1 - once, at its first use
os.environ.setdefault("RSASA_TUNING", "1")
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def ensure_built():
    """The engine library and the host-API test driver are build products (never tracked): build
    them when a fresh checkout has none.  Returns the driver's path."""
    import subprocess
    lib = os.path.join(ROOT, "rustsasa_amd", "lib", "librustsasa_amd.so")
    cli = os.path.join(ROOT, "rustsasa_amd", "lib", "sasa_host_cli")
    if not (os.path.exists(lib) and os.path.exists(cli)):
        subprocess.run(["make", "-C", os.path.join(ROOT, "rustsasa_amd", "csrc"), "all"], check=True,
                       stdout=subprocess.DEVNULL)
    return cli
