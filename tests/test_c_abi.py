"""The C ABI consumed from plain C99 (tests/c/abi_smoke.c), compiled with gcc."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "rustsasa_amd", "lib")


def _build(tmp_path):
    exe = str(tmp_path / "abi_smoke")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", exe, "-L", LIBDIR, "-lrustsasa_amd",
           "-Wl,-rpath," + LIBDIR]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    return exe


def test_header_compiles_as_c99_and_links(tmp_path):
    import rustsasa_amd
    exe = _build(tmp_path)
    if rustsasa_amd.device_count() > 0:
        pytest.skip("GPU host: covered by the gpu-marked test")
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 0 and "no device" in p.stdout, (p.returncode, p.stdout, p.stderr)


@pytest.mark.gpu
def test_c_consumer_on_gpu(tmp_path):
    p = subprocess.run([_build(tmp_path)], capture_output=True, text=True)
    assert p.returncode == 0 and "abi ok" in p.stdout, (p.returncode, p.stdout, p.stderr)
