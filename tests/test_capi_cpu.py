"""CPU-side checks of the C ABI: the library loads, exports every symbol the
header declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rustsasa_amd.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rsasa_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from rustsasa_amd import _capi
    lib = _capi.load()
    names = _declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/rustsasa_amd.h but not exported"
    assert set(names) == set(_capi.SYMBOLS), set(names) ^ set(_capi.SYMBOLS)
    assert lib.rsasa_abi_version() == 4


def test_struct_layouts_match_header():
    from rustsasa_amd import _capi
    assert _capi.ATOM_DTYPE.itemsize == 24
    assert _capi.ATOM_DTYPE.fields["radius"][1] == 12 and _capi.ATOM_DTYPE.fields["id"][1] == 16
    assert C.sizeof(_capi.DeviceBatch) == 13 * 8
    assert C.sizeof(_capi.Timings) == 40


def test_status_strings():
    from rustsasa_amd import _capi
    assert _capi.status_string(0) == "ok"
    assert "no CPU fallback" in _capi.status_string(_capi.RSASA_ERR_NO_DEVICE)


def test_lattice_is_bit_identical_to_oracle():
    import rustsasa_amd
    for n in (1, 100, 960, 50000):
        a = rustsasa_amd.sphere_points(n)
        b = po.sphere_points(n)
        assert all(np.array_equal(u, v) for u, v in zip(a, b))


def test_no_gpu_means_loud_failure():
    import rustsasa_amd
    if rustsasa_amd.device_count() > 0:
        pytest.skip("a GPU is visible; the loud-failure path is for GPU-less hosts")
    with pytest.raises(rustsasa_amd.RsasaError) as e:
        rustsasa_amd.Context(0)
    assert e.value.status == -2
    # the NULL-context convenience path fails the same way instead of computing on the CPU
    from rustsasa_amd import _capi
    x = np.zeros(3, np.float32)
    out = np.zeros(3, np.float32)
    rc = _capi.load().rsasa_calculate_sasa_soa(None, x.ctypes.data, x.ctypes.data, x.ctypes.data,
                                               x.ctypes.data, None, 3, 1.4, 100, out.ctypes.data)
    assert rc == _capi.RSASA_ERR_NO_DEVICE


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "rustsasa_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("no oracle", ""), f"{f} mentions the oracle"
