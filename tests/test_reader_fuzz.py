"""The structure reader / writer must reject or digest damaged files, never crash (no GPU used).
tools/fuzz_reader.py --asan runs the long version under AddressSanitizer + UBSan."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_reader_survives_mutated_files():
    import fuzz_reader
    from conftest import ensure_built
    cli = ensure_built()
    assert fuzz_reader.run_cases(cli, 60, seed=11) == []
