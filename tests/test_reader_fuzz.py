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


def test_short_cuts_equal_the_general_reader_on_seeded_mutants():
    """Directory mode's text -> atoms short cuts against the general reader + selection on the same mutants (structural
    mutations, byte noise in atom rows, the mmCIF row splitter's aligned <-> character-wise transitions): a short, seeded
    round with the regular build; `tools/fuzz_reader.py --differential --asan` runs 2 000 per format under ASan + UBSan
    (profiles/round5_reader_differential_fuzz.txt)."""
    import fuzz_reader
    from conftest import ensure_built
    cli = ensure_built()
    stats, bad = fuzz_reader.differential(cli, 120, seed=5, jobs=4)
    assert bad == [], bad[:3]
    by_class = stats.pop("by_class")
    assert stats["pdb"][1] >= 20 and stats["cif"][1] >= 20, stats          # the short cuts were taken ...
    assert stats["pdb"][0] - stats["pdb"][1] >= 20 and stats["cif"][0] - stats["cif"][1] >= 20  # ... and abandoned
    assert by_class[("cif", "splitter")][1] >= 10, by_class                # odd rows INSIDE the short cut
