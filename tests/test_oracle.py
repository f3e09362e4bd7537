"""Pins the CPU oracle (oracle/sasa_oracle.c) to the reference's own tests.

* golden vector FIXED_LOW_RES_ATOMS            (reference tests/units.rs:18-43)
* six analytic cases at 50 000 points, 0.5 %    (reference tests/sanity.rs:20-157)
* neighbour-list membership on four atoms       (reference tests/units.rs:132-209)
"""
import math

import numpy as np
import pytest

import structio as sio
from oracle import pyoracle as po

PROBE = 1.4
HI_N = 50000
REL = 0.005


@pytest.fixture(scope="module")
def example_vdw():
    atoms = sio.read_structure(sio.data_path("example.cif"))
    return sio.soa_vdw(atoms)


@pytest.mark.parametrize("simd_width", [1, 4, 8, 16])
def test_golden_vector_per_atom(example_vdw, simd_width):
    x, y, z, r, ids = example_vdw
    gold = sio.load_golden_low_res()
    assert gold.shape == (2622,) == x.shape
    out, pts, k = po.calculate_sasa_internal(x, y, z, r, ids, PROBE, 100, simd_width,
                                             return_details=True)
    # north_star tolerance: 1e-4 A^2 absolute per atom
    assert np.max(np.abs(out - gold)) <= 1e-4
    # the golden values decode to integer point counts: k * 4*pi*(r+p)^2 / 100
    unit = 4.0 * math.pi * (r.astype(np.float64) + PROBE) ** 2 / 100.0
    k_gold = np.rint(gold / unit).astype(np.int64)
    assert np.max(np.abs(gold - k_gold * unit)) < 2e-5
    assert np.array_equal(k_gold, pts.astype(np.int64))
    # reference literal for the same run (tests/units.rs:58), f32 sequential sum
    total = np.float32(0)
    for v in out:
        total = np.float32(total + v)
    assert abs(float(total) - 20268.004) < 0.05
    assert k.max() <= 90 and 40 < k.mean() < 50


def test_lattice_shape():
    x, y, z = po.sphere_points(100)
    assert x[0] == 0.0 and y[0] == 0.0 and z[0] == 1.0  # point 0 is the +z pole
    assert np.allclose(x * x + y * y + z * z, 1.0, atol=1e-6)
    assert z[-1] > -1.0  # no -z pole point (lib.rs:52: t = i/N, no half offset)


def _sasa(coords, radii, n_points=HI_N):
    c = np.asarray(coords, np.float32)
    ids = np.arange(1, len(radii) + 1, dtype=np.uint64)
    return po.calculate_sasa_internal(c[:, 0], c[:, 1], c[:, 2], np.asarray(radii, np.float32),
                                      ids, PROBE, n_points, 8)


def test_single_sphere():
    s = _sasa([[0, 0, 0]], [2.0])
    assert s[0] == pytest.approx(4 * math.pi * 3.4 ** 2, rel=REL)


def test_two_non_overlapping_spheres():
    s = _sasa([[0, 0, 0], [10, 0, 0]], [2.0, 2.0])
    e = 4 * math.pi * 3.4 ** 2
    assert s[0] == pytest.approx(e, rel=REL) and s[1] == pytest.approx(e, rel=REL)
    assert float(s.sum()) == pytest.approx(2 * e, rel=REL)


def test_two_overlapping_spheres():
    s = _sasa([[0, 0, 0], [4, 0, 0]], [2.0, 2.0])
    r, dist = 3.4, 4.0
    exposed = 4 * math.pi * r * r - 2 * math.pi * r * (r - dist / 2)
    assert s[0] == pytest.approx(exposed, rel=REL) and s[1] == pytest.approx(exposed, rel=REL)


def test_contained_sphere():
    s = _sasa([[0, 0, 0], [2, 0, 0]], [10.0, 2.0])
    assert s[0] == pytest.approx(4 * math.pi * 11.4 ** 2, rel=REL)
    assert abs(s[1]) <= REL


def test_three_spheres_linear_chain():
    s = _sasa([[0, 0, 0], [5, 0, 0], [10, 0, 0]], [2.0, 2.0, 2.0])
    r = 3.4
    buried = 2 * math.pi * r * (r - 2.5)
    full = 4 * math.pi * r * r
    assert s[0] == pytest.approx(full - buried, rel=REL)
    assert s[2] == pytest.approx(full - buried, rel=REL)
    assert s[1] == pytest.approx(full - 2 * buried, rel=REL)


def test_empty_atom_list():
    e = np.zeros(0, np.float32)
    assert po.calculate_sasa_internal(e, e, e, e, None, PROBE, HI_N, 8).shape == (0,)


def test_spatial_grid_membership():
    c = np.array([[0, 0, 0], [3, 0, 0], [0, 3, 0], [20, 20, 20]], np.float32)
    r = np.full(4, 1.5, np.float32)
    ids = np.arange(1, 5, dtype=np.uint64)
    lists = po.neighbor_lists(c[:, 0], c[:, 1], c[:, 2], r, ids, probe_radius=1.4,
                              max_radius=1.5, cell_size=5.0, max_search_radius=1.5 + 1.5 + 2.8)
    n0 = set(lists[0]["idx"].tolist())
    assert len(n0) >= 2 and {1, 2} <= n0 and 3 not in n0
    assert len(lists[3]) == 0
    assert 0 in lists[1]["idx"] and 0 in lists[2]["idx"]
    # payload: threshold = (r_nb + probe)^2
    assert lists[0]["threshold_squared"][0] == np.float32(np.float32(1.5) + np.float32(1.4)) ** 2


def test_duplicate_ids_never_occlude():
    # two coincident-ish atoms sharing an id are "the same atom" (lib.rs:124, grid :314)
    c = np.array([[0, 0, 0], [1, 0, 0]], np.float32)
    r = np.array([2.0, 2.0], np.float32)
    same = po.calculate_sasa_internal(c[:, 0], c[:, 1], c[:, 2], r, np.array([7, 7], np.uint64),
                                      PROBE, 1000, 8)
    full = np.float32(4 * math.pi) * np.float32(3.4) ** 2
    assert np.allclose(same, full, rtol=1e-6)


def test_residue_sums_sequential():
    v = np.array([1e8, 1.0, -1e8, 3.0, 4.0], np.float32)
    out = po.residue_sums(v, np.array([0, 3, 3, 5], np.uint32))
    assert out[0] == np.float32(np.float32(np.float32(1e8) + np.float32(1.0)) - np.float32(1e8))
    assert out[1] == 0.0 and out[2] == 7.0


def test_batch_matches_single(example_vdw):
    x, y, z, r, ids = example_vdw
    one = po.calculate_sasa_internal(x, y, z, r, ids, PROBE, 100, 8)
    xs, ys, zs, rs = (np.concatenate([a, a[:500]]) for a in (x, y, z, r))
    idb = np.concatenate([ids, ids[:500]])
    offs = np.array([0, len(x), len(x), len(x) + 500], np.uint32)
    out = po.calculate_sasa_batch(xs, ys, zs, rs, idb, offs, PROBE, 100, 8, threads=2)
    assert np.array_equal(out[:len(x)], one)
    sub = po.calculate_sasa_internal(x[:500], y[:500], z[:500], r[:500], ids[:500], PROBE, 100, 8)
    assert np.array_equal(out[len(x):], sub)
