"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and
the reference's golden vector.  Run on a MI355X with `pytest -m gpu`.

Tolerance (north_star): per-atom |GPU - reference| <= 1e-4 A^2.  Because a
per-atom value is k * 4*pi*(r+p)^2 / N with integer k, the tests additionally
require bit-equality with the oracle wherever the oracle is run.
"""
import math

import numpy as np
import pytest

import bench_workloads as bw
import structio as sio
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

PROBE = 1.4
TOL = 1e-4


@pytest.fixture(scope="module")
def ctx():
    import rustsasa_amd
    c = rustsasa_amd.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def example_vdw():
    return sio.soa_vdw(sio.read_structure(sio.data_path("example.cif")))


def test_library_loaded_is_in_tree():
    from rustsasa_amd import _capi
    assert _capi.LIB_PATH.endswith("rustsasa_amd/lib/librustsasa_amd.so")
    assert _capi.load().rsasa_abi_version() == 4


def test_golden_vector_example_cif(ctx, example_vdw):
    """Reference tests/units.rs:18-43 at the north-star tolerance instead of +-25."""
    x, y, z, r, ids = example_vdw
    gold = sio.load_golden_low_res()
    got = ctx.calculate_sasa_soa(x, y, z, r, ids, PROBE, 100)
    assert np.max(np.abs(got - gold)) <= TOL
    want = po.calculate_sasa_internal(x, y, z, r, ids, PROBE, 100, 8)
    assert np.array_equal(got, want)


def test_aos_entry_matches_soa(ctx, example_vdw):
    import rustsasa_amd
    x, y, z, r, ids = example_vdw
    atoms = rustsasa_amd.make_atoms(x, y, z, r, ids)
    got = ctx.calculate_sasa_internal(atoms, PROBE, 100, -1)
    assert np.array_equal(got, ctx.calculate_sasa_soa(x, y, z, r, ids, PROBE, 100))
    assert np.array_equal(got, ctx.calculate_sasa_internal(atoms, PROBE, 100, 1))


@pytest.mark.parametrize("name", ["1jcd.pdb", "151L_H3.pdb", "bad_seqadv_1A06.pdb", "example.cif"])
@pytest.mark.parametrize("n_points", [100, 960])
def test_fixtures_protor(ctx, name, n_points):
    """BASELINE configs 1/2: single PDB, ProtOr radii, AtomLevel, diff vs CPU <= 1e-4."""
    xyz, r, _, ids = bw.fixture_soa(name)
    x, y, z = (np.ascontiguousarray(xyz[:, k]).astype(np.float32) for k in range(3))
    got = ctx.calculate_sasa_soa(x, y, z, r, ids, PROBE, n_points)
    want = po.calculate_sasa_internal(x, y, z, r, ids, PROBE, n_points, 8)
    assert np.max(np.abs(got - want)) <= TOL
    assert np.array_equal(got, want)


@pytest.mark.parametrize("simd_width", [1, 4, 8, 16])
@pytest.mark.parametrize("n_points", [1, 7, 63, 64, 65, 100, 128, 129, 200, 257, 1000, 1100])
def test_point_counts_and_remainder_rule(simd_width, n_points):
    import rustsasa_amd
    xyz, r, _, ids = bw.fixture_soa("1jcd.pdb")
    x, y, z = (np.ascontiguousarray(xyz[:300, k]).astype(np.float32) for k in range(3))
    r, ids = r[:300], ids[:300]
    with rustsasa_amd.Context(0, simd_width=simd_width) as c:
        got = c.calculate_sasa_soa(x, y, z, r, ids, PROBE, n_points)
    want = po.calculate_sasa_internal(x, y, z, r, ids, PROBE, n_points, simd_width)
    assert np.array_equal(got, want)


# ---- the reference's analytic tests (tests/sanity.rs) on the GPU path -------

HI_N = 50000
REL = 0.005


def _sasa(ctx, coords, radii, n_points=HI_N):
    c = np.asarray(coords, np.float32)
    ids = np.arange(1, len(radii) + 1, dtype=np.uint64)
    return ctx.calculate_sasa_soa(c[:, 0].copy(), c[:, 1].copy(), c[:, 2].copy(),
                                  np.asarray(radii, np.float32), ids, PROBE, n_points)


def test_single_sphere(ctx):
    assert _sasa(ctx, [[0, 0, 0]], [2.0])[0] == pytest.approx(4 * math.pi * 3.4 ** 2, rel=REL)


def test_two_non_overlapping_spheres(ctx):
    s = _sasa(ctx, [[0, 0, 0], [10, 0, 0]], [2.0, 2.0])
    e = 4 * math.pi * 3.4 ** 2
    assert s[0] == pytest.approx(e, rel=REL) and s[1] == pytest.approx(e, rel=REL)


def test_two_overlapping_spheres(ctx):
    s = _sasa(ctx, [[0, 0, 0], [4, 0, 0]], [2.0, 2.0])
    r = 3.4
    exposed = 4 * math.pi * r * r - 2 * math.pi * r * (r - 2.0)
    assert s[0] == pytest.approx(exposed, rel=REL) and s[1] == pytest.approx(exposed, rel=REL)


def test_contained_sphere(ctx):
    s = _sasa(ctx, [[0, 0, 0], [2, 0, 0]], [10.0, 2.0])
    assert s[0] == pytest.approx(4 * math.pi * 11.4 ** 2, rel=REL)
    assert abs(s[1]) <= REL


def test_three_spheres_linear_chain(ctx):
    s = _sasa(ctx, [[0, 0, 0], [5, 0, 0], [10, 0, 0]], [2.0, 2.0, 2.0])
    r = 3.4
    buried = 2 * math.pi * r * (r - 2.5)
    full = 4 * math.pi * r * r
    assert s[0] == pytest.approx(full - buried, rel=REL)
    assert s[2] == pytest.approx(full - buried, rel=REL)
    assert s[1] == pytest.approx(full - 2 * buried, rel=REL)


def test_sanity_cases_bit_equal_to_oracle(ctx):
    for coords, radii in ([[[0, 0, 0]], [2.0]],
                          [[[0, 0, 0], [4, 0, 0]], [2.0, 2.0]],
                          [[[0, 0, 0], [2, 0, 0]], [10.0, 2.0]],
                          [[[0, 0, 0], [5, 0, 0], [10, 0, 0]], [2.0, 2.0, 2.0]]):
        c = np.asarray(coords, np.float32)
        ids = np.arange(1, len(radii) + 1, dtype=np.uint64)
        want = po.calculate_sasa_internal(c[:, 0], c[:, 1], c[:, 2], np.asarray(radii, np.float32),
                                          ids, PROBE, HI_N, 8)
        assert np.array_equal(_sasa(ctx, coords, radii), want)


def test_empty_atom_list(ctx):
    e = np.zeros(0, np.float32)
    assert ctx.calculate_sasa_soa(e, e, e, e, None, PROBE, HI_N).shape == (0,)


def test_duplicate_ids_never_occlude(ctx):
    c = np.array([[0, 0, 0], [1, 0, 0], [2.5, 0, 0]], np.float32)
    r = np.array([2.0, 2.0, 1.5], np.float32)
    for ids in (np.array([7, 7, 9], np.uint64), np.array([7, 8, 7], np.uint64), None):
        got = ctx.calculate_sasa_soa(c[:, 0].copy(), c[:, 1].copy(), c[:, 2].copy(), r, ids, PROBE, 500)
        want = po.calculate_sasa_internal(c[:, 0], c[:, 1], c[:, 2], r, ids, PROBE, 500, 8)
        assert np.array_equal(got, want)


# ---- candidate rule (spatial_grid.rs) ------------------------------------------

def _device_run(ctx, b, n_points=100, with_ids=True, want_k=True, want_res=True):
    import torch
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    x, y, z, r = t(b.x), t(b.y), t(b.z), t(b.radius)
    ids = t(b.ids.view(np.int64)) if with_ids else None
    ro = t(b.residue_offsets.view(np.int32)) if want_res else None
    out = torch.full((b.n_atoms,), -1.0, dtype=torch.float32, device=dev)
    res = torch.full((b.n_residues,), -1.0, dtype=torch.float32, device=dev) if want_res else None
    k = torch.zeros(b.n_atoms, dtype=torch.int32, device=dev) if want_k else None
    torch.cuda.synchronize()
    ctx.enqueue_device(x, y, z, r, ids, b.structure_offsets, out, ro, res, k, PROBE, n_points,
                       stream=torch.cuda.current_stream().cuda_stream)
    ctx.wait()
    return (out.cpu().numpy(), None if res is None else res.cpu().numpy(),
            None if k is None else k.cpu().numpy().view(np.uint32))


def test_neighbor_counts_match_reference_lists(ctx, example_vdw):
    x, y, z, r, ids = example_vdw
    b = bw.Batch(x, y, z, r, ids, np.array([0, len(x)], np.uint32), np.array([0, len(x)], np.uint32))
    _, _, k = _device_run(ctx, b)
    _, _, k_ref = po.calculate_sasa_internal(x, y, z, r, ids, PROBE, 100, 8, return_details=True)
    assert np.array_equal(k, k_ref)


def test_spatial_grid_membership_via_counts(ctx):
    """Reference tests/units.rs:132-209, through the public path (cell = probe + max_r)."""
    c = np.array([[0, 0, 0], [3, 0, 0], [0, 3, 0], [20, 20, 20]], np.float32)
    r = np.full(4, 1.5, np.float32)
    ids = np.arange(1, 5, dtype=np.uint64)
    b = bw.Batch(c[:, 0].copy(), c[:, 1].copy(), c[:, 2].copy(), r, ids,
                 np.array([0, 4], np.uint32), np.array([0, 4], np.uint32))
    _, _, k = _device_run(ctx, b)
    assert k.tolist() == [2, 2, 2, 0]


# ---- batches (directory mode) + ResidueLevel sums ---------------------------------

def test_batch_with_empty_and_tiny_structures(ctx):
    rng = np.random.default_rng(3)
    parts = [bw.synthetic_structure(n, rng) for n in (700, 1, 2, 1500, 150)]
    xs = np.concatenate([p[0] for p in parts])
    rs = np.concatenate([p[1] for p in parts])
    sizes = [len(p[1]) for p in parts]
    # an empty structure in the middle and at the end
    so = np.cumsum([0, sizes[0], 0, sizes[1], sizes[2], sizes[3], sizes[4], 0]).astype(np.uint32)
    ids = np.arange(len(rs), dtype=np.uint64)
    x, y, z = (np.ascontiguousarray(xs[:, k]) for k in range(3))
    res_off = np.unique(np.concatenate([so, np.arange(0, len(rs), 7, dtype=np.uint32)])).astype(np.uint32)
    atom, res = ctx.calculate_sasa_batch(x, y, z, rs, ids, so, PROBE, 100, residue_offsets=res_off)
    want = po.calculate_sasa_batch(x, y, z, rs, ids, so, PROBE, 100, 8, threads=4)
    assert np.array_equal(atom, want)
    assert np.array_equal(res, po.residue_sums(want, res_off))
    # residue-only request
    atom2, res2 = ctx.calculate_sasa_batch(x, y, z, rs, ids, so, PROBE, 100,
                                           residue_offsets=res_off, want_atoms=False)
    assert atom2 is None and np.array_equal(res2, res)


def test_proteome_slice_device_resident(ctx):
    """BASELINE config 3 on a 300-structure slice: every atom and residue vs the oracle."""
    b = bw.synthetic_proteome(300, seed=11)
    atom, res, k = _device_run(ctx, b)
    want = po.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, PROBE, 100,
                                   8, threads=8)
    assert np.max(np.abs(atom - want)) <= TOL
    assert np.array_equal(atom, want)
    assert np.array_equal(res, po.residue_sums(want, b.residue_offsets))
    assert 30 < k.mean() < 55
    # without ids (all distinct) the result is the same here: ids are unique per structure
    atom2, _, _ = _device_run(ctx, b, with_ids=False, want_k=False, want_res=False)
    assert np.array_equal(atom2, atom)


def test_dense_cluster_flushes_candidate_tiles(ctx):
    """More than 192 candidates per atom: the LDS candidate list is processed in tiles."""
    rng = np.random.default_rng(5)
    n = 600
    c = rng.normal(scale=2.0, size=(n, 3)).astype(np.float32)
    r = rng.choice(np.array([1.2, 1.6, 1.88], np.float32), size=n)
    ids = np.arange(n, dtype=np.uint64)
    for n_points in (100, 300):
        got = ctx.calculate_sasa_soa(c[:, 0].copy(), c[:, 1].copy(), c[:, 2].copy(), r, ids, PROBE, n_points)
        want, _, k = po.calculate_sasa_internal(c[:, 0], c[:, 1], c[:, 2], r, ids, PROBE, n_points, 8,
                                                return_details=True)
        assert k.max() > 192
        assert np.array_equal(got, want)


def test_uniform_box_960_points(ctx):
    """BASELINE config 5 at reduced size (60k atoms): 960 points, AtomLevel."""
    b = bw.synthetic_uniform(60_000, seed=5)
    atom, _, _ = _device_run(ctx, b, n_points=960, want_k=False, want_res=False)
    want = po.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, PROBE, 960,
                                   8, threads=8)
    assert np.array_equal(atom, want)


def test_probe_and_radius_variants(ctx):
    xyz, r, _, ids = bw.fixture_soa("151L_H3.pdb")
    x, y, z = (np.ascontiguousarray(xyz[:, k]).astype(np.float32) for k in range(3))
    for probe in (0.0, 0.5, 1.2, 2.5):
        got = ctx.calculate_sasa_soa(x, y, z, r, ids, probe, 100)
        assert np.array_equal(got, po.calculate_sasa_internal(x, y, z, r, ids, probe, 100, 8))
    r2 = r.copy()
    r2[::50] = 3.5  # a few fat atoms change max_r, hence the cell size
    got = ctx.calculate_sasa_soa(x, y, z, r2, ids, PROBE, 100)
    assert np.array_equal(got, po.calculate_sasa_internal(x, y, z, r2, ids, PROBE, 100, 8))


def test_errors_do_not_poison_context(ctx, example_vdw):
    import rustsasa_amd
    x, y, z, r, ids = example_vdw
    with pytest.raises(rustsasa_amd.RsasaError):
        ctx.calculate_sasa_soa(x, y, z, r, ids, PROBE, 0)          # n_points == 0
    with pytest.raises(rustsasa_amd.RsasaError):
        ctx.calculate_sasa_soa(x, y, z, r, ids, float("nan"), 100)
    with pytest.raises(rustsasa_amd.RsasaError):
        ctx.calculate_sasa_soa(x[:2], y[:2], z[:2], np.zeros(2, np.float32), None, 0.0, 100)  # cell size 0
    far = np.array([0.0, 1e9], np.float32)
    with pytest.raises(rustsasa_amd.RsasaError):
        ctx.calculate_sasa_soa(far, far, far, np.ones(2, np.float32), None, PROBE, 100)  # grid too large
    got = ctx.calculate_sasa_soa(x, y, z, r, ids, PROBE, 100)
    assert np.max(np.abs(got - sio.load_golden_low_res())) <= TOL


def test_full_proteome_properties(ctx):
    """BASELINE config 3 at full size: size-independent checks (no oracle pass over 11.7 M atoms).

    * a structure's values do not depend on its batch neighbours: 40 structures
      drawn from the batch are recomputed alone by the oracle and compared exactly;
    * every value is an integer multiple of 4*pi*(r+p)^2/100 within f32 rounding;
    * residue values are the sequential f32 sums of the atom values.
    """
    b = bw.synthetic_proteome()
    atom, res, _ = _device_run(ctx, b, want_k=False)
    assert atom.min() >= 0.0
    unit = (4.0 * math.pi * (b.radius.astype(np.float64) + PROBE) ** 2) / 100.0
    kf = atom / unit
    assert np.max(np.abs(kf - np.rint(kf))) < 1e-4 and kf.max() <= 100.0001
    assert np.array_equal(res, po.residue_sums(atom, b.residue_offsets))
    rng = np.random.default_rng(0)
    for s in rng.choice(b.n_structures, 40, replace=False):
        x, y, z, r, ids = b.structure(int(s))
        want = po.calculate_sasa_internal(x, y, z, r, ids, PROBE, 100, 8)
        lo, hi = b.structure_offsets[s], b.structure_offsets[s + 1]
        assert np.array_equal(atom[lo:hi], want)


def test_full_proteome_every_atom_and_residue(ctx):
    """BASELINE config 3 at FULL size against the oracle: all 4 363 structures, 11.72 M atoms and
    1.51 M residues, bit for bit (the tolerance of north_star is 1e-4 A^2; we demand equality)."""
    b = bw.synthetic_proteome()
    assert b.n_structures == 4363 and b.n_atoms > 11_000_000
    atom, res, k = _device_run(ctx, b)
    want = po.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, PROBE, 100,
                                   8, threads=0)
    bad = np.flatnonzero(atom != want)
    assert bad.size == 0, (bad.size, bad[:5], atom[bad[:5]], want[bad[:5]])
    assert float(np.max(np.abs(atom - want))) <= TOL
    assert np.array_equal(res, po.residue_sums(want, b.residue_offsets))
    assert 40.0 < k.mean() < 50.0   # 43-49 candidates per atom on protein-like input


def test_uniform_1m_atoms_960_points_full_size(ctx):
    """BASELINE config 5 at FULL size: one structure of 1 000 000 atoms, 960 points, AtomLevel.
    65 536 atoms or more: the batch-wide (tail) binning route, which the 60 000-atom case above
    does not take."""
    b = bw.synthetic_uniform(1_000_000, seed=5)
    atom, _, k = _device_run(ctx, b, n_points=960, want_res=False)
    want, _, want_k = po.calculate_sasa_internal(b.x, b.y, b.z, b.radius, b.ids, PROBE, 960, 8,
                                                 return_details=True, threads=0)
    bad = np.flatnonzero(atom != want)
    assert bad.size == 0, (bad.size, bad[:5], atom[bad[:5]], want[bad[:5]])
    assert np.array_equal(k, want_k)


@pytest.mark.parametrize("n_points", [200, 1000, 1100, 1344, 1348, 1352, 2000])
def test_uniform_box_many_points_with_remainder(ctx, n_points):
    """More than 128 points, with and without a remainder (1100 = 137 * 8 + 4), on a structure large
    enough for the batch-wide binning route and the matrix-core kernel: 4-, 8- and 12-wave workgroups
    (200 / 1000 / 1344 points), the largest count the kernel takes (1344), the first ones it leaves
    to the general kernel."""
    b = bw.synthetic_uniform(70_000, seed=9)
    atom, _, _ = _device_run(ctx, b, n_points=n_points, want_k=False, want_res=False)
    want = po.calculate_sasa_internal(b.x, b.y, b.z, b.radius, b.ids, PROBE, n_points, 8, threads=0)
    assert np.array_equal(atom, want)


@pytest.mark.parametrize("env", [{"RSASA_OCCLUSION_KERNEL": "0"},
                                 {"RSASA_OCCLUSION_KERNEL": "2", "RSASA_ATOMS_PER_WAVE": "5"},
                                 {"RSASA_OCCLUSION_KERNEL": "3", "RSASA_ATOMS_PER_WAVE": "1"},
                                 {"RSASA_OCCLUSION_KERNEL": "3", "RSASA_ATOMS_PER_WAVE": "7"},
                                 {"RSASA_OCCLUSION_KERNEL": "4", "RSASA_ATOMS_PER_WAVE": "1"},
                                 {"RSASA_OCCLUSION_KERNEL": "4", "RSASA_ATOMS_PER_WAVE": "3"},
                                 {"RSASA_OCCLUSION_KERNEL": "5"},
                                 {"RSASA_OCCLUSION_KERNEL": "5", "RSASA_ATOMS_PER_WAVE": "8"},
                                 {"RSASA_OCCLUSION_KERNEL": "5", "RSASA_ATOMS_PER_WAVE": "23"},
                                 {"RSASA_OVERLAP_TAIL": "1"},
                                 {"RSASA_SMALL_PATH": "0"}])
def test_kernel_variants_agree(env, monkeypatch):
    """Every occlusion kernel variant / wave schedule gives bit-identical results."""
    import rustsasa_amd
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    b = bw.synthetic_proteome(60, seed=21)
    want = po.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, PROBE, 100,
                                   8, threads=8)
    with rustsasa_amd.Context(0) as c:
        atom, _, k = _device_run(c, b)
        atom960, _ = c.calculate_sasa_batch(b.x[:3000], b.y[:3000], b.z[:3000], b.radius[:3000],
                                            b.ids[:3000], np.array([0, 3000], np.uint32), PROBE, 960)
    assert np.array_equal(atom, want)
    _, _, k_ref = po.calculate_sasa_internal(*b.structure(0), PROBE, 100, 8, return_details=True)
    assert np.array_equal(k[:len(k_ref)], k_ref)
    assert np.array_equal(atom960, po.calculate_sasa_internal(b.x[:3000], b.y[:3000], b.z[:3000],
                                                              b.radius[:3000], b.ids[:3000], PROBE, 960, 8))


# ---- size-independent properties ----------------------------------------------------

def test_small_inputs_on_the_matrix_core_kernel(monkeypatch):
    """By default batches below 32 768 atoms take the per-atom kernels (lower latency per call), so
    the small cases of this file would never reach k_occlusion_mx: force it (RSASA_OCCLUSION_KERNEL=5)
    through point counts with and without remainder points, every lane count of the remainder rule,
    duplicate ids, coincident atoms, empty and tiny structures."""
    import rustsasa_amd
    monkeypatch.setenv("RSASA_OCCLUSION_KERNEL", "5")
    rng = np.random.default_rng(12)
    with rustsasa_amd.Context(0) as c:
        for name in ("1jcd.pdb", "example.cif"):
            xyz, r, _, ids = bw.fixture_soa(name)
            x, y, z = (np.ascontiguousarray(xyz[:, k]).astype(np.float32) for k in range(3))
            for n_points in (1, 17, 96, 100, 128, 129, 200, 960, 1000, 1100, 2020):
                got = c.calculate_sasa_soa(x, y, z, r, ids, PROBE, n_points)
                assert np.array_equal(got, po.calculate_sasa_internal(x, y, z, r, ids, PROBE, n_points, 8)), (name, n_points)
            # ids: none, and duplicated (atoms sharing an id never occlude each other)
            got = c.calculate_sasa_soa(x, y, z, r, None, PROBE, 100)
            assert np.array_equal(got, po.calculate_sasa_internal(x, y, z, r, None, PROBE, 100, 8))
            dup = ids.copy()
            dup[rng.choice(len(dup), 200, replace=False)] = dup[0]
            got = c.calculate_sasa_soa(x, y, z, r, dup, PROBE, 100)
            assert np.array_equal(got, po.calculate_sasa_internal(x, y, z, r, dup, PROBE, 100, 8))
        # coincident atoms, an isolated atom, tiny and empty structures in one batch
        x = np.array([0, 0, 0, 50, 1.0, 1.5, 2.0, 9.0], np.float32)
        y = np.zeros(8, np.float32)
        z = np.zeros(8, np.float32)
        r = np.array([1.5, 1.5, 1.7, 1.5, 1.2, 1.8, 1.5, 1.5], np.float32)
        so = np.array([0, 4, 4, 5, 8], np.uint32)
        got, _ = c.calculate_sasa_batch(x, y, z, r, None, so, PROBE, 100)
        want = po.calculate_sasa_batch(x, y, z, r, None, so, PROBE, 100, 8, threads=1)
        assert np.array_equal(got, want)
    for w in (1, 4, 16):
        xyz, r, _, ids = bw.fixture_soa("151L_H3.pdb")
        x, y, z = (np.ascontiguousarray(xyz[:, k]).astype(np.float32) for k in range(3))
        with rustsasa_amd.Context(0, simd_width=w) as c:
            for n_points in (100, 103, 960):
                got = c.calculate_sasa_soa(x, y, z, r, ids, PROBE, n_points)
                assert np.array_equal(got, po.calculate_sasa_internal(x, y, z, r, ids, PROBE, n_points, w)), (w, n_points)


@pytest.mark.parametrize("n_points,simd_width", [(35, 8), (63, 4), (66, 16), (70, 4), (83, 16), (97, 8), (99, 16), (100, 8), (100, 16),
                                                 (103, 4), (115, 16), (122, 8), (127, 4), (113, 16)])
def test_remainder_points_have_an_exact_pass_of_their_own(monkeypatch, n_points, simd_width):
    """k_occlusion_mx, at most 128 points: the remainder points (the last n_points % W, lib.rs:163-218: plain products, `<=`) the
    filter leaves alive are tested apart, exactly (phase_b_rem) - wherever they sit: among the first 64 points (35 / 8, 63 / 4:
    lanes of the filter's first result), at the start of the second 64 (66 / 16), in any of its rows (83, 97-103, 113-127), one to
    four of them, with 6, 7 and 8 tiles of points.  Every atom against the oracle run with the same lane count; the general
    kernel gets nothing because of them."""
    import rustsasa_amd
    assert 0 < n_points % simd_width <= 4
    monkeypatch.setenv("RSASA_OCCLUSION_KERNEL", "5")
    for name in ("example.cif", "1jcd.pdb"):
        xyz, r, _, ids = bw.fixture_soa(name)
        x, y, z = (np.ascontiguousarray(xyz[:, k]).astype(np.float32) for k in range(3))
        with rustsasa_amd.Context(0, simd_width=simd_width) as c:
            got = c.calculate_sasa_soa(x, y, z, r, ids, PROBE, n_points)
            want = po.calculate_sasa_internal(x, y, z, r, ids, PROBE, n_points, simd_width)
            assert np.array_equal(got, want), (name, n_points, simd_width, int((got != want).sum()))
            # (the same atoms without ids, timed: nothing is left to the general kernel - before round 6 a remainder point
            # inside the matrix rounds' certainty band was)
            c.enable_timing(True)
            got = c.calculate_sasa_soa(x, y, z, r, None, PROBE, n_points)
            assert c.timings()["n_deferred"] == 0
            c.enable_timing(False)
            assert np.array_equal(got, po.calculate_sasa_internal(x, y, z, r, None, PROBE, n_points, simd_width))


def test_patch_test_of_the_many_point_kernel_on_cap_boundaries(monkeypatch):
    """More than 128 points: whole tiles of 16 points (compact patches of the sphere) are dropped when ONE near
    candidate's cap holds the patch (occlusion_mx.inc, patch test).  Caps of every size - a neighbour at distances
    from touching to almost concentric, several radii, large and zero probes - put their rim across patches in every
    way; tile counts that are not whole blocks of 64, point counts that are not whole tiles, with remainder points
    in the last tile."""
    import rustsasa_amd
    monkeypatch.setenv("RSASA_OCCLUSION_KERNEL", "5")
    rng = np.random.default_rng(77)
    xs, ys, zs, rs, so = [], [], [], [], [0]
    for d in np.linspace(0.05, 7.9, 160):          # pairs at growing distance, random direction
        u = rng.normal(size=3)
        u /= np.linalg.norm(u)
        c0 = rng.uniform(-50, 50, 3)
        pts = [c0, c0 + d * u]
        if rng.random() < 0.5:                       # and sometimes a third atom nearby
            pts.append(c0 + rng.normal(size=3) * 2.0)
        for q in pts:
            xs.append(q[0]); ys.append(q[1]); zs.append(q[2])
            rs.append(rng.choice([1.2, 1.5, 1.88, 2.5, 0.7]))
        so.append(len(xs))
    x, y, z = (np.round(np.array(a), 3).astype(np.float32) for a in (xs, ys, zs))
    r = np.array(rs, np.float32)
    so = np.array(so, np.uint32)
    for w in (8, 16):
        with rustsasa_amd.Context(0, simd_width=w) as c:
            for probe in (1.4, 0.0, 4.0):
                for n_points in (129, 144, 145, 500, 960, 1030, 1344):
                    got, _ = c.calculate_sasa_batch(x, y, z, r, None, so, probe, n_points)
                    want = po.calculate_sasa_batch(x, y, z, r, None, so, probe, n_points, w, threads=0)
                    assert np.array_equal(got, want), (w, probe, n_points, int(np.count_nonzero(got != want)))


def test_matrix_core_kernel_hands_over_what_it_does_not_take(monkeypatch):
    """k_occlusion_mx takes atoms whose radii and probe keep its f16 operands, its three-step quotient and the
    remainder rule's error bound valid (probe in [0, 32], r + probe >= 0.5, r + max_r + 2 probe <= 64, every
    radius of the structure in [0, 64]); everything else goes to the general kernel.  All of it must equal
    the oracle: negative, zero, huge and tiny radii, large and zero probes, mixed in one batch with ordinary
    structures."""
    import rustsasa_amd
    monkeypatch.setenv("RSASA_OCCLUSION_KERNEL", "5")
    b = bw.synthetic_proteome(8, seed=5)
    so = b.structure_offsets
    rng = np.random.default_rng(8)
    with rustsasa_amd.Context(0) as c:
        for probe, scale in ((PROBE, 1.0), (0.0, 1.0), (0.05, 0.2), (33.0, 1.0), (2.0, 12.0)):
            r = (b.radius * np.float32(scale)).astype(np.float32)
            # structure 1: a few negative radii; 2: one sphere of radius 70; 3: zeros; 4: radii just below / above 64
            for s, vals in ((1, (-1.0, -0.25)), (2, (70.0,)), (3, (0.0,)), (4, (63.5, 64.5))):
                if s + 1 >= len(so):
                    continue
                idx = rng.choice(np.arange(so[s], so[s + 1]), 4 * len(vals), replace=False)
                r[idx] = np.resize(np.array(vals, np.float32), idx.shape)
            got, _ = c.calculate_sasa_batch(b.x, b.y, b.z, r, b.ids, so, probe, 100)
            want = po.calculate_sasa_batch(b.x, b.y, b.z, r, b.ids, so, probe, 100, 8, threads=4)
            assert np.array_equal(got, want), (probe, scale, int(np.sum(got != want)))


def test_matrix_core_kernel_and_grids_wider_than_1024_cells(monkeypatch):
    """k_occlusion_mx keeps an atom's three cell coordinates in one register, ten bits each; a structure whose grid has
    more than 1024 cells along an axis is left to the general kernel.  Stretched copies of ordinary structures - two
    compact halves 4 000 A apart along x, y or z - among ordinary ones, all equal to the oracle."""
    import rustsasa_amd
    monkeypatch.setenv("RSASA_OCCLUSION_KERNEL", "5")
    b = bw.synthetic_proteome(6, seed=9)
    so = b.structure_offsets
    x, y, z = b.x.copy(), b.y.copy(), b.z.copy()
    for s, axis in ((1, x), (3, y), (4, z)):
        half = (so[s] + so[s + 1]) // 2
        axis[half:so[s + 1]] += np.float32(4000.0)  # > 1024 cells of probe + max_r (about 3.3 A) between the halves
    with rustsasa_amd.Context(0) as c:
        got, _ = c.calculate_sasa_batch(x, y, z, b.radius, b.ids, so, PROBE, 100)
    want = po.calculate_sasa_batch(x, y, z, b.radius, b.ids, so, PROBE, 100, 8, threads=4)
    assert np.array_equal(got, want), int(np.sum(got != want))


def test_default_dispatch_at_the_matrix_core_threshold(ctx):
    """Batches of 32 768 atoms or more take k_occlusion_mx, smaller ones the per-atom kernels: both sides of the
    boundary, default settings, against the oracle."""
    b = bw.synthetic_proteome(40, seed=19)
    assert b.n_atoms > 40000
    for n in (32767, 32768):
        so = np.append(b.structure_offsets[b.structure_offsets < n], np.uint32(n)).astype(np.uint32)
        got, _ = ctx.calculate_sasa_batch(b.x[:n], b.y[:n], b.z[:n], b.radius[:n], b.ids[:n], so, PROBE, 100)
        want = po.calculate_sasa_batch(b.x[:n], b.y[:n], b.z[:n], b.radius[:n], b.ids[:n], so, PROBE, 100, 8, threads=4)
        assert np.array_equal(got, want), n


def test_permutation_invariance(ctx):
    """A per-atom value depends on the SET of atoms, not on their order in the input: shuffling
    the atoms permutes the output bit for bit (the candidate set and every test are order free)."""
    xyz, r, _, ids = bw.fixture_soa("bad_seqadv_1A06.pdb")
    x, y, z = (np.ascontiguousarray(xyz[:, k]).astype(np.float32) for k in range(3))
    base = ctx.calculate_sasa_soa(x, y, z, r, ids, PROBE, 100)
    rng = np.random.default_rng(4)
    for _ in range(3):
        perm = rng.permutation(len(x))
        got = ctx.calculate_sasa_soa(x[perm].copy(), y[perm].copy(), z[perm].copy(), r[perm].copy(),
                                     ids[perm].copy(), PROBE, 100)
        assert np.array_equal(got, base[perm])


def test_batch_neighbours_do_not_interact(ctx):
    """Structures of a batch are independent even when they overlap in space."""
    xyz, r, _, ids = bw.fixture_soa("1jcd.pdb")
    x, y, z = (np.ascontiguousarray(xyz[:, k]).astype(np.float32) for k in range(3))
    one = ctx.calculate_sasa_soa(x, y, z, r, ids, PROBE, 100)
    reps = 5
    so = np.arange(0, (reps + 1) * len(x), len(x), dtype=np.uint32)
    atom, _ = ctx.calculate_sasa_batch(np.tile(x, reps), np.tile(y, reps), np.tile(z, reps),
                                       np.tile(r, reps), np.tile(ids, reps), so, PROBE, 100)
    assert np.array_equal(atom, np.tile(one, reps))


def test_coincident_and_far_atoms(ctx):
    """Coincident atoms (distance 0), atoms exactly at the search radius and a sparse 2 km box."""
    r0 = np.float32(1.8)
    sr = np.float32(np.float32(r0 + r0) + np.float32(2.8))      # r_i + max_r + 2p
    c = np.array([[0, 0, 0], [0, 0, 0], [float(sr), 0, 0], [0, float(sr) + 1e-3, 0],
                  [2e3, 2e3, 2e3], [2e3 + 1.5, 2e3, 2e3]], np.float32)  # 625^3 cells
    r = np.full(len(c), r0, np.float32)
    ids = np.arange(len(c), dtype=np.uint64)
    got = ctx.calculate_sasa_soa(c[:, 0].copy(), c[:, 1].copy(), c[:, 2].copy(), r, ids, PROBE, 100)
    want = po.calculate_sasa_internal(c[:, 0], c[:, 1], c[:, 2], r, ids, PROBE, 100, 8)
    assert np.array_equal(got, want)


def test_hypothesis_random_clouds(ctx):
    """Random point clouds over a range of densities, radii and probes (seeded; oracle compared)."""
    rng = np.random.default_rng(123)
    for trial in range(12):
        n = int(rng.integers(1, 400))
        box = float(rng.uniform(4.0, 40.0))
        c = rng.uniform(0, box, size=(n, 3)).astype(np.float32)
        r = rng.uniform(0.8, 2.4, size=n).astype(np.float32)
        ids = rng.integers(0, max(2, n // 2 if trial % 3 == 0 else 10 * n), size=n).astype(np.uint64)
        probe = float(rng.choice([0.0, 0.7, 1.4, 2.0]))
        n_points = int(rng.choice([17, 64, 100, 333]))
        got = ctx.calculate_sasa_soa(c[:, 0].copy(), c[:, 1].copy(), c[:, 2].copy(), r, ids, probe, n_points)
        want = po.calculate_sasa_internal(c[:, 0], c[:, 1], c[:, 2], r, ids, probe, n_points, 8)
        assert np.array_equal(got, want), (trial, n, box, probe, n_points)


def test_trajectory_mode(ctx):
    """Frames of one topology = independent structures; radii / ids / residues given once."""
    xyz, r, res, ids = bw.fixture_soa("1jcd.pdb")
    rng = np.random.default_rng(8)
    frames = np.stack([(xyz + rng.normal(scale=0.3, size=xyz.shape)) for _ in range(7)]).astype(np.float32)
    ro = res.astype(np.uint32)
    atom, rsum = ctx.calculate_sasa_trajectory(frames, r, ids, PROBE, 100, residue_offsets=ro)
    assert atom.shape == (7, len(r)) and rsum.shape == (7, len(ro) - 1)
    for f in range(7):
        want = po.calculate_sasa_internal(frames[f, :, 0], frames[f, :, 1], frames[f, :, 2], r, ids,
                                          PROBE, 100, 8)
        assert np.array_equal(atom[f], want)
        assert np.array_equal(rsum[f], po.residue_sums(want, ro))
    only_res = ctx.calculate_sasa_trajectory(frames, r, None, PROBE, 100, residue_offsets=ro,
                                             want_atoms=False)
    assert only_res[0] is None and np.array_equal(only_res[1], rsum)


def test_trajectory_residues_that_do_not_cover_all_atoms(ctx):
    """Residue offsets that start after atom 0 and end before the last atom: every frame's last
    residue must stop at its own end, not run on into the next frame."""
    xyz, r, res, ids = bw.fixture_soa("1jcd.pdb")
    rng = np.random.default_rng(18)
    frames = np.stack([(xyz + rng.normal(scale=0.3, size=xyz.shape)) for _ in range(5)]).astype(np.float32)
    ro = res.astype(np.uint32)[2:-3]          # skips the first two and the last three residues
    assert ro[0] > 0 and ro[-1] < len(r)
    atom, rsum = ctx.calculate_sasa_trajectory(frames, r, ids, PROBE, 100, residue_offsets=ro)
    assert rsum.shape == (5, len(ro) - 1)
    for f in range(5):
        want = po.calculate_sasa_internal(frames[f, :, 0], frames[f, :, 1], frames[f, :, 2], r, ids,
                                          PROBE, 100, 8)
        assert np.array_equal(atom[f], want)
        assert np.array_equal(rsum[f], po.residue_sums(want, ro))
    # the batch entry point with the same kind of offsets
    x, y, z = (np.ascontiguousarray(frames[0, :, k]) for k in range(3))
    _, rs = ctx.calculate_sasa_batch(x, y, z, r, ids, np.array([0, len(r)], np.uint32), PROBE, 100,
                                     residue_offsets=ro)
    assert np.array_equal(rs, rsum[0])


def test_python_wrapper_rejects_mismatched_columns(ctx, example_vdw):
    x, y, z, r, ids = example_vdw
    so = np.array([0, len(x)], np.uint32)
    with pytest.raises(ValueError):
        ctx.calculate_sasa_batch(x, y[:-1], z, r, ids, so)
    with pytest.raises(ValueError):
        ctx.calculate_sasa_batch(x, y, z, r, ids[:10], so)
    with pytest.raises(ValueError):
        ctx.calculate_sasa_batch(x, y, z, r, ids, np.array([0, len(x) + 5], np.uint32))
    with pytest.raises(ValueError):
        ctx.calculate_sasa_soa(x, y, z, r[:5], ids)
    with pytest.raises(ValueError):
        ctx.calculate_sasa_batch(x, y, z, r, ids, so, res_out=np.zeros(3, np.float32),
                                 residue_offsets=np.array([0, 5, len(x)], np.uint32))
    got = ctx.calculate_sasa_soa(x, y, z, r, ids, PROBE, 100)
    assert np.max(np.abs(got - sio.load_golden_low_res())) <= TOL


def test_calls_leave_the_current_device_alone(ctx, example_vdw):
    """Entry points run on the context's device and restore the caller's current device."""
    import torch
    x, y, z, r, ids = example_vdw
    before = torch.cuda.current_device()
    ctx.calculate_sasa_soa(x, y, z, r, ids, PROBE, 100)
    assert torch.cuda.current_device() == before


def test_fast_kernel_defers_dense_atoms_inside_a_mixed_batch(ctx):
    """Kernel 4 = straight-line fast kernel + the general kernel over the atoms it defers (more
    than 256 atoms in the culled runs, or a candidate list past the LDS list).  A batch mixing
    protein-like structures with a dense blob must use both, and still match the oracle."""
    rng = np.random.default_rng(99)
    prot = bw.synthetic_proteome(6, seed=5)
    n_blob = 1500
    blob = rng.normal(scale=4.0, size=(n_blob, 3)).astype(np.float32)      # ~10x protein density
    x = np.concatenate([prot.x, blob[:, 0]])
    y = np.concatenate([prot.y, blob[:, 1]])
    z = np.concatenate([prot.z, blob[:, 2]])
    r = np.concatenate([prot.radius, rng.uniform(1.2, 2.0, n_blob).astype(np.float32)])
    ids = np.arange(len(x), dtype=np.uint64)
    so = np.concatenate([prot.structure_offsets, [len(x)]]).astype(np.uint32)
    b = bw.Batch(x, y, z, r, ids, so, so)
    ctx.enable_timing(True)
    try:
        atom, _, k = _device_run(ctx, b, want_res=False)
        n_deferred = ctx.timings()["n_deferred"]
    finally:
        ctx.enable_timing(False)
    want = po.calculate_sasa_batch(x, y, z, r, ids, so, PROBE, 100, 8, threads=8)
    assert np.array_equal(atom, want)
    assert 0 < n_deferred < len(x)
    _, _, k_blob = po.calculate_sasa_internal(blob[:, 0], blob[:, 1], blob[:, 2], r[-n_blob:], ids[-n_blob:],
                                              PROBE, 100, 8, return_details=True)
    assert np.array_equal(k[-n_blob:], k_blob)
    assert k_blob.max() > 160


@pytest.mark.parametrize("overlap", ["0", "1"])
def test_grid_build_paths_mixed(overlap, monkeypatch):
    """Cell binning has two routes: one workgroup per window of 36 864 cells with 16-bit LDS counters
    (any number of windows; 16-bit cell starts) for structures with fewer than 65 536 atoms, and the
    batch-wide histogram / scan / scatter for the others (32-bit cell starts).  One batch with both,
    interleaved: one window, several, hundreds (a sparse structure: the cell array has to grow and
    the batch runs again), more atoms than a workgroup's registers hold, empty structures.
    (overlap = 1: the batch-wide binning runs on the context's side stream next to the occlusion
    launch over the LDS-binned structures.)"""
    rng = np.random.default_rng(17)
    parts = []
    parts.append(rng.uniform(0, 30, size=(900, 3)))
    parts.append(rng.uniform(0, 1, size=(700, 3)) * np.array([900.0, 40.0, 40.0]))
    parts.append(np.zeros((0, 3)))
    parts.append(rng.uniform(0, 1, size=(800, 3)) * np.array([2500.0, 45.0, 45.0]))     # 6 windows
    parts.append(rng.uniform(0, 1, size=(400, 3)) * np.array([9000.0, 60.0, 60.0]))     # 30 windows
    parts.append(rng.uniform(0, 1, size=(300, 3)) * np.array([30000.0, 100.0, 100.0]))  # 9 M cells: 250 windows
    parts.append(rng.uniform(0, 70, size=(20000, 3)))                                   # 20 000 atoms, 3 windows
    parts.append(rng.uniform(0, 25, size=(500, 3)))
    parts.append(rng.uniform(0, 95, size=(70000, 3)))
    parts.append(np.zeros((0, 3)))
    parts.append(rng.uniform(0, 20, size=(300, 3)))
    xyz = np.concatenate(parts).astype(np.float32)
    so = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.uint32)
    r = rng.uniform(1.2, 2.0, len(xyz)).astype(np.float32)
    ids = np.arange(len(xyz), dtype=np.uint64)
    b = bw.Batch(xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), r, ids, so, so)
    import rustsasa_amd
    monkeypatch.setenv("RSASA_OVERLAP_TAIL", overlap)
    with rustsasa_amd.Context(0) as c:
        for _ in range(3):  # repeated batches reuse the side stream and its events
            atom, _, k = _device_run(c, b, want_res=False)
    want = po.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, so, PROBE, 100, 8, threads=8)
    assert np.array_equal(atom, want)


@pytest.mark.parametrize("dims_z, n_cells", [(16, 36864), (32, 73728)])
def test_grid_of_exactly_whole_windows(dims_z, n_cells):
    """A grid of exactly one / two windows of 36 864 cells: the last window is full and its end
    marker is the word after the window's cells (kernels.hip, k_sort_window)."""
    rng = np.random.default_rng(dims_z)
    # cell size 1.4 + 1.6 = 3.0; dims = ceil(range / 3 + 2) + 1 = 48, 48, dims_z
    box = np.array([134.0, 134.0, 38.0 if dims_z == 16 else 86.0])
    xyz = rng.uniform(0, 1, size=(6000, 3)) * box
    xyz[0] = 0.0
    xyz[1] = box
    xyz = xyz.astype(np.float32)
    r = np.full(len(xyz), 1.6, np.float32)
    so = np.array([0, len(xyz)], np.uint32)
    b = bw.Batch(xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), r, np.arange(len(xyz), dtype=np.uint64), so, so)
    import rustsasa_amd
    with rustsasa_amd.Context(0) as c:
        c.enable_timing(True)
        atom, _, _ = _device_run(c, b, want_res=False)
        assert c.timings()["n_cells"] == n_cells
    want = po.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, so, PROBE, 100, 8, threads=8)
    assert np.array_equal(atom, want)


@pytest.mark.parametrize("n_atoms", [65535, 65536])
def test_largest_structure_binned_in_lds_and_smallest_that_is_not(n_atoms):
    """65 535 atoms: 16-bit positions, k_sort_window; 65 536: the batch-wide kernels."""
    rng = np.random.default_rng(n_atoms)
    xyz = rng.uniform(0, 110, size=(n_atoms, 3)).astype(np.float32)
    r = rng.uniform(1.2, 2.0, n_atoms).astype(np.float32)
    so = np.array([0, n_atoms], np.uint32)
    b = bw.Batch(xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), r, np.arange(n_atoms, dtype=np.uint64), so, so)
    import rustsasa_amd
    with rustsasa_amd.Context(0) as c:
        atom, _, _ = _device_run(c, b, want_res=False)
    want = po.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, so, PROBE, 100, 8, threads=8)
    assert np.array_equal(atom, want)


def test_two_contexts_with_batches_in_flight_at_the_same_time():
    """Two contexts (own workspaces) on two streams, batch k + 1 enqueued before batch k is waited
    for: the kernels of both share the GPU; every batch equals the oracle."""
    import torch
    import rustsasa_amd
    dev = torch.device("cuda:0")
    batches = [bw.synthetic_proteome(60, seed=500 + k) for k in range(4)]
    wants = [po.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, PROBE, 100, 8, threads=8)
             for b in batches]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    dv = [(t(b.x), t(b.y), t(b.z), t(b.radius), t(b.ids.view(np.int64))) for b in batches]
    outs = [torch.full((b.n_atoms,), -1.0, dtype=torch.float32, device=dev) for b in batches]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    with rustsasa_amd.Context(0) as c0, rustsasa_amd.Context(0) as c1:
        ctxs = [c0, c1]
        for rep in range(3):
            for k, b in enumerate(batches):
                if k >= 2:
                    ctxs[k % 2].wait()
                x, y, z, r, ids = dv[k]
                ctxs[k % 2].enqueue_device(x, y, z, r, ids, b.structure_offsets, outs[k], None, None, None, PROBE, 100,
                                           stream=streams[k % 2].cuda_stream)
            c0.wait()
            c1.wait()
            for k in range(4):
                assert np.array_equal(outs[k].cpu().numpy(), wants[k]), (rep, k)


def test_empty_structure_in_a_batch_with_zero_probe(ctx):
    """probe + max radius of an EMPTY structure is 0 when the probe is 0: not an invalid batch."""
    rng = np.random.default_rng(3)
    xyz = rng.uniform(0, 20, (300, 3)).astype(np.float32)
    r = rng.uniform(1.2, 2.0, 300).astype(np.float32)
    so = np.array([0, 0, 120, 120, 300, 300], np.uint32)
    x, y, z = (np.ascontiguousarray(xyz[:, k]) for k in range(3))
    for probe in (0.0, 1.4):
        got, _ = ctx.calculate_sasa_batch(x, y, z, r, None, so, probe, 100)
        want = po.calculate_sasa_batch(x, y, z, r, None, so, probe, 100, 8)
        assert np.array_equal(got, want)


def test_host_threads_share_one_context_and_use_their_own():
    """What rayon workers would do through the Rust shim: threads calling concurrently, (a) every
    thread with its own context, (b) all on ONE context (calls are serialised inside).  Different
    inputs and point counts per thread; every result equals the oracle; errors of one thread stay its own."""
    import threading
    import rustsasa_amd
    rng = np.random.default_rng(77)
    jobs = []
    for t in range(8):
        n = int(rng.integers(200, 3000))
        xyz = rng.uniform(0, (n / 0.05) ** (1 / 3), (n, 3)).astype(np.float32)
        r = rng.uniform(1.2, 2.0, n).astype(np.float32)
        pts = int(rng.choice([20, 100, 200, 960]))
        x, y, z = (np.ascontiguousarray(xyz[:, k]) for k in range(3))
        jobs.append((x, y, z, r, pts, po.calculate_sasa_internal(x, y, z, r, None, PROBE, pts, 8)))

    def run(make_ctx, shared=None):
        errors = []

        def work(t):
            try:
                c = shared if shared is not None else make_ctx()
                x, y, z, r, pts, want = jobs[t]
                for rep in range(20):
                    got = c.calculate_sasa_soa(x, y, z, r, None, PROBE, pts)
                    if not np.array_equal(got, want):
                        errors.append((t, rep, "mismatch"))
                    if rep == 7:  # an invalid call in the middle: the error belongs to this thread
                        try:
                            c.calculate_sasa_soa(x, y, z, r, None, float("nan"), pts)
                            errors.append((t, rep, "no error"))
                        except rustsasa_amd.RsasaError as e:
                            if "probe_radius" not in str(e):
                                errors.append((t, rep, str(e)))
                if shared is None:
                    c.close()
            except Exception as e:  # noqa: BLE001
                errors.append((t, -1, repr(e)))

        ts = [threading.Thread(target=work, args=(t,)) for t in range(len(jobs))]
        for th in ts:
            th.start()
        for th in ts:
            th.join()
        assert not errors, errors[:5]

    run(lambda: rustsasa_amd.Context(0))
    with rustsasa_amd.Context(0) as one:
        run(None, shared=one)


def test_single_structure_calls_of_random_shapes(ctx):
    """The per-structure call (one structure, up to 8 192 atoms: no upload, results written to pinned
    memory by the kernels): random sizes around that limit, shapes, ids, residues and point counts."""
    rng = np.random.default_rng(4242)
    for it in range(60):
        n = int(rng.choice([1, 2, 5, 64, 700, 2600, 8191, 8192, 8193, 12000]))
        kind = it % 4
        if kind == 0:
            xyz = rng.uniform(0, (n / 0.05) ** (1 / 3) + 1.0, (n, 3))
        elif kind == 1:
            xyz = rng.uniform(0, 1, (n, 3)) * np.array([rng.uniform(100, 2000), 30.0, 30.0])  # several windows
        elif kind == 2:
            n = min(n, 700)
            xyz = rng.normal(scale=1.5, size=(n, 3))                                           # one dense cluster (every atom a candidate)
        else:
            xyz = rng.uniform(0, 1, (n, 3)) * rng.uniform(200, 900, 3)                          # sparse
        xyz = (xyz + rng.uniform(-300, 300, 3)).astype(np.float32)
        r = rng.uniform(1.0, 2.1, n).astype(np.float32)
        ids = np.arange(n, dtype=np.uint64) if it % 3 else None
        if ids is not None and n > 3 and it % 2:
            ids[1] = ids[0]
        pts = int(rng.choice([1, 64, 100, 100, 960]))
        so = np.array([0, n], np.uint32)
        ro = np.unique(np.concatenate([[0, n], rng.integers(0, n + 1, size=max(1, n // 8))])).astype(np.uint32)
        x, y, z = (np.ascontiguousarray(xyz[:, k]) for k in range(3))
        got, got_res = ctx.calculate_sasa_batch(x, y, z, r, ids, so, PROBE, pts, residue_offsets=ro)
        want = po.calculate_sasa_internal(x, y, z, r, ids, PROBE, pts, 8, threads=0)
        assert np.array_equal(got, want), (it, n, pts)
        for a, b_, v in zip(ro[:-1], ro[1:], got_res):  # sequential f32 sums (options.rs:209-216)
            acc = np.cumsum(want[a:b_], dtype=np.float32)[-1] if b_ > a else np.float32(0)
            assert v == acc, (it, n, a)


def test_many_tiny_structures(ctx):
    """70 000 structures of 1-3 atoms: the grid placement scan runs over several chunks of
    per-workgroup sums, every structure is its own LDS-binned grid."""
    rng = np.random.default_rng(23)
    counts = rng.integers(1, 4, 70000)
    so = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
    n = int(so[-1])
    xyz = rng.uniform(0, 4, size=(n, 3)).astype(np.float32)
    r = rng.uniform(1.2, 2.0, n).astype(np.float32)
    ids = np.arange(n, dtype=np.uint64)
    b = bw.Batch(xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), r, ids, so, so)
    atom, _, _ = _device_run(ctx, b, want_res=False)
    want = po.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, so, PROBE, 100, 8, threads=8)
    assert np.array_equal(atom, want)


def test_host_batch_pipelined_sub_batches(ctx):
    """rsasa_calculate_sasa_batch cuts a large host batch into sub-batches of whole structures (and
    whole residues) and overlaps their uploads with compute.  The result must equal the
    device-resident single-batch run, also when the residue segments ignore structure boundaries."""
    b = bw.synthetic_proteome(1300, seed=31)
    assert b.n_atoms > 3_000_000
    atom_dev, res_dev, _ = _device_run(ctx, b, want_k=False)
    atom, res = ctx.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, PROBE, 100,
                                         residue_offsets=b.residue_offsets)
    assert np.array_equal(atom, atom_dev) and np.array_equal(res, res_dev)
    # residues only (no per-atom output)
    _, res2 = ctx.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, PROBE, 100,
                                       residue_offsets=b.residue_offsets, want_atoms=False)
    assert np.array_equal(res2, res_dev)
    # segments of 7 atoms across the whole array: most structure boundaries are not segment boundaries
    seg = np.arange(0, b.n_atoms + 1, 7, dtype=np.uint32)
    if seg[-1] != b.n_atoms:
        seg = np.append(seg, np.uint32(b.n_atoms))
    _, res7 = ctx.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, PROBE, 100,
                                       residue_offsets=seg, want_atoms=False)
    want7 = po.residue_sums(atom_dev, seg)
    assert np.array_equal(res7, want7)
    # a few structures against the oracle
    rng = np.random.default_rng(3)
    for s in rng.choice(b.n_structures, 6, replace=False):
        lo, hi = int(b.structure_offsets[s]), int(b.structure_offsets[s + 1])
        want = po.calculate_sasa_internal(*b.structure(int(s)), PROBE, 100, 8)
        assert np.array_equal(atom[lo:hi], want)
    # pinned (page-locked) host arrays in and out: every copy is asynchronous, the results leave on
    # their own stream while the next sub-batch computes
    import torch
    pin = lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory().numpy()  # noqa: E731
    pa, pr = pin(np.full(b.n_atoms, -1.0, np.float32)), pin(np.full(b.n_residues, -1.0, np.float32))
    for _ in range(2):
        a3, r3 = ctx.calculate_sasa_batch(pin(b.x), pin(b.y), pin(b.z), pin(b.radius), pin(b.ids),
                                          b.structure_offsets, PROBE, 100, residue_offsets=pin(b.residue_offsets),
                                          atom_out=pa, res_out=pr)
        assert a3 is pa and r3 is pr
        assert np.array_equal(pa, atom_dev) and np.array_equal(pr, res_dev)
        pa[:] = -1.0
        pr[:] = -1.0
    # pinned residues only, pageable inputs
    _, r4 = ctx.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, PROBE, 100,
                                     residue_offsets=b.residue_offsets, want_atoms=False, res_out=pr)
    assert np.array_equal(r4, res_dev)
    # Ids in pinned memory cross the link as 32-bit folds (the host folds them one sub-batch ahead of the
    # uploads); atoms whose folds collide are decided by the general kernel on the full ids, which it reads
    # from the caller's array.  Duplicated ids (spatial_grid.rs:314, lib.rs:124: same id = same atom): the
    # folded run equals the run with pageable ids (64-bit upload) and the oracle.
    ids2 = b.ids.copy()
    sel = np.arange(5, b.n_atoms - 1, 997)
    ids2[sel + 1] = ids2[sel]
    a5, _ = ctx.calculate_sasa_batch(pin(b.x), pin(b.y), pin(b.z), pin(b.radius), pin(ids2), b.structure_offsets, PROBE, 100)
    a6, _ = ctx.calculate_sasa_batch(b.x, b.y, b.z, b.radius, ids2, b.structure_offsets, PROBE, 100)
    assert np.array_equal(a5, a6)
    assert not np.array_equal(a5, atom_dev)
    for s in rng.choice(b.n_structures, 6, replace=False):
        lo, hi = int(b.structure_offsets[s]), int(b.structure_offsets[s + 1])
        want = po.calculate_sasa_internal(b.x[lo:hi], b.y[lo:hi], b.z[lo:hi], b.radius[lo:hi], ids2[lo:hi], PROBE, 100, 8)
        assert np.array_equal(a5[lo:hi], want)
    # Radii cross the link as one-byte codes into the table of the batch's distinct radii (coded by the same
    # worker threads).  More than 256 distinct radii: the f32 radii are uploaded instead - from the start, or from
    # the sub-batch in which the 257th turns up.
    r_many = b.radius.copy()
    r_many += (rng.integers(0, 4000, b.n_atoms) * np.float32(1e-4)).astype(np.float32)   # thousands of distinct values
    r_late = b.radius.copy()
    late = b.n_atoms * 2 // 3
    r_late[late:] += (rng.integers(0, 400, b.n_atoms - late) * np.float32(1e-3)).astype(np.float32)
    for rr in (r_many, r_late):
        got, _ = ctx.calculate_sasa_batch(pin(b.x), pin(b.y), pin(b.z), pin(rr), pin(b.ids), b.structure_offsets, PROBE, 100)
        for s in rng.choice(b.n_structures, 5, replace=False).tolist() + [b.n_structures - 1]:
            lo, hi = int(b.structure_offsets[s]), int(b.structure_offsets[s + 1])
            want = po.calculate_sasa_internal(b.x[lo:hi], b.y[lo:hi], b.z[lo:hi], rr[lo:hi], b.ids[lo:hi], PROBE, 100, 8)
            assert np.array_equal(got[lo:hi], want)


def test_small_host_batches_take_the_short_path_and_agree(monkeypatch):
    """Host batches of up to 32 768 atoms in up to 256 structures: the host computes the grids and
    the device gets one upload, four launches and one download.  Same results as the general path,
    including batches it has to hand back (empty structure, a grid too large for the LDS windows)."""
    import rustsasa_amd
    b = bw.synthetic_proteome(9, seed=77)
    cut = int(np.searchsorted(b.structure_offsets, 30000, side="right")) - 1
    so = b.structure_offsets[:cut + 1]
    n = int(so[-1])
    ro = b.residue_offsets[:int(np.searchsorted(b.residue_offsets, n, side="right"))]
    args = (b.x[:n], b.y[:n], b.z[:n], b.radius[:n], b.ids[:n], so, PROBE, 100)
    want = po.calculate_sasa_batch(b.x[:n], b.y[:n], b.z[:n], b.radius[:n], b.ids[:n], so, PROBE, 100, 8, threads=4)
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("RSASA_SMALL_PATH", flag)
        with rustsasa_amd.Context(0) as c:
            atom, res = c.calculate_sasa_batch(*args, residue_offsets=ro)
            # an empty structure in the middle / a very elongated structure: general path either way
            so2 = np.array([0, 100, 100, 300], np.uint32)
            a2, _ = c.calculate_sasa_batch(b.x[:300], b.y[:300], b.z[:300], b.radius[:300], b.ids[:300], so2, PROBE, 100)
            x3 = b.x[:400].copy()
            x3[200:] += 30000.0
            a3, _ = c.calculate_sasa_batch(x3, b.y[:400], b.z[:400], b.radius[:400], b.ids[:400],
                                           np.array([0, 400], np.uint32), PROBE, 100)
            outs.append((atom, res, a2, a3))
    assert np.array_equal(outs[0][0], want) and np.array_equal(outs[1][0], want)
    for u, v in zip(outs[0], outs[1]):
        assert np.array_equal(u, v)
    assert np.array_equal(outs[0][1], po.residue_sums(want, ro))


def test_geometry_stress(ctx):
    """Awkward geometry for the run culling (which works in cell units with a rounding tolerance):
    large coordinate offsets (coarse f32 spacing), flat and linear clouds, atoms on cell
    boundaries, huge and tiny radii.  Host small path and device-resident path against the oracle."""
    rng = np.random.default_rng(2024)
    for trial in range(24):
        n = int(rng.integers(2, 600))
        shape = np.array([[30, 30, 30], [60, 60, 0.0], [200, 0.0, 0.0], [12, 12, 12]][trial % 4], float)
        c = rng.uniform(0, 1, size=(n, 3)) * shape
        if trial % 3 == 1:
            c = np.round(c / 3.3) * 3.3          # many atoms exactly on cell-boundary-like positions
        offset = float([0.0, 1.0e3, 1.0e5, -7.7e4][trial % 4 if trial >= 4 else 0])
        c = (c + offset).astype(np.float32)
        r = rng.uniform(0.5, 3.0 if trial % 5 else 6.0, size=n).astype(np.float32)
        ids = np.arange(n, dtype=np.uint64)
        probe = float(rng.choice([0.0, 1.4, 3.0]))
        x, y, z = c[:, 0].copy(), c[:, 1].copy(), c[:, 2].copy()
        want = po.calculate_sasa_internal(x, y, z, r, ids, probe, 100, 8)
        got = ctx.calculate_sasa_soa(x, y, z, r, ids, probe, 100)
        assert np.array_equal(got, want), ("host path", trial, n, offset, probe)
        b = bw.Batch(x, y, z, r, ids, np.array([0, n], np.uint32), np.array([0, n], np.uint32))
        import torch
        dev = torch.device("cuda:0")
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        out = torch.empty(n, dtype=torch.float32, device=dev)
        ctx.enqueue_device(t(x), t(y), t(z), t(r), t(ids.view(np.int64)), b.structure_offsets, out, None, None, None,
                           probe, 100, stream=torch.cuda.current_stream().cuda_stream)
        ctx.wait()
        assert np.array_equal(out.cpu().numpy(), want), ("device path", trial, n, offset, probe)


def test_duplicate_ids_in_device_batches():
    """The id rule on the device-resident path: heavy duplication, sparse duplication, all-equal ids,
    ids that differ in the high word only, and a -0.0 radius must all match the oracle."""
    import rustsasa_amd
    rng = np.random.default_rng(5)
    b = bw.synthetic_proteome(40, seed=13)
    n = b.n_atoms
    variants = {
        "triples": (np.arange(n, dtype=np.uint64) // 3),
        "sparse": np.where(rng.random(n) < 0.01, np.uint64(7), np.arange(n, dtype=np.uint64) + 100),
        "all_equal": np.full(n, 42, np.uint64),
        "high_word_only": (np.arange(n, dtype=np.uint64) % 5) << np.uint64(40),
    }
    with rustsasa_amd.Context(0) as c:
        for name, ids in variants.items():
            bb = bw.Batch(b.x, b.y, b.z, b.radius, ids.astype(np.uint64), b.structure_offsets, b.residue_offsets)
            atom, _, k = _device_run(c, bb, want_res=False)
            want = po.calculate_sasa_batch(b.x, b.y, b.z, b.radius, bb.ids, b.structure_offsets, PROBE, 100, 8, threads=8)
            assert np.array_equal(atom, want), name
        # -0.0 as a radius
        r2 = b.radius.copy()
        r2[5] = np.float32(-0.0)
        bb = bw.Batch(b.x, b.y, b.z, r2, variants["triples"].astype(np.uint64), b.structure_offsets, b.residue_offsets)
        atom, _, _ = _device_run(c, bb, want_res=False)
        want = po.calculate_sasa_batch(b.x, b.y, b.z, r2, bb.ids, b.structure_offsets, PROBE, 100, 8, threads=8)
        assert np.array_equal(atom, want)


def _id_variants(b, rng):
    """Id columns for the check that drops ids which cannot matter (BatchView::ids_check / IdOrder): ones that pass
    it, ones that fail it in a single place - inside a structure, on a 4 096-atom seam of the bounds kernel, on a
    65 536-atom seam of the host's coding blocks - and ones that pass although they fall at every structure's start.
    Returns the columns and the names of those whose repeated id changes a value (most atoms are buried)."""
    n, so = b.n_atoms, b.structure_offsets.astype(np.int64)
    sizes = np.diff(so)
    rising = np.arange(n, dtype=np.uint64) * 3 + 10
    per_structure = np.concatenate([np.arange(m, dtype=np.uint64) for m in sizes])  # serials start over: still all different
    hashed = rising * np.uint64(0x9E3779B97F4A7C15)  # (an odd multiplier: a bijection - all different, in no order)
    out = {"rising": rising, "per_structure": per_structure, "falling": rising[::-1].copy(), "hashed": hashed}
    sensitive = set()
    base_of = {}

    def with_pair(at):
        """Atom `at` takes the id of an EARLIER atom of its structure, so the ids fall exactly at `at` - one within reach,
        and if the oracle says so, one whose absence among `at`'s neighbours changes a value."""
        k = int(np.searchsorted(so, at, side="right") - 1)
        s0, s1 = int(so[k]), int(so[k + 1])
        cols = (b.x[s0:s1], b.y[s0:s1], b.z[s0:s1], b.radius[s0:s1])
        if k not in base_of:
            base_of[k] = po.calculate_sasa_internal(*cols, None, PROBE, 100, 8)
        d2 = (b.x[s0:at] - b.x[at]) ** 2 + (b.y[s0:at] - b.y[at]) ** 2 + (b.z[s0:at] - b.z[at]) ** 2
        ids = rising.copy()
        for j in np.argsort(d2)[:8]:
            ids[at] = rising[s0 + int(j)]
            if not np.array_equal(po.calculate_sasa_internal(*cols, ids[s0:s1], PROBE, 100, 8), base_of[k]):
                return ids, True
        return ids, False

    def first_that_shows(name, candidates):
        ids = None
        for at in candidates:
            ids, shows = with_pair(int(at))
            if shows:
                sensitive.add(name)
                break
        if ids is not None:
            out[name] = ids

    s0, s1 = int(so[1]), int(so[2])
    base1 = po.calculate_sasa_internal(b.x[s0:s1], b.y[s0:s1], b.z[s0:s1], b.radius[s0:s1], None, PROBE, 100, 8)
    first_that_shows("pair_inside", [s0 + int(i) for i in np.flatnonzero(base1 > 5.0) if i > 50][:10])
    first_that_shows("pair_on_bounds_seam", [int(so[k]) + seam for k in np.flatnonzero(sizes > 4200)[:6]
                                             for seam in (4096, 8192) if seam + 50 < sizes[k]])
    if n > 70000:
        first_that_shows("pair_on_block_seam", [65536 if 65536 not in so else 65537])
    first_that_shows("pair_at_the_very_end", [n - 1])
    eq_across = rising.copy()                        # the last atom of a structure and the first of the next: not a pair
    eq_across[so[2]] = eq_across[so[2] - 1]
    out["equal_across_structures"] = eq_across
    for name in [k for k in out if k.startswith("pair")]:  # the same pairs among ids in no order
        ids = out[name]
        out["hashed_" + name] = ids * np.uint64(0x9E3779B97F4A7C15)
        if name in sensitive:
            sensitive.add("hashed_" + name)
    hashed_eq = hashed.copy()
    hashed_eq[so[5]] = hashed_eq[so[3] + 1]          # equal ids in DIFFERENT structures: not a pair either
    out["hashed_equal_in_two_structures"] = hashed_eq
    return out, sensitive


def test_ids_that_cannot_matter_are_dropped_and_ones_that_do_are_not(monkeypatch):
    """Ids only matter where two atoms of one structure share one (lib.rs:127).  The engine checks whether the ids of
    every structure are all different - they rise (one comparison per atom: k_bounds, or the host's coding workers), or,
    for 64-bit ids on the device, a hash table per structure says so (k_ids_distinct) - and runs such a batch as one
    without ids; a single repeated id anywhere must keep the ids in play.  Device-resident batches, one large host
    sub-batch and pipelined host batches with pinned (folded) and pageable ids, every atom against the oracle and the
    drop counter against what should have been dropped."""
    import rustsasa_amd
    import torch
    rng = np.random.default_rng(77)
    b = bw.synthetic_proteome(125, seed=31)
    assert b.n_atoms > 262144 and np.diff(b.structure_offsets).max() > 4200
    variants, sensitive = _id_variants(b, rng)
    assert {"pair_inside", "pair_on_bounds_seam"} <= sensitive  # (pairs that change values: the test can tell whether ids were kept)
    want_none = po.calculate_sasa_batch(b.x, b.y, b.z, b.radius, None, b.structure_offsets, PROBE, 100, 8, threads=0)
    wants = {}
    for name, ids in variants.items():
        wants[name] = po.calculate_sasa_batch(b.x, b.y, b.z, b.radius, ids, b.structure_offsets, PROBE, 100, 8, threads=0)
        assert np.array_equal(wants[name], want_none) == (name not in sensitive), name

    def pin(a):
        return torch.from_numpy(np.ascontiguousarray(a)).pin_memory().numpy()

    rises = ("rising", "per_structure", "equal_across_structures")      # found by one comparison per atom, host or device
    distinct = rises + ("falling", "hashed", "hashed_equal_in_two_structures")  # the others need the device's hash tables
    # (The hash tables for ids in no order are only part of a batch when the context's LAST checked batch had such ids -
    # their workgroups wait for LDS even when they have nothing to do: every variant runs twice, and the counter is
    # checked on the second run; the values on both.)
    with rustsasa_amd.Context(0) as c:
        for name, ids in variants.items():
            bb = bw.Batch(b.x, b.y, b.z, b.radius, ids, b.structure_offsets, b.residue_offsets)
            for second in (False, True):
                n0 = c.ids_dropped()
                atom, _, _ = _device_run(c, bb, want_res=False)
                assert np.array_equal(atom, wants[name]), ("device", name, second)
                n_drop = c.ids_dropped() - n0
                if second or name in rises or name.startswith("pair"):  # (a first run's tables depend on the variant before it)
                    assert n_drop == (1 if name in distinct else 0), ("device", name, second, n_drop)
            for second in (False, True):
                n0 = c.ids_dropped()
                atom, res = c.calculate_sasa_batch(b.x, b.y, b.z, b.radius, ids, b.structure_offsets, PROBE, 100,
                                                   residue_offsets=b.residue_offsets)
                assert np.array_equal(atom, wants[name]), ("host, one sub-batch", name, second)
                assert np.array_equal(res, po.residue_sums(wants[name], b.residue_offsets)), name
                if second:
                    assert c.ids_dropped() - n0 == (1 if name in distinct else 0), ("host, one sub-batch", name)
    monkeypatch.setenv("RSASA_SUB_ATOMS", "60000")  # the pipelined path from 120 k atoms on: several sub-batches here
    with rustsasa_amd.Context(0) as c:
        for name, ids in variants.items():
            for pinned in (True, False):  # (pinned ids are folded and checked by the workers, pageable ones on the device)
                cols = [b.x, b.y, b.z, b.radius, ids]
                if pinned:
                    cols = [pin(a) for a in cols]
                for second in (False, True):
                    n0 = c.ids_dropped()
                    atom, _ = c.calculate_sasa_batch(*cols, b.structure_offsets, PROBE, 100)
                    assert np.array_equal(atom, wants[name]), ("host, pipelined", name, pinned, second)
                n_drop = c.ids_dropped() - n0
                if name in rises or (name in distinct and not pinned):
                    assert n_drop >= 2, (name, pinned, n_drop)      # every sub-batch
                elif name in distinct:
                    # (pinned ids in no order travel as 32-bit folds: the device's tables prove the FOLDS of a structure
                    # different - which proves the ids different - wherever no two folds collide)
                    assert n_drop >= 1, (name, pinned, n_drop)
                elif name.startswith("pair") or not pinned:
                    assert 1 <= n_drop, (name, pinned, n_drop)      # every sub-batch but the pair's
    # a structure too large for the hash table (more than 55 296 atoms: 16-bit entries, 144 KB) keeps ids in no order in play;
    # rising ones still go
    big = bw.synthetic_uniform(60_000, seed=3)
    with rustsasa_amd.Context(0) as c:
        for ids, dropped in ((big.ids * np.uint64(0x9E3779B97F4A7C15), 0), (big.ids * np.uint64(0x9E3779B97F4A7C15), 0), (big.ids, 1)):
            bb = bw.Batch(big.x, big.y, big.z, big.radius, ids, big.structure_offsets, big.residue_offsets)
            atom, _, _ = _device_run(c, bb, want_res=False)
            assert np.array_equal(atom, po.calculate_sasa_batch(big.x, big.y, big.z, big.radius, ids, big.structure_offsets,
                                                                PROBE, 100, 8, threads=0))
            assert c.ids_dropped() == dropped
    # one that the large table takes since its entries are 16 bits wide (40 000 atoms; round 5's table stopped at 27 648): hashes
    # are proven different once the tables are part of the batch, and a repeated id in it is found
    mid = bw.synthetic_uniform(40_000, seed=4)
    hashed = mid.ids * np.uint64(0x9E3779B97F4A7C15)
    rep = hashed.copy()
    rep[39_999] = rep[17]
    with rustsasa_amd.Context(0) as c:
        # (the first batch of a context: launched id-less alone, found to hold ids in no order, run again - by then WITH the tables)
        for ids, kept in ((hashed, None), (hashed, 0), (rep, 1), (hashed, 0)):
            bb = bw.Batch(mid.x, mid.y, mid.z, mid.radius, ids, mid.structure_offsets, mid.residue_offsets)
            atom, _, _ = _device_run(c, bb, want_res=False)
            assert np.array_equal(atom, po.calculate_sasa_batch(mid.x, mid.y, mid.z, mid.radius, ids, mid.structure_offsets,
                                                                PROBE, 100, 8, threads=0))
            assert kept is None or c.ids_kept() == kept, (kept, c.ids_kept())
    monkeypatch.setenv("RSASA_NO_ID_CHECK", "1")    # the switch that turns the check off: same values
    with rustsasa_amd.Context(0) as c:
        for name in ("rising", "pair_inside"):
            bb = bw.Batch(b.x, b.y, b.z, b.radius, variants[name], b.structure_offsets, b.residue_offsets)
            atom, _, _ = _device_run(c, bb, want_res=False)
            assert np.array_equal(atom, wants[name]), ("no check", name)
        assert c.ids_dropped() == 0


# ---- non-finite input (include/rustsasa_amd.h, "Non-finite input") ----------------------------------------------
# The reference has no checks: NaN coordinates fall out of its arithmetic (f32::min / max skip them in the bounds,
# `as u32` sends them to cell 0, every distance to them is NaN and fails every comparison: spatial_grid.rs:113-121,
# 139-141, 321-335), a NaN radius makes that atom's own value NaN (lib.rs:101-102,220-222); an infinite coordinate or
# radius overflows its grid arithmetic (a panic).  Here: NaN behaves as in the oracle, bit for bit, on every entry
# point and kernel, without touching the other atoms or structures of a batch; infinities are the call's error.

def _poison(b, where):
    x, y, z, r = b.x.copy(), b.y.copy(), b.z.copy(), b.radius.copy()
    so = b.structure_offsets
    neg_nan = np.array([0xFFC00000], np.uint32).view(np.float32)[0]
    if "coords" in where:
        x[so[3] + 5] = np.nan            # structure 3: two atoms with NaN coordinates, one of them a negative NaN
        y[so[3] + 40] = neg_nan
        z[so[3] + 40] = np.nan
    if "radius" in where:
        r[so[7] + 11] = np.nan           # structure 7: a NaN radius
    if "all" in where:
        x[so[9]:so[10]] = np.nan         # structure 9: every atom without a position
    return bw.Batch(x, y, z, r, b.ids, b.structure_offsets, b.residue_offsets)


@pytest.mark.parametrize("where", ["coords", "radius", "coords+radius+all"])
@pytest.mark.parametrize("kernel", [None, "5", "3"])
def test_nan_input_matches_the_oracle_and_leaves_the_neighbours_alone(where, kernel, monkeypatch):
    import rustsasa_amd
    if kernel:
        monkeypatch.setenv("RSASA_OCCLUSION_KERNEL", kernel)
    clean = bw.synthetic_proteome(40, seed=17)
    b = _poison(clean, where)
    want = po.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, PROBE, 100, 8, threads=0)
    want_clean = po.calculate_sasa_batch(clean.x, clean.y, clean.z, clean.radius, clean.ids, clean.structure_offsets,
                                         PROBE, 100, 8, threads=0)
    touched = {3} if where == "coords" else {7} if where == "radius" else {3, 7, 9}
    so = b.structure_offsets
    for s in range(b.n_structures):   # the oracle itself: other structures do not notice
        if s not in touched:
            assert np.array_equal(want[so[s]:so[s + 1]], want_clean[so[s]:so[s + 1]])
    with rustsasa_amd.Context(0) as c:
        atom, res, _ = _device_run(c, b, want_k=False)                       # device-resident batch
        assert np.array_equal(atom, want, equal_nan=True)
        assert np.array_equal(res, po.residue_sums(want, b.residue_offsets), equal_nan=True)
        atom_h, res_h = c.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, PROBE, 100,
                                               residue_offsets=b.residue_offsets)   # host batch
        assert np.array_equal(atom_h, want, equal_nan=True) and np.array_equal(res_h, res, equal_nan=True)
        for s in sorted(touched):                                            # one structure per call
            sl = slice(so[s], so[s + 1])
            got = c.calculate_sasa_soa(b.x[sl], b.y[sl], b.z[sl], b.radius[sl], b.ids[sl], PROBE, 100)
            assert np.array_equal(got, want[sl], equal_nan=True), s


def test_infinite_input_is_the_calls_error_and_the_context_lives_on(ctx, example_vdw):
    import rustsasa_amd
    from rustsasa_amd import _capi
    x, y, z, r, ids = example_vdw
    want = po.calculate_sasa_internal(x, y, z, r, ids, PROBE, 100, 8)
    for col in (0, 1, 2):
        for v in (np.inf, -np.inf):
            cols = [x.copy(), y.copy(), z.copy(), r.copy()]
            cols[col][17] = v
            with pytest.raises(rustsasa_amd.RsasaError) as e:
                ctx.calculate_sasa_soa(*cols, ids, PROBE, 100)
            assert e.value.status == _capi.RSASA_ERR_GRID_TOO_LARGE, (col, v, e.value)
            assert np.array_equal(ctx.calculate_sasa_soa(x, y, z, r, ids, PROBE, 100), want)
    # an infinite radius makes the cell size infinite (lib.rs:76): invalid input, like probe + largest radius <= 0
    r_inf = r.copy()
    r_inf[17] = np.inf
    with pytest.raises(rustsasa_amd.RsasaError) as e:
        ctx.calculate_sasa_soa(x, y, z, r_inf, ids, PROBE, 100)
    assert e.value.status == _capi.RSASA_ERR_INVALID_ARGUMENT
    assert np.array_equal(ctx.calculate_sasa_soa(x, y, z, r, ids, PROBE, 100), want)
    # in a batch the error is the batch's (the grids of all its structures are placed by one scan)
    b = bw.synthetic_proteome(40, seed=17)
    xb = b.x.copy()
    xb[b.structure_offsets[2] + 1] = np.inf
    bad = bw.Batch(xb, b.y, b.z, b.radius, b.ids, b.structure_offsets, b.residue_offsets)
    with pytest.raises(rustsasa_amd.RsasaError) as e:
        _device_run(ctx, bad, want_k=False)
    assert e.value.status == _capi.RSASA_ERR_GRID_TOO_LARGE
    atom, _, _ = _device_run(ctx, b, want_k=False)
    assert np.array_equal(atom, po.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, PROBE, 100, 8, threads=0))


def test_real_coordinates_tiled_batch_matches_the_oracle(ctx):
    """bench.py's `real_coords` leg at test size: the reference's quality set (tests/quality.rs:200-258 - 87 real
    structures as the reader selects them: whole complexes of up to 32 500 atoms, ProtOr radii, the reader's hashed ids
    in no order) plus two copies of it under rigid motions (real_coords.tiled: 1.37 M atoms, 261 structures), through
    the device-resident batch path - the default k_occlusion_mx dispatch with the id tables - against the oracle:
    every atom, every candidate count, every chain sum; then once more, as the second batch of the context (the id
    tables are only part of a batch once the context has seen ids in no order)."""
    import real_coords as rc
    base = rc.quality_set_batch()
    assert base.n_structures == 87 and base.n_atoms > 450_000
    b = rc.tiled(base, 3 * base.n_atoms, seed=7)
    assert b.n_structures == 3 * 87
    want, _, want_k = None, None, None
    want = po.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, PROBE, 100, 8, threads=0)
    for _ in range(2):
        atom, res, k = _device_run(ctx, b)
        bad = np.flatnonzero(atom != want)
        assert bad.size == 0, (bad.size, bad[:5], atom[bad[:5]], want[bad[:5]])
        assert np.array_equal(res, po.residue_sums(want, b.residue_offsets))
    # candidate counts of the files' own frame against the oracle's lists, structure by structure (a sample)
    so = b.structure_offsets
    for s in (0, 17, 40, 86):
        x, y, z, r, ids = b.structure(s)
        _, _, wk = po.calculate_sasa_internal(x, y, z, r, ids, PROBE, 100, 8, return_details=True)
        assert np.array_equal(k[so[s]:so[s + 1]], wk)
    assert 30.0 < k.mean() < 60.0


def test_rank_shard_of_the_eight_way_split(ctx):
    """BASELINE.json configs[3] at one rank's size: rank 0's shard of the 8-way largest-first split of the proteome
    (bench.py --shard-of 8: 545 structures, 1.47 M atoms) - what every GPU of an 8-GPU node computes -, through the
    device-resident stepping with two batches in flight (the way a rank steps) AND as a stream of host batches
    (rsasa_host_batch_enqueue), every atom and residue against the oracle."""
    import torch
    full = bw.synthetic_proteome()
    sizes = np.diff(full.structure_offsets.astype(np.int64))
    parts = bw.shard_largest_first(sizes, 8)
    assert sorted(np.concatenate(parts).tolist()) == list(range(full.n_structures))
    b = bw.select(full, parts[0])
    assert b.n_structures == 545 and 1_400_000 < b.n_atoms < 1_550_000
    want = po.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, PROBE, 100, 8, threads=0)
    want_res = po.residue_sums(want, b.residue_offsets)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    x, y, z, r, ids, ro = t(b.x), t(b.y), t(b.z), t(b.radius), t(b.ids.view(np.int64)), t(b.residue_offsets.view(np.int32))
    outs = [(torch.full((b.n_atoms,), -1.0, dtype=torch.float32, device=dev),
             torch.full((b.n_residues,), -1.0, dtype=torch.float32, device=dev)) for _ in range(2)]
    torch.cuda.synchronize()
    ctx.enqueue_device(x, y, z, r, ids, b.structure_offsets, outs[0][0], ro, outs[0][1], None, PROBE, 100)
    for i in range(1, 6):  # step k + 1 is enqueued before step k is waited for
        ctx.enqueue_device(x, y, z, r, ids, b.structure_offsets, outs[i % 2][0], ro, outs[i % 2][1], None, PROBE, 100)
        ctx.wait()
        a, rs = outs[(i - 1) % 2]
        assert np.array_equal(a.cpu().numpy(), want) and np.array_equal(rs.cpu().numpy(), want_res)
        a.fill_(-1.0)
        rs.fill_(-1.0)
    ctx.wait()
    assert np.array_equal(outs[1][0].cpu().numpy(), want) and np.array_equal(outs[1][1].cpu().numpy(), want_res)
    # the same shard as a stream of host batches: three enqueued, then collected oldest first
    got = [ctx.host_batch_enqueue(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, PROBE, 100,
                                  residue_offsets=b.residue_offsets) for _ in range(3)]
    ctx.host_batch_wait_all()
    for a, rs in got:
        assert np.array_equal(a, want) and np.array_equal(rs, want_res)


def test_the_verdict_on_ids_is_each_structures_own(monkeypatch):
    """ABI 4: a structure whose ids repeat (a file whose serial numbers start over, an altloc pair hashed alike) keeps its
    ids and runs in the kernels' instantiation with ids; the other structures of the same batch run without theirs -
    a single such structure used to put the whole batch on the slower path (found on the reference's quality set, whose
    real files hold repeated serials).  Rising ids and hashes, device-resident batches at 100 points and at 960 (the
    persistent waves' block counters serve both launches of the pair), a host batch with pinned hashes cut into
    sub-batches: every atom against the oracle, and the count of structures that kept their ids."""
    import rustsasa_amd
    import torch
    b = bw.synthetic_proteome(125, seed=31)
    so = b.structure_offsets.astype(np.int64)
    rising = b.ids.copy()
    hashed = b.ids * np.uint64(0x9E3779B97F4A7C15)

    shown = {}

    def pair_that_shows(s):
        """(at, j): atom `at` of structure s takes the id of its near neighbour j, and a value of the structure changes"""
        if s not in shown:
            s0, s1 = int(so[s]), int(so[s + 1])
            cols = (b.x[s0:s1], b.y[s0:s1], b.z[s0:s1], b.radius[s0:s1])
            base = po.calculate_sasa_internal(*cols, None, PROBE, 100, 8)
            for at in np.flatnonzero(base > 5.0)[10:40]:
                d2 = (cols[0] - cols[0][at]) ** 2 + (cols[1] - cols[1][at]) ** 2 + (cols[2] - cols[2][at]) ** 2
                d2[at] = np.inf
                for j in np.argsort(d2)[:6]:
                    ids = np.arange(1, s1 - s0 + 1, dtype=np.uint64)
                    ids[at] = ids[j]
                    if not np.array_equal(po.calculate_sasa_internal(*cols, ids, PROBE, 100, 8), base):
                        shown[s] = (s0 + int(at), s0 + int(j))
                        break
                if s in shown:
                    break
            assert s in shown, s
        return shown[s]

    def with_pairs(ids, structures):
        ids = ids.copy()
        for s in structures:  # an atom takes a near neighbour's id, chosen so that a value changes: the test can tell
            at, j = pair_that_shows(s)
            ids[at] = ids[j]
        return ids

    cases = [("rising, pairs in 2 structures", with_pairs(rising, (7, 124)), 2), ("hashed, pair in the last structure", with_pairs(hashed, (124,)), 1),
             ("hashed, none", hashed, 0), ("hashed, pairs in 3 structures", with_pairs(hashed, (0, 60, 124)), 3)]
    wants = {}
    for name, ids, _ in cases:
        for n_points in (100, 960):
            wants[(name, n_points)] = po.calculate_sasa_batch(b.x, b.y, b.z, b.radius, ids, b.structure_offsets, PROBE, n_points, 8, threads=0)
    none = po.calculate_sasa_batch(b.x, b.y, b.z, b.radius, None, b.structure_offsets, PROBE, 100, 8, threads=0)
    assert not np.array_equal(wants[(cases[0][0], 100)], none)  # (the pairs change values: the test can tell)
    with rustsasa_amd.Context(0) as c:
        for name, ids, kept in cases:
            bb = bw.Batch(b.x, b.y, b.z, b.radius, ids, b.structure_offsets, b.residue_offsets)
            for n_points in (100, 960):
                for second in (False, True):  # (the hash tables join a batch once the context has seen ids in no order)
                    n0 = c.ids_dropped()
                    atom, _, _ = _device_run(c, bb, n_points=n_points, want_res=False)
                    bad = np.flatnonzero(atom != wants[(name, n_points)])
                    assert bad.size == 0, (name, n_points, second, bad.size, np.searchsorted(so, bad[:5], side="right") - 1)
                assert c.ids_kept() == kept, (name, n_points, c.ids_kept())
                assert c.ids_dropped() - n0 == (1 if kept == 0 else 0), (name, n_points)

    def pin(a):
        return torch.from_numpy(np.ascontiguousarray(a)).pin_memory().numpy()

    monkeypatch.setenv("RSASA_SUB_ATOMS", "60000")  # (read per call under RSASA_TUNING=1: several sub-batches of this batch)
    with rustsasa_amd.Context(0) as c:
        px, py, pz, pr = (pin(a) for a in (b.x, b.y, b.z, b.radius))
        for name, ids, kept in cases[1:3]:
            pid = pin(ids)
            for second in (False, True):
                atom, _ = c.calculate_sasa_batch(px, py, pz, pr, pid, b.structure_offsets, PROBE, 100)
                assert np.array_equal(atom, wants[(name, 100)]), ("host, pipelined, pinned hashes", name, second)
            assert c.ids_kept() == kept, (name, c.ids_kept())  # (of the last sub-batch, which holds the last structure)
