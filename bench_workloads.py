"""Seeded synthetic workloads for bench.py and the parity tests.

The AlphaFold E. coli proteome is not available offline, so the proteome
configuration is synthesised from the reference's own test structures
(tests/golden/data): every synthetic structure is a set of residue-aligned
fragments of those proteins, each randomly rotated, jittered (sigma 0.05 A) and
placed in its own lattice slot, with ProtOr radii and real residue boundaries.
That keeps the local geometry -- bond lengths, ~43 candidates per atom, burial
statistics -- protein-like, which is what the occlusion kernel's cost depends on.

Nothing here is product code; it needs numpy only.
"""
from __future__ import annotations

import os
import sys
from dataclasses import dataclass

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import structio as sio  # noqa: E402

PROTEOME_SEED = 20260807
PROTEOME_STRUCTURES = 4363
FIXTURES = ("example.cif", "bad_seqadv_1A06.pdb", "151L_H3.pdb", "1jcd.pdb")


@dataclass
class Batch:
    """Concatenated SoA of independent structures (what the batch C ABI takes)."""
    x: np.ndarray
    y: np.ndarray
    z: np.ndarray
    radius: np.ndarray
    ids: np.ndarray                # uint64, unique per structure
    structure_offsets: np.ndarray  # uint32 [S + 1]
    residue_offsets: np.ndarray    # uint32 [R + 1], global atom offsets

    @property
    def n_atoms(self):
        return int(self.x.shape[0])

    @property
    def n_structures(self):
        return int(self.structure_offsets.shape[0] - 1)

    @property
    def n_residues(self):
        return int(self.residue_offsets.shape[0] - 1)

    def structure(self, s):
        b, e = int(self.structure_offsets[s]), int(self.structure_offsets[s + 1])
        return self.x[b:e], self.y[b:e], self.z[b:e], self.radius[b:e], self.ids[b:e]


@dataclass
class Domain:
    xyz: np.ndarray       # float64 [n, 3], centred
    radius: np.ndarray    # float32 [n]
    res_start: np.ndarray  # int64 [n_res + 1] atom offsets of residues
    extent: float


_domains = None


def fixture_soa(name, radii="protor"):
    """Heavy, non-HETATM atoms of a fixture with ProtOr (or vdW) radii + residue offsets."""
    atoms = [a for a in sio.read_structure(sio.data_path(name))
             if not a.hetero and a.element != "H"]
    if radii == "protor":
        tab = sio.parse_protor(sio.data_path("protor.config"))
        r = np.array([tab[(a.resname, a.name)] for a in atoms], np.float32)
    else:
        r = np.array([sio.VDW[a.element] for a in atoms], np.float32)
    xyz = np.array([[a.x, a.y, a.z] for a in atoms], np.float64)
    keys = [(a.chain, a.resseq, a.icode) for a in atoms]
    starts = [0] + [i for i in range(1, len(keys)) if keys[i] != keys[i - 1]] + [len(keys)]
    ids = np.array([a.serial for a in atoms], np.uint64)
    return xyz, r, np.array(starts, np.int64), ids


def load_domains():
    global _domains
    if _domains is None:
        _domains = []
        for name in FIXTURES:
            xyz, r, starts, _ = fixture_soa(name)
            c = xyz.mean(axis=0)
            xyz = xyz - c
            _domains.append(Domain(xyz, r, starts, float(np.abs(xyz).max())))
    return _domains


def _random_rotation(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    a, b, c, d = q
    return np.array([
        [a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
        [2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b)],
        [2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d]])


def synthetic_structure(n_target, rng, domains=None):
    """One protein-like structure of about n_target heavy atoms."""
    domains = domains or load_domains()
    xs, rs, res = [], [], [0]
    n = 0
    slot = 0
    while n < n_target:
        d = domains[rng.integers(len(domains))]
        n_res = len(d.res_start) - 1
        want = n_target - n
        # residue-aligned fragment of at most `want` atoms: the whole domain when it fits,
        # otherwise a contiguous stretch that starts early enough to hold `want` atoms
        if want >= d.res_start[-1]:
            r0 = 0
        else:
            last = int(np.searchsorted(d.res_start, d.res_start[-1] - want, side="right") - 1)
            r0 = int(rng.integers(max(last, 0) + 1))
        a0 = d.res_start[r0]
        r1 = int(np.searchsorted(d.res_start, a0 + want, side="right") - 1)
        r1 = max(r1, r0 + 1)
        r1 = min(r1, n_res)
        a1 = d.res_start[r1]
        frag = d.xyz[a0:a1]
        frag = frag - frag.mean(axis=0)
        frag = frag @ _random_rotation(rng).T + rng.normal(scale=0.05, size=frag.shape)
        # fragments sit in separate slots of a 90 A lattice (no inter-fragment contacts)
        gx, gy, gz = slot % 4, (slot // 4) % 4, slot // 16
        frag = frag + np.array([gx, gy, gz]) * 90.0 + rng.uniform(-3, 3, size=3)
        slot += 1
        xs.append(frag)
        rs.append(d.radius[a0:a1])
        res.extend((d.res_start[r0 + 1:r1 + 1] - a0 + n).tolist())
        n += a1 - a0
    xyz = np.concatenate(xs)
    xyz = np.round(xyz + rng.uniform(-50, 50, size=3), 3)  # PDB text precision
    return xyz.astype(np.float32), np.concatenate(rs), np.array(res, np.int64)


def synthetic_proteome(n_structures=PROTEOME_STRUCTURES, seed=PROTEOME_SEED,
                       median=2000.0, sigma=0.75, lo=150, hi=25000) -> Batch:
    """AF-proteome-like batch: lognormal heavy-atom counts clipped to [lo, hi]."""
    rng = np.random.default_rng(seed)
    sizes = np.clip(rng.lognormal(np.log(median), sigma, n_structures), lo, hi).astype(np.int64)
    domains = load_domains()
    X, R, S, RES = [], [], [0], [0]
    total = 0
    for n_t in sizes:
        xyz, r, res = synthetic_structure(int(n_t), rng, domains)
        X.append(xyz)
        R.append(r)
        RES.extend((res[1:] + total).tolist())
        total += xyz.shape[0]
        S.append(total)
    xyz = np.concatenate(X)
    so = np.array(S, np.uint32)
    ids = np.concatenate([np.arange(1, S[i + 1] - S[i] + 1, dtype=np.uint64)
                          for i in range(n_structures)])
    return Batch(np.ascontiguousarray(xyz[:, 0]), np.ascontiguousarray(xyz[:, 1]),
                 np.ascontiguousarray(xyz[:, 2]), np.concatenate(R).astype(np.float32), ids, so,
                 np.array(RES, np.uint32))


PROTOR_MIX = ((1.88, .45), (1.61, .15), (1.76, .05), (1.64, .17), (1.42, .12), (1.46, .05),
              (1.77, .01))


def synthetic_uniform(n_atoms=1_000_000, seed=5, density=0.05, jitter=0.7) -> Batch:
    """Config 5: one big structure, jittered lattice at `density` atoms/A^3
    (minimum separation >= spacing - 2*jitter = 1.31 A), ProtOr radii mix."""
    rng = np.random.default_rng(seed)
    a = (1.0 / density) ** (1.0 / 3.0)
    side = int(np.ceil(n_atoms ** (1.0 / 3.0)))
    g = np.stack(np.meshgrid(*(np.arange(side),) * 3, indexing="ij"), -1).reshape(-1, 3)
    g = g[rng.permutation(g.shape[0])[:n_atoms]].astype(np.float64) * a
    xyz = np.round(g + rng.uniform(-jitter, jitter, size=g.shape), 3).astype(np.float32)
    vals = np.array([v for v, _ in PROTOR_MIX], np.float32)
    p = np.array([w for _, w in PROTOR_MIX])
    r = vals[rng.choice(len(vals), size=n_atoms, p=p / p.sum())]
    ids = np.arange(1, n_atoms + 1, dtype=np.uint64)
    res = np.arange(0, n_atoms + 1, 8, dtype=np.uint32)
    if res[-1] != n_atoms:
        res = np.append(res, np.uint32(n_atoms))
    return Batch(np.ascontiguousarray(xyz[:, 0]), np.ascontiguousarray(xyz[:, 1]),
                 np.ascontiguousarray(xyz[:, 2]), r, ids, np.array([0, n_atoms], np.uint32), res)


def select(batch: Batch, sel) -> Batch:
    """The structures `sel` (indices, in that order) as a batch of their own."""
    sel = np.asarray(sel, np.int64)
    so = batch.structure_offsets.astype(np.int64)
    ro = batch.residue_offsets.astype(np.int64)
    idx, new_so, new_ro = [], [0], [0]
    for s in sel:
        b, e = so[s], so[s + 1]
        idx.append(np.arange(b, e))
        r0, r1 = np.searchsorted(ro, b), np.searchsorted(ro, e)
        new_ro.extend((ro[r0 + 1:r1 + 1] - b + new_so[-1]).tolist())
        new_so.append(new_so[-1] + (e - b))
    idx = np.concatenate(idx) if idx else np.zeros(0, np.int64)
    return Batch(batch.x[idx], batch.y[idx], batch.z[idx], batch.radius[idx], batch.ids[idx],
                 np.array(new_so, np.uint32), np.array(new_ro, np.uint32))


def shard(batch: Batch, rank: int, world: int) -> Batch:
    """Structures rank, rank + world, ... (sizes are i.i.d., so shards are balanced)."""
    return select(batch, np.arange(rank, batch.n_structures, world))


def shard_largest_first(sizes, world: int):
    """Strong-scaling partition of SURVEY 8e: structures sorted by atom count, largest first, each
    handed to the rank with the fewest atoms so far (what a dynamic largest-first queue converges
    to when the cost of a structure is its atom count).  Returns one index array per rank, each
    in descending size order; together they are a partition of range(len(sizes))."""
    sizes = np.asarray(sizes, np.int64)
    order = np.argsort(-sizes, kind="stable")
    load = np.zeros(world, np.int64)
    parts = [[] for _ in range(world)]
    for s in order:
        r = int(np.argmin(load))  # ties: lowest rank
        parts[r].append(int(s))
        load[r] += sizes[s]
    return [np.array(p, np.int64) for p in parts]
