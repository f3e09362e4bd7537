"""Real, diverse coordinates for bench.py's `real_coords` leg and its parity test.

The AlphaFold E. coli proteome is not available offline; what the repo does hold of real structures is the reference's
own quality set (tests/golden/freesasa_set.tar.xz: the 88 PDB files of tests/quality.rs:200-258 - single chains and
complexes of up to 30 000 atoms, with alternate locations, hydrogens, ligands).  This module turns them into a batch the
way the reference's directory mode would see them (reference src/main.rs:342-480, src/options.rs:139-190):

    quality_set_batch()    every file through the C++ reader's selection (sasa_host_cli select: first conformer, no
                           hydrogens / HETATMs, ProtOr radii, FNV ids - what SASAOptions::process hands to the hot path),
                           WHOLE structures, nothing fragmented or re-packed: 87 structures, 458 k atoms;
    tiled(batch, n_atoms)  copies of that set, every copy of every structure under its own rigid motion (a random
                           rotation changes how the atoms fall into the cell grid, so no two copies cost the same),
                           coordinates at PDB text precision, until the batch has about n_atoms atoms.

Nothing here is product code; numpy only (+ the engine's own command-line driver for the reader)."""
from __future__ import annotations

import json
import os
import subprocess
import tarfile
import tempfile

import numpy as np

import bench_workloads as bw

ROOT = os.path.dirname(os.path.abspath(__file__))
ARCHIVE = os.path.join(ROOT, "tests", "golden", "freesasa_set.tar.xz")
CLI = os.path.join(ROOT, "rustsasa_amd", "lib", "sasa_host_cli")
SEED = 20261004

_cache = None


def quality_set_batch() -> bw.Batch:
    """The reader's selection of every file of the reference's quality set as one batch (one residue entry per chain:
    the leg reads atom values; ids are the reader's own 64-bit hashes, in no order)."""
    global _cache
    if _cache is not None:
        return _cache
    if not os.path.exists(CLI):
        raise RuntimeError(f"{CLI} is not built (make -C rustsasa_amd/csrc)")
    X, R, I, S, RES = [], [], [], [0], [0]
    names = []
    with tempfile.TemporaryDirectory(prefix="rsasa_real_") as d:
        with tarfile.open(ARCHIVE, "r:xz") as tar:
            tar.extractall(d)
        pdb_dir = os.path.join(d, "freesasa_pdbs")
        for f in sorted(os.listdir(pdb_dir)):
            if not f.endswith(".pdb"):
                continue
            p = subprocess.run([CLI, "select", os.path.join(pdb_dir, f)], capture_output=True, text=True)
            if p.returncode != 0:
                continue  # (3sqz: the reference's own "Failed to get residue name")
            sel = json.loads(p.stdout)
            a = sel["atoms"]
            if not a:
                continue
            X.append(np.array([v[:3] for v in a], np.float32))
            R.append(np.array([v[3] for v in a], np.float32))
            I.append(np.array([int(v[4]) for v in a], np.uint64))
            RES.extend(S[-1] + int(e) for e in sel["chain_end"] if int(e) > (RES[-1] - S[-1]))
            S.append(S[-1] + len(a))
            if RES[-1] != S[-1]:
                RES.append(S[-1])
            names.append(f[:-4])
    xyz = np.concatenate(X)
    b = bw.Batch(np.ascontiguousarray(xyz[:, 0]), np.ascontiguousarray(xyz[:, 1]), np.ascontiguousarray(xyz[:, 2]),
                 np.concatenate(R), np.concatenate(I), np.array(S, np.uint32), np.array(RES, np.uint32))
    b.names = names
    _cache = b
    return b


def tiled(base: bw.Batch, n_atoms: int, seed: int = SEED) -> bw.Batch:
    """Copies of `base` until about n_atoms atoms, each structure of each copy under its own rotation + translation."""
    rng = np.random.default_rng(seed)
    n_tiles = max(1, int(round(n_atoms / base.n_atoms)))
    so = base.structure_offsets.astype(np.int64)
    ro = base.residue_offsets.astype(np.int64)
    X, S, RES = [], [0], [0]
    for t in range(n_tiles):
        for s in range(base.n_structures):
            b, e = so[s], so[s + 1]
            xyz = np.stack([base.x[b:e], base.y[b:e], base.z[b:e]], 1).astype(np.float64)
            if t > 0:  # (the first copy is the files' own frame)
                c = xyz.mean(axis=0)
                xyz = (xyz - c) @ bw._random_rotation(rng).T + c + rng.uniform(-40, 40, size=3)
                xyz = np.round(xyz, 3)  # PDB text precision
            X.append(xyz.astype(np.float32))
            r0, r1 = np.searchsorted(ro, b), np.searchsorted(ro, e)
            RES.extend((ro[r0 + 1:r1 + 1] - b + S[-1]).tolist())
            S.append(S[-1] + (e - b))
    xyz = np.concatenate(X)
    return bw.Batch(np.ascontiguousarray(xyz[:, 0]), np.ascontiguousarray(xyz[:, 1]), np.ascontiguousarray(xyz[:, 2]),
                    np.tile(base.radius, n_tiles), np.tile(base.ids, n_tiles), np.array(S, np.uint32), np.array(RES, np.uint32))
