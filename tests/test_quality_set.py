"""The reference's whole quality gate (tests/quality.rs:17-18,200-258,340-442): its 88 FreeSASA files through
`SASAOptions::process_files` - ONE batch on the GPU, which takes the default `k_occlusion_mx` dispatch on real
coordinates, alternate locations, hydrogens and multi-chain complexes - with the RMSE of the chain totals against
FreeSASA's (Lee & Richards) at most 43.99 + 20 at all four output levels and with the radii read from the occupancy
column, and every atom of every file bit-equal to the oracle on the same selection.

The files travel as tests/golden/freesasa_set.tar.xz (the reference's own test inputs and FreeSASA's outputs: data;
tests/golden/make_golden.py writes the archive).  Without a GPU the same gate runs on reader + selection + oracle."""
import json
import os
import subprocess
import tarfile

import numpy as np
import pytest

import structio as sio
from oracle import pyoracle as po
from conftest import ensure_built

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = ensure_built()
ARCHIVE = os.path.join(ROOT, "tests", "golden", "freesasa_set.tar.xz")
RMSE_BASELINE, TOLERANCE = 43.99, 20.0   # tests/quality.rs:17-18
RMSE_GATE = RMSE_BASELINE + TOLERANCE    # tests/quality.rs:225
# the one file of the set that stops with the reference's own error (a residue whose conformers carry different
# names: "Failed to get residue name", src/options.rs:255-262): it has no output file on either side
KNOWN_ERRORS = {"3sqz": 5}


@pytest.fixture(scope="module")
def quality_set(tmp_path_factory):
    d = tmp_path_factory.mktemp("freesasa_set")
    with tarfile.open(ARCHIVE, "r:xz") as tar:
        tar.extractall(d)
    pdb_dir, ref_dir = os.path.join(d, "freesasa_pdbs"), os.path.join(d, "freesasa_reference")
    ids = sorted(f[:-4] for f in os.listdir(pdb_dir) if f.endswith(".pdb"))
    assert len(ids) == 88 and all(os.path.exists(os.path.join(ref_dir, i + ".json")) for i in ids)
    return {"dir": str(d), "ids": ids, "pdb": lambda i: os.path.join(pdb_dir, i + ".pdb"),
            "ref": lambda i: os.path.join(ref_dir, i + ".json")}


def freesasa_chains(path, file_total):
    """load_freesasa_chains (tests/quality.rs:20-58)"""
    ref = json.load(open(path))
    chains = [(c["label"], c["area"]["total"]) for r in ref["results"] for s in r["structure"] for c in s["chains"]]
    if file_total:
        return {os.path.basename(path)[:-5]: sum(v for _, v in chains)}
    return dict(chains)  # (a label seen twice: the last one wins, as in the reference's HashMap)


def rmse_of(pairs):
    a = np.array([p[0] for p in pairs], np.float64)
    b = np.array([p[1] for p in pairs], np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)))


def select(path, *opts):
    p = subprocess.run([CLI, "select", path, *opts], capture_output=True, text=True)
    if p.returncode != 0:
        return None
    d = json.loads(p.stdout)
    a = d["atoms"]
    cols = [np.array([v[k] for v in a], np.float32) for k in range(4)]
    return cols, np.array([int(v[4]) for v in a], np.uint64), d


def with_radii_in_occupancy(src, dst):
    """prepare_pdbs_with_radii_in_occupancy (tests/quality.rs:262-334): every atom's occupancy becomes its ProtOr
    radius (the element's van-der-Waals radius where the table has none)."""
    tab = sio.parse_protor(sio.data_path("protor.config"))
    out = []
    for line in open(src):
        if line.startswith(("ATOM", "HETATM")) and len(line) >= 60:
            r = tab.get((line[17:20].strip(), line[12:16].strip()))
            if r is None:
                r = sio.VDW.get(line[76:78].strip().upper() if len(line) >= 78 else "", 2.0)
            line = line[:54] + f"{r:6.2f}" + line[60:]
        out.append(line)
    open(dst, "w").write("".join(out))


def run_files(level, lst, *opts):
    p = subprocess.run([CLI, "files", level, lst, "--full", "--labels", "--threads", "8", "--batch", "0", *opts],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[:500]
    return json.loads(p.stdout)


def write_list(tmp_path, qs, name="files.txt", prepare=None):
    paths = []
    for i in qs["ids"]:
        src = qs["pdb"](i)
        if prepare:
            dst = str(tmp_path / (i + ".pdb"))
            prepare(src, dst)
            src = dst
        paths.append(src)
    lst = str(tmp_path / name)
    open(lst, "w").write("\n".join(paths) + "\n")
    return paths, lst


def test_archive_holds_the_references_whole_set(quality_set):
    qs = quality_set
    n_chains = sum(len(freesasa_chains(qs["ref"](i), False)) for i in qs["ids"])
    assert n_chains >= 160
    # the fixtures committed one by one are members of the set, byte for byte
    for pid in ("2gpi", "4c1a", "1jcd"):
        a = open(qs["pdb"](pid), "rb").read()
        b = open(sio.data_path(f"freesasa/{pid}.pdb" if pid != "1jcd" else "1jcd.pdb"), "rb").read()
        assert a == b, pid


def test_whole_set_meets_the_gate_on_reader_and_oracle(quality_set):
    """No GPU: C++ reader + selection (`select`) feeding the oracle, chain totals against FreeSASA."""
    qs = quality_set
    pairs, errors = [], {}
    for i in qs["ids"]:
        got = select(qs["pdb"](i))
        if got is None:
            errors[i] = True
            continue
        (x, y, z, r), ids, d = got
        atom = po.calculate_sasa_internal(x, y, z, r, ids, 1.4, 100, 8, threads=0)
        sums = po.residue_sums(atom, np.array([0] + d["chain_end"], np.uint32))
        ours = dict(zip(d["chains"], (float(v) for v in sums)))
        want = freesasa_chains(qs["ref"](i), False)
        pairs += [(ours[k], want[k]) for k in sorted(set(ours) & set(want))]
    assert set(errors) == set(KNOWN_ERRORS)
    rmse = rmse_of(pairs)
    assert len(pairs) >= 160 and rmse <= RMSE_GATE, rmse
    assert abs(rmse - RMSE_BASELINE) < 0.5, rmse   # the reference's own figure (tests/quality.rs:17) - not only its gate


@pytest.mark.gpu
def test_whole_set_in_one_batch_is_bit_equal_to_the_oracle_and_meets_the_gate(quality_set, tmp_path):
    """All 88 files in ONE `process_files` batch (about 0.6 M atoms: the matrix-core kernel's default dispatch) at the
    atom level: every atom equals the oracle on the reader's own selection; then the gate at all four levels."""
    qs = quality_set
    paths, lst = write_list(tmp_path, qs)
    got = run_files("atom", lst)
    assert got["n_files"] == 88 and got["n_atoms"] >= 32768  # (k_occlusion_mx takes batches of 32 768 atoms or more)
    n_compared, file_pairs = 0, []
    for i, r in zip(qs["ids"], got["results"]):
        if i in KNOWN_ERRORS:
            assert r == {"error": KNOWN_ERRORS[i]}, (i, r)
            continue
        (x, y, z, rad), ids, _ = select(qs["pdb"](i))
        want = po.calculate_sasa_internal(x, y, z, rad, ids, 1.4, 100, 8, threads=0)
        assert np.array_equal(np.array(r, np.float32), want), i
        n_compared += len(want)
        file_pairs.append((float(np.sum(np.array(r, np.float64))), freesasa_chains(qs["ref"](i), True)[i]))
    assert 0.99 * got["n_atoms"] <= n_compared <= got["n_atoms"]  # (n_atoms also counts the file that stops with an error)
    rmse = {"atom": rmse_of(file_pairs)}   # atom and protein depth: one total per file (tests/quality.rs:160-170)

    res = run_files("residue", lst)["results"]
    chain = run_files("chain", lst)["results"]
    prot = run_files("protein", lst)["results"]
    pairs_res, pairs_chain, pairs_prot = [], [], []
    for k, i in enumerate(qs["ids"]):
        if i in KNOWN_ERRORS:
            assert "error" in res[k] and "error" in chain[k] and "error" in prot[k]
            continue
        want = freesasa_chains(qs["ref"](i), False)
        sums = {}
        for cid, v in res[k]:                                   # tests/quality.rs:82-87
            sums[cid] = sums.get(cid, 0.0) + v
        pairs_res += [(sums[c], want[c]) for c in sorted(set(sums) & set(want))]
        by_name = dict((cid, v) for cid, v in chain[k])         # tests/quality.rs:88-92
        pairs_chain += [(by_name[c], want[c]) for c in sorted(set(by_name) & set(want))]
        pairs_prot.append((prot[k][0], freesasa_chains(qs["ref"](i), True)[i]))
    rmse.update(residue=rmse_of(pairs_res), chain=rmse_of(pairs_chain), protein=rmse_of(pairs_prot))
    assert len(pairs_res) >= 160 and len(pairs_chain) >= 160 and len(pairs_prot) == 87
    for level, v in rmse.items():
        assert v <= RMSE_GATE, (level, rmse)
    assert abs(rmse["residue"] - RMSE_BASELINE) < 0.5, rmse     # the reference's own RMSE at the level it is quoted for


@pytest.mark.gpu
def test_whole_set_with_radii_from_occupancy_meets_the_gate(quality_set, tmp_path):
    """tests/quality.rs:340-442: ProtOr radii written into the occupancy column, --read-radii-from-occupancy, residue level."""
    qs = quality_set
    paths, lst = write_list(tmp_path, qs, prepare=with_radii_in_occupancy)
    res = run_files("residue", lst, "--read-radii-from-occupancy")["results"]
    pairs = []
    for k, i in enumerate(qs["ids"]):
        if i in KNOWN_ERRORS:
            continue
        assert "error" not in res[k] if isinstance(res[k], dict) else True, i
        want = freesasa_chains(qs["ref"](i), False)
        sums = {}
        for cid, v in res[k]:
            sums[cid] = sums.get(cid, 0.0) + v
        pairs += [(sums[c], want[c]) for c in sorted(set(sums) & set(want))]
    rmse = rmse_of(pairs)
    assert len(pairs) >= 160 and rmse <= RMSE_GATE, rmse


def test_real_coords_workload_is_the_readers_selection_under_rigid_motions(quality_set):
    """bench.py's real_coords leg (real_coords.py): the batch is the 87 readable files' selections, whole and in the files'
    own frame; every tiled copy is a rigid motion of its structure (distances kept to text precision, radii and ids
    untouched), and no two copies share coordinates."""
    import real_coords as rc
    qs = quality_set
    base = rc.quality_set_batch()
    assert base.n_structures == 87 and "3sqz" not in base.names
    (x, y, z, r), ids, _ = select(qs["pdb"](base.names[5]))
    bx, by, bz, br, bids = base.structure(5)
    assert np.array_equal(bx, x) and np.array_equal(by, y) and np.array_equal(bz, z) and np.array_equal(br, r) and np.array_equal(bids, ids)
    t = rc.tiled(base, 2 * base.n_atoms, seed=3)
    assert t.n_structures == 2 * 87 and t.n_atoms == 2 * base.n_atoms and t.n_residues == 2 * base.n_residues
    so = t.structure_offsets.astype(np.int64)
    for s in (0, 33, 86):
        a = np.stack(t.structure(s)[:3], 1).astype(np.float64)
        m = np.stack(t.structure(87 + s)[:3], 1).astype(np.float64)
        assert np.array_equal(a, np.stack(base.structure(s)[:3], 1).astype(np.float64))
        i, j = np.arange(0, len(a) - 7, 5), np.arange(7, len(a), 5)[: len(np.arange(0, len(a) - 7, 5))]
        da, dm = np.linalg.norm(a[i] - a[j], axis=1), np.linalg.norm(m[i] - m[j], axis=1)
        assert np.max(np.abs(da - dm)) < 5e-3 and not np.array_equal(a, m)
        assert np.array_equal(t.radius[so[s]:so[s + 1]], t.radius[so[87 + s]:so[87 + s + 1]])
