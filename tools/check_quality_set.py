#!/usr/bin/env python3
"""The reference's quality gate on its WHOLE FreeSASA set, without a GPU (build container only).

    tools/check_quality_set.py [/root/reference]

tests/quality.rs:200-258 runs the reference over tests/data/freesasa_pdbs (88 files, 71 with alternate
locations, 12 with hydrogens) and requires the RMSE of the chain totals against FreeSASA's
(tests/data/freesasa_reference, Lee & Richards) to stay below 43.99 + 20; tests/quality.rs:340-442 does the same
with the ProtOr radii written into the occupancy column and --read-radii-from-occupancy.  Here the C++ reader
and atom selection (`sasa_host_cli select`: first conformer, hydrogen / HETATM filters, radii) feed the oracle
(oracle/sasa_oracle.c, the pinned restatement of the hot path), so the gate tests OUR reader semantics on every
file of the set - the part of the boundary pdbtbx does not pin.  Prints one JSON line; exit status 1 if a gate
fails.  Skips (exit 0) when the reference tree is absent (the GPU box)."""
import glob
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as po  # noqa: E402
import structio as sio  # noqa: E402

CLI = os.path.join(ROOT, "rustsasa_amd", "lib", "sasa_host_cli")
RMSE_GATE = 43.99 + 20.0  # tests/quality.rs:17-18,225


def chain_totals(path, *opts):
    p = subprocess.run([CLI, "select", path, *opts], capture_output=True, text=True)
    if p.returncode != 0:
        return None, (p.stdout + p.stderr).strip()[:200]
    d = json.loads(p.stdout)
    a = d["atoms"]
    if not a:
        return {}, None
    x, y, z, r = (np.array([v[k] for v in a], np.float32) for k in range(4))
    ids = np.array([int(v[4]) for v in a], np.uint64)
    atom = po.calculate_sasa_internal(x, y, z, r, ids, 1.4, 100, 8, threads=0)
    sums = po.residue_sums(atom, np.array([0] + d["chain_end"], np.uint32))
    return {c: float(s) for c, s in zip(d["chains"], sums)}, None  # (a repeated chain id: the last one wins, as in a HashMap)


def freesasa_chains(path):
    ref = json.load(open(path))
    return {c["label"]: c["area"]["total"] for r in ref["results"] for s in r["structure"] for c in s["chains"]}


def vdw_table():
    """The host API's van-der-Waals table (host_api.cpp vdw_radius), read from its source."""
    import re
    src = open(os.path.join(ROOT, "rustsasa_amd", "csrc", "host", "host_api.cpp")).read()
    body = src[src.index("bool vdw_radius("):]
    body = body[:body.index("return false;")]
    return {m.group(1): float(m.group(2)) for m in re.finditer(r'\{"([A-Z]+)",\s*([0-9.]+)f\}', body)}


def with_radii_in_occupancy(src, dst, protor, vdw):
    """tests/quality.rs:262-334: every atom's occupancy becomes its ProtOr radius (van der Waals radius of the
    element where the table has none)."""
    out = []
    for line in open(src):
        if line.startswith(("ATOM", "HETATM")) and len(line) >= 60:
            res, name = line[17:20].strip(), line[12:16].strip()
            r = protor.get((res, name))
            if r is None:
                el = line[76:78].strip().upper() if len(line) >= 78 else ""
                r = vdw.get(el)
            if r is None:
                if line.startswith("ATOM"):
                    raise ValueError(f"no radius for {res} {name} in {src}")
                r = 2.0  # (a HETATM of an element outside our table: not selected by default, any number will do)
            line = line[:54] + f"{r:6.2f}" + line[60:]
        out.append(line)
    open(dst, "w").write("".join(out))


def gate(pdbs, ref_dir, opts, prepare=None):
    ours, theirs, failed, files = [], [], [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for pdb in pdbs:
            pid = os.path.basename(pdb)[:-4]
            ref = os.path.join(ref_dir, pid + ".json")
            if not os.path.exists(ref):
                continue
            src = pdb
            if prepare:
                src = os.path.join(tmp, pid + ".pdb")
                prepare(pdb, src)
            got, err = chain_totals(src, *opts)
            if got is None:
                failed.append((pid, err))
                continue
            want = freesasa_chains(ref)
            common = sorted(set(want) & set(got))
            files += bool(common)
            ours += [got[k] for k in common]
            theirs += [want[k] for k in common]
    rmse = float(np.sqrt(np.mean((np.array(ours) - np.array(theirs)) ** 2)))
    return {"files_compared": files, "chains_compared": len(ours), "rmse": round(rmse, 3), "gate": RMSE_GATE,
            "passed": bool(rmse <= RMSE_GATE and ours), "files_with_errors": failed}


def main():
    ref_root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    pdb_dir = os.path.join(ref_root, "tests", "data", "freesasa_pdbs")
    ref_dir = os.path.join(ref_root, "tests", "data", "freesasa_reference")
    if not os.path.isdir(pdb_dir):
        print(json.dumps({"skipped": f"{pdb_dir} not present"}))
        return 0
    pdbs = sorted(glob.glob(os.path.join(pdb_dir, "*.pdb")))
    protor = sio.parse_protor(sio.data_path("protor.config"))
    vdw = vdw_table()
    res = {
        "files": len(pdbs),
        "default options (tests/quality.rs:200-258)": gate(pdbs, ref_dir, []),
        "radii from occupancy (tests/quality.rs:340-442)": gate(pdbs, ref_dir, ["--read-radii-from-occupancy"],
                                                                lambda s, d: with_radii_in_occupancy(s, d, protor, vdw)),
    }
    print(json.dumps(res))
    return 0 if all(v["passed"] for v in res.values() if isinstance(v, dict)) else 1


if __name__ == "__main__":
    sys.exit(main())
