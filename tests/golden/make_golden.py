#!/usr/bin/env python3
"""Regenerates the golden fixtures under tests/golden/ from the reference tree.

Run in the build container only (needs /root/reference, which does not exist
on the GPU box).  Fixtures are DATA: the reference's own test inputs
(tests/data/...) and the numeric literals of its golden vector
FIXED_LOW_RES_ATOMS (tests/common/data.rs:4-238), written one value per line.
No reference source text is stored.
"""
import io
import os
import re
import shutil
import sys
import tarfile

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

DATA_FILES = [
    "tests/data/pdbs/example.cif",
    "tests/data/pdbs/151L_H3.pdb",
    "tests/data/pdbs/bad_seqadv_1A06.pdb",
    "tests/data/freesasa_pdbs/1jcd.pdb",
    "tests/data/freesasa_pdbs/2drt.pdb",   # small clean multi-chain file with HETATM
    "radii/protor.config",                 # ProtOr radii table (FreeSASA data file)
]
# Small files WITH alternate locations from the reference's quality set (tests/quality.rs runs the
# whole tests/data/freesasa_pdbs directory, 46 MB; these are its six smallest alt-loc files), each
# with its FreeSASA chain totals (tests/data/freesasa_reference/<id>.json).
ALTLOC_IDS = ["2gpi", "3w7y", "3uc7", "3kyz", "4oxx", "3zsj"]
# A small file WITH hydrogens from the same set (the include_hydrogens option, tests/test_host_api.py).
HYDROGEN_IDS = ["4c1a"]


def main():
    os.makedirs(os.path.join(HERE, "data"), exist_ok=True)
    for rel in DATA_FILES:
        dst = os.path.join(HERE, "data", os.path.basename(rel))
        shutil.copyfile(os.path.join(REF, rel), dst)
        os.chmod(dst, 0o644)

    os.makedirs(os.path.join(HERE, "data", "freesasa"), exist_ok=True)
    for pid in ALTLOC_IDS + HYDROGEN_IDS:
        for rel in (f"tests/data/freesasa_pdbs/{pid}.pdb", f"tests/data/freesasa_reference/{pid}.json"):
            dst = os.path.join(HERE, "data", "freesasa", os.path.basename(rel))
            shutil.copyfile(os.path.join(REF, rel), dst)
            os.chmod(dst, 0o644)

    # The WHOLE quality set (tests/quality.rs:200-258: 88 PDB files, 46 MB, and FreeSASA's chain totals for each) as one
    # xz archive: the GPU test sends all of it through process_files in one batch.  Sorted names, fixed metadata: the
    # archive only changes when the data does.
    n_set = 0
    with tarfile.open(os.path.join(HERE, "freesasa_set.tar.xz"), "w:xz", preset=9) as tar:
        for sub in ("freesasa_pdbs", "freesasa_reference"):
            for name in sorted(os.listdir(os.path.join(REF, "tests/data", sub))):
                data = open(os.path.join(REF, "tests/data", sub, name), "rb").read()
                info = tarfile.TarInfo(f"{sub}/{name}")
                info.size, info.mtime, info.mode = len(data), 0, 0o644
                tar.addfile(info, io.BytesIO(data))
                n_set += 1
    print(f"wrote freesasa_set.tar.xz with {n_set} files")

    text = open(os.path.join(REF, "tests/common/data.rs")).read()
    m = re.search(r"FIXED_LOW_RES_ATOMS:\s*\[f32;\s*(\d+)\]\s*=\s*\[(.*?)\];", text, re.S)
    n = int(m.group(1))
    vals = [v.strip() for v in m.group(2).replace("\n", " ").split(",") if v.strip()]
    assert len(vals) == n == 2622, (len(vals), n)
    with open(os.path.join(HERE, "fixed_low_res_atoms.txt"), "w") as f:
        f.write("# FIXED_LOW_RES_ATOMS (reference tests/common/data.rs:4-238): per-atom SASA,\n"
                "# example.cif, pdbtbx van-der-Waals radii, ids = serials, probe 1.4, 100 points\n")
        f.write("\n".join(vals) + "\n")
    print(f"wrote {n} golden values and {len(DATA_FILES)} data files")


if __name__ == "__main__":
    main()
