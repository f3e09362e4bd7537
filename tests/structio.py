"""Minimal PDB / mmCIF atom-record reader used by the tests (test infrastructure).

It is deliberately independent of the product's C++ reader
(rustsasa_amd/csrc/host) so the two can be checked against each other.
Only the columns the SASA boundary needs are kept.
"""
from __future__ import annotations

import os
import shlex
from dataclasses import dataclass

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DATA_DIR = os.path.join(GOLDEN_DIR, "data")

# pdbtbx element van-der-Waals radii used by the reference's tests/units.rs:18-33
VDW = {"H": 1.20, "C": 1.77, "N": 1.66, "O": 1.50, "S": 1.89, "P": 1.90, "SE": 1.82, "CL": 1.82, "ZN": 2.39, "D": 1.20}


@dataclass
class AtomRec:
    hetero: bool
    serial: int
    name: str
    altloc: str
    resname: str
    chain: str
    resseq: int
    icode: str
    x: float
    y: float
    z: float
    occupancy: float
    element: str
    model: int


def read_pdb(path):
    atoms = []
    model = 1
    seen_model = False
    with open(path) as f:
        for line in f:
            rec = line[:6]
            if rec.startswith("MODEL"):
                if seen_model:
                    break  # first model only
                seen_model = True
                continue
            if rec.startswith("ENDMDL"):
                break
            if rec not in ("ATOM  ", "HETATM"):
                continue
            element = line[76:78].strip().upper() if len(line) >= 78 else ""
            name = line[12:16].strip()
            if not element:
                element = "".join(ch for ch in name if ch.isalpha())[:1].upper()
            atoms.append(AtomRec(
                hetero=(rec == "HETATM"), serial=int(line[6:11]), name=name,
                altloc=line[16].strip(), resname=line[17:20].strip(), chain=line[21].strip(),
                resseq=int(line[22:26]), icode=line[26].strip(),
                x=float(line[30:38]), y=float(line[38:46]), z=float(line[46:54]),
                occupancy=float(line[54:60]) if line[54:60].strip() else 1.0,
                element=element, model=model))
    return atoms


def read_mmcif(path):
    atoms = []
    cols = []
    in_loop = False
    with open(path) as f:
        for line in f:
            s = line.strip()
            if s == "loop_":
                in_loop, cols = True, []
                continue
            if in_loop and s.startswith("_atom_site."):
                cols.append(s.split(".", 1)[1])
                continue
            if in_loop and cols and (s.startswith("ATOM") or s.startswith("HETATM")):
                t = shlex.split(s) if ("'" in s or '"' in s) else s.split()
                g = dict(zip(cols, t))

                def dot(v):
                    return "" if v in (".", "?") else v
                model = int(g.get("pdbx_PDB_model_num", "1"))
                if atoms and model != atoms[0].model:
                    continue
                atoms.append(AtomRec(
                    hetero=(g["group_PDB"] == "HETATM"), serial=int(g["id"]),
                    name=g["label_atom_id"], altloc=dot(g.get("label_alt_id", ".")),
                    resname=g["label_comp_id"],
                    chain=g.get("auth_asym_id", g.get("label_asym_id")),
                    resseq=int(g.get("auth_seq_id", g.get("label_seq_id"))),
                    icode=dot(g.get("pdbx_PDB_ins_code", "?")),
                    x=float(g["Cartn_x"]), y=float(g["Cartn_y"]), z=float(g["Cartn_z"]),
                    occupancy=float(g.get("occupancy", "1.0")),
                    element=g["type_symbol"].upper(), model=model))
                continue
            if in_loop and cols and s.startswith("#"):
                in_loop, cols = False, []
    return atoms


def read_structure(path):
    return read_mmcif(path) if path.endswith(".cif") else read_pdb(path)


def conformers(residue_atoms):
    """The conformers of one residue as the reference's model holds them (pdbtbx): one per (residue name,
    alternate location) in order of first appearance; in a residue with alternate locations the atoms WITHOUT
    one belong to every conformer (appended after the conformer's own), and the blank conformer goes.
    Returns [(name, altloc, [atoms])]."""
    order, groups = [], {}
    for a in residue_atoms:
        key = (a.resname, a.altloc)
        if key not in groups:
            groups[key] = []
            order.append(key)
        groups[key].append(a)
    if len(order) > 1:
        blank = next((k for k in order if k[1] == ""), None)
        if blank is not None:
            order.remove(blank)
            for k in order:
                groups[k] = groups[k] + groups[blank]
    return [(k[0], k[1], groups[k]) for k in order]


def data_path(name):
    return os.path.join(DATA_DIR, name)


def soa_vdw(atoms):
    """All atoms, pdbtbx vdW radii, ids = serials (reference tests/units.rs:18-33)."""
    x = np.array([a.x for a in atoms], np.float64).astype(np.float32)
    y = np.array([a.y for a in atoms], np.float64).astype(np.float32)
    z = np.array([a.z for a in atoms], np.float64).astype(np.float32)
    r = np.array([VDW[a.element] for a in atoms], np.float64).astype(np.float32)
    ids = np.array([a.serial for a in atoms], np.uint64)
    return x, y, z, r, ids


def load_golden_low_res():
    path = os.path.join(GOLDEN_DIR, "fixed_low_res_atoms.txt")
    return np.loadtxt(path, dtype=np.float32, comments="#")


def parse_protor(path):
    """FreeSASA-format radii config -> {(residue, atom): radius} (consts.rs:31-81)."""
    types, table = {}, {}
    section = None
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or line.startswith("#") or line.startswith("name:"):
                continue
            if line == "types:":
                section = "types"
                continue
            if line == "atoms:":
                section = "atoms"
                continue
            parts = line.split()
            if section == "types" and len(parts) >= 2:
                try:
                    types[parts[0]] = float(np.float32(parts[1]))
                except ValueError:
                    pass
            elif section == "atoms" and len(parts) >= 3 and parts[2] in types:
                table[(parts[0], parts[1])] = types[parts[2]]
    return table
