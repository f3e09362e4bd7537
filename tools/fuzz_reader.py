#!/usr/bin/env python3
"""Mutation fuzzing of the PDB / mmCIF reader and writer (no GPU needed).

    python tools/fuzz_reader.py [--asan] [-n 400]
    python tools/fuzz_reader.py --differential [--asan] [-n 2000] [--seed 7] [--jobs 8]

Feeds mutated copies of the fixture files to `sasa_host_cli parse|rewrite|prepare-fast` and reports any run that
dies from a signal or prints a sanitizer report.  --asan first builds the host layer with
`g++ -fsanitize=address,undefined` (CPU only; GPU sanitizers are not available on this pool).
tests/test_reader_fuzz.py runs a short round of the same mutations with the regular build.

--differential: directory mode's short cuts (fast_pdb_prepare / fast_cif_prepare: text -> kept atoms) against the general
reader + selection on the SAME mutant, `-n` mutants per format (PDB and mmCIF), seeded: `sasa_host_cli prepare-fast` and
`prepare-general` must print the same JSON (bit patterns of coordinates and radii, ids, segment ends, metadata, errors).
Mutation classes: structural (the tests' hand-written exits: alternate locations, chains / residues that come back, short
rows, missing elements, second models, ...), byte noise inside atom rows, and - mmCIF - the row splitter's state machine:
runs of rows whose separator mask equals the previous row's (offsets reused) broken by rows that must take the
character-wise tokenizer (tabs, quotes, bytes above 127, more than 256 bytes, shifted or widened columns), then aligned
rows again, so every transition aligned -> fallback -> aligned is crossed with the remembered mask in every state.
"""
import argparse
import os
import random
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "golden", "data")
FIXTURES = ["1jcd.pdb", "example.cif", "151L_H3.pdb", "bad_seqadv_1A06.pdb", "2drt.pdb"]
KINDS = ["flip", "trunc", "dropline", "dupline", "insert", "longline", "swapcols", "empty", "nulls"]


def mutate(data: bytes, rng: random.Random):
    b = bytearray(data)
    k = rng.choice(KINDS)
    if k == "flip":
        for _ in range(rng.randint(1, 40)):
            b[rng.randrange(len(b))] = rng.randrange(256)
    elif k == "trunc":
        b = b[:rng.randrange(len(b))]
    elif k in ("dropline", "dupline", "swapcols"):
        ls = bytes(b).split(b"\n")
        if k == "dropline":
            for _ in range(rng.randint(1, 20)):
                ls.pop(rng.randrange(len(ls)))
        elif k == "dupline":
            i = rng.randrange(len(ls))
            ls[i:i] = [ls[i]] * rng.randint(1, 5)
        else:
            for _ in range(30):
                i = rng.randrange(len(ls))
                l = bytearray(ls[i])
                if len(l) > 10:
                    a, c = rng.randrange(len(l)), rng.randrange(len(l))
                    l[a], l[c] = l[c], l[a]
                    ls[i] = bytes(l)
        b = bytearray(b"\n".join(ls))
    elif k == "insert":
        i = rng.randrange(len(b))
        b[i:i] = bytes(rng.randrange(32, 127) for _ in range(rng.randint(1, 200)))
    elif k == "longline":
        i = rng.randrange(len(b))
        b[i:i] = b"X" * rng.randint(1000, 100000)
    elif k == "empty":
        b = bytearray()
    elif k == "nulls":
        for _ in range(20):
            b[rng.randrange(len(b))] = 0
    return bytes(b), k


def run_cases(cli, n, seed=7, workdir=None):
    """Returns a list of (case, kind, mode, returncode, stderr tail) for runs that crashed."""
    rng = random.Random(seed)
    failures = []
    with tempfile.TemporaryDirectory(dir=workdir) as d:
        for it in range(n):
            name = rng.choice(FIXTURES)
            data, kind = mutate(open(os.path.join(DATA, name), "rb").read(), rng)
            p = os.path.join(d, f"c{it}{os.path.splitext(name)[1]}")
            with open(p, "wb") as f:
                f.write(data)
            # parse / rewrite: the general reader and the writer; prepare-fast: directory mode's short cuts
            # (fast_pdb_prepare / fast_cif_prepare, falling back to the general reader + selection)
            for mode in ("parse", "rewrite", "prepare-fast"):
                r = subprocess.run([cli, mode, p] + (["--level", str(it % 4), "--allow-vdw-fallback"] if mode == "prepare-fast" else []),
                                   capture_output=True, timeout=120)
                if r.returncode < 0 or b"Sanitizer" in r.stderr or b"runtime error" in r.stderr:
                    failures.append((it, kind, mode, r.returncode, r.stderr[-400:].decode(errors="replace")))
            os.remove(p)
    return failures


# ---- differential: short cuts against the general reader ------------------------------------------------------

def pdb_to_mmcif(text):
    """ATOM / HETATM records as an AlphaFold-style `_atom_site` loop (tests/test_host_api.py _pdb_to_mmcif)."""
    cols = ["group_PDB", "id", "type_symbol", "label_atom_id", "label_alt_id", "label_comp_id", "label_asym_id", "auth_asym_id",
            "label_entity_id", "label_seq_id", "auth_seq_id", "pdbx_PDB_ins_code", "Cartn_x", "Cartn_y", "Cartn_z", "occupancy",
            "B_iso_or_equiv", "pdbx_formal_charge", "pdbx_PDB_model_num"]
    rows, model = [], 1
    for l in text.split("\n"):
        if l.startswith("MODEL"):
            model = int(l[10:14] or 1)
        if not l.startswith(("ATOM  ", "HETATM")) or len(l) < 54:
            continue
        name = l[12:16].strip()
        if "'" in name:
            name = '"%s"' % name
        dot = lambda t: t.strip() or "."  # noqa: E731
        rows.append(" ".join([l[:6].strip(), dot(l[6:11]), dot(l[76:78]), name or ".", dot(l[16:17]), dot(l[17:20]), dot(l[21:22]),
                              dot(l[21:22]), "1", dot(l[22:26]), dot(l[22:26]), dot(l[26:27]), dot(l[30:38]), dot(l[38:46]),
                              dot(l[46:54]), dot(l[54:60]), dot(l[60:66]), "?", str(model)]))
    return "data_test\n#\nloop_\n" + "".join("_atom_site.%s\n" % c for c in cols) + "\n".join(rows) + "\n#\n"


def align_columns(text):
    """The loop's rows padded to common column positions, as writers do: rows then share one separator mask."""
    lines = text.split("\n")
    idx = [i for i, l in enumerate(lines) if l.startswith(("ATOM ", "HETATM "))]
    toks = [lines[i].split() for i in idx]
    n = max(len(t) for t in toks)
    w = [max((len(t[k]) for t in toks if len(t) > k), default=1) for k in range(n)]
    for i, t in zip(idx, toks):
        lines[i] = " ".join(tok.ljust(w[k]) for k, tok in enumerate(t)).rstrip()
    return "\n".join(lines)


def mutate_pdb_structural(lines, rng):
    atom_idx = [i for i, l in enumerate(lines) if l.startswith(("ATOM  ", "HETATM"))]
    mut = list(lines)
    for _ in range(rng.randint(1, 3)):
        i = rng.choice(atom_idx[5:-5])
        kind = rng.randrange(16)
        l = mut[i]
        if kind == 0:   mut[i] = l[:16] + rng.choice("AB") + l[17:]
        elif kind == 1: mut[i] = l[:21] + rng.choice("ZQ ") + l[22:]
        elif kind == 2: mut[i], mut[i + 1] = mut[i + 1], mut[i]
        elif kind == 3: mut[i] = l[:rng.randrange(6, 80)]
        elif kind == 4: mut[i] = l[:76] + "  " + l[78:]
        elif kind == 5: mut[i] = l[:12] + " XX " + l[16:]
        elif kind == 6: mut.insert(i, "ENDMDL"); mut.insert(i + 1, "MODEL        2")
        elif kind == 7: mut[i] = l[:22] + "%4d" % rng.randrange(-9, 40) + l[26:]
        elif kind == 8: mut[i] = l[:17] + rng.choice(["GLY", "ALA", "HOH", "  A"]) + l[20:]
        elif kind == 9: mut[i] = l[:26] + rng.choice("AB") + l[27:]
        elif kind == 10: mut[i] = l[:6] + rng.choice(["  abc", "*****", "A0000", "     "]) + l[11:]
        elif kind == 11: mut[i] = ("HETATM" if l.startswith("ATOM") else "ATOM  ") + l[6:]
        elif kind == 12: mut[i] = l[:30] + rng.choice(["     nan", "  1.5e01", "        ", " 1234567", "-999.999", "     inf"]) + l[38:]
        elif kind == 13: a, b = sorted(rng.sample(atom_idx[2:40], 2)); mut = mut[:i] + mut[a:b] + mut[i:]
        elif kind == 14: mut[i] = l.rstrip() + "\r"
        elif kind == 15: mut[i] = l[:54] + rng.choice(["  1.00", "  0.50", " -1.00", "   nan", "      "]) + l[60:]
    return mut


def mutate_cif_structural(lines, rng):
    atom_idx = [i for i, l in enumerate(lines) if l.startswith(("ATOM ", "HETATM "))]
    mut = list(lines)
    for _ in range(rng.randint(1, 3)):
        i = rng.choice(atom_idx[5:-5])
        f = mut[i].split()
        if len(f) < 19:
            continue
        kind = rng.randrange(14)

        def put(k, v):
            g = list(f)
            g[k] = v
            return " ".join(g)
        if kind == 0:   mut[i] = put(4, rng.choice("AB"))
        elif kind == 1: mut[i] = put(7, "Z")
        elif kind == 2: mut[i], mut[i + 1] = mut[i + 1], mut[i]
        elif kind == 3: mut[i] = " ".join(f[:rng.randrange(1, 19)])
        elif kind == 4: mut[i] = put(2, "?")
        elif kind == 5: mut[i] = put(3, "XX")
        elif kind == 6: mut[i:] = [" ".join(l.split()[:18] + ["2"]) if l.startswith(("ATOM ", "HETATM ")) else l for l in mut[i:]]
        elif kind == 7: mut[i] = put(10, str(rng.randrange(-5, 30)))
        elif kind == 8: mut[i] = put(5, "GLY" if f[5] != "GLY" else "ALA")
        elif kind == 9: mut[i] = put(11, "B")
        elif kind == 10: mut[i] = put(1, rng.choice(["abc", ".", "?", "-7"]))
        elif kind == 11: mut[i] = put(0, "HETATM" if f[0] == "ATOM" else "ATOM")
        elif kind == 12: mut[i] = put(rng.choice([12, 13, 14, 15]), rng.choice(["?", ".", "nan", "1e1", "--1", "1.5(3)"]))
        elif kind == 13: mut[i] = mut[i] + " extra"
    return mut


def mutate_cif_splitter(lines, rng):
    """The row splitter's state machine: aligned rows reuse the previous row's separator offsets; these rows break the
    run in every way the splitter distinguishes, singly and in short bursts, with aligned rows between them."""
    atom_idx = [i for i, l in enumerate(lines) if l.startswith(("ATOM ", "HETATM "))]
    mut = list(lines)
    n_breaks = rng.randint(1, 12)
    # three mutants in four use only rows the short cut keeps (it must split them like the general tokenizer does);
    # the fourth also has rows that make it give up (a byte above 127 in a name it looks up, a row cut short)
    kinds = list(range(12)) if rng.random() < 0.25 else [0, 1, 2, 3, 4, 5, 7, 8, 9, 10]
    for _ in range(n_breaks):
        i = rng.choice(atom_idx[2:-2])
        for j in range(i, min(i + rng.choice([1, 1, 1, 2, 3]), atom_idx[-1])):   # a burst of consecutive odd rows
            if not mut[j].startswith(("ATOM ", "HETATM ")):
                continue
            f = mut[j].split()
            if len(f) < 19:
                continue
            kind = rng.choice(kinds)
            if kind == 0:   mut[j] = mut[j].replace(" ", "\t", rng.randint(1, 6))           # tabs
            elif kind == 1: f[3] = '"%s"' % f[3]; mut[j] = " ".join(f)                        # a quoted name (unpadded row)
            elif kind == 2: mut[j] = mut[j].replace(f[3], "'%s'" % f[3], 1)                   # ... inside the aligned layout
            elif kind == 3: mut[j] = " " * rng.randint(1, 4) + mut[j]                         # the whole row shifted
            elif kind == 4: mut[j] = mut[j].replace(" ", "  ", rng.randint(1, 3))             # a separator widened
            elif kind == 5: f[16] = "1." + "0" * rng.randint(240, 400); mut[j] = " ".join(f)  # more than 256 bytes
            elif kind == 6: mut[j] = mut[j].replace(f[5], "\u00c5" + f[5][1:], 1)            # a byte above 127
            elif kind == 7: f[1] = str(10 ** rng.randint(6, 9) + j); mut[j] = " ".join(f)      # a wider token
            elif kind == 8: mut[j] = mut[j].rstrip() + " " * rng.randint(1, 30)               # trailing blanks
            elif kind == 9: mut[j] = mut[j].rstrip() + "\r"                                   # CR LF
            elif kind == 10: mut[j] = " ".join(f)                                              # single blanks (another mask)
            elif kind == 11: mut[j] = mut[j][:rng.randrange(8, max(9, len(mut[j])))]          # cut short
    return mut


def mutate_row_bytes(lines, rng, prefixes):
    idx = [i for i, l in enumerate(lines) if l.startswith(prefixes)]
    mut = list(lines)
    for _ in range(rng.randint(1, 6)):
        i = rng.choice(idx)
        l = bytearray(mut[i].encode("utf-8", "replace"))
        if not l:
            continue
        for _ in range(rng.randint(1, 3)):
            l[rng.randrange(len(l))] = rng.choice([32, 32, 46, 45, 48, 57, 65, 9, 39, 34, 63, rng.randrange(33, 127)])
        mut[i] = l.decode("utf-8", "replace")
    return mut


def differential(cli, n, seed, jobs, workdir=None):
    from concurrent.futures import ThreadPoolExecutor
    import json
    pdb_names = ["151L_H3.pdb", "bad_seqadv_1A06.pdb", "1jcd.pdb", "2drt.pdb", "freesasa/2gpi.pdb", "freesasa/4c1a.pdb"]
    pdb_texts = {k: open(os.path.join(DATA, k)).read() for k in pdb_names}
    # The short cuts take plain files (one conformer per residue, chains and residues that do not come back): crystal
    # structures with waters behind their chains leave them at once.  Half of the base texts are therefore "plain"
    # versions of the fixtures - ATOM records without alternate locations only - so that a mutant's short cut is
    # abandoned (or not) because of the MUTATION.
    for k in list(pdb_texts):
        plain = [l for l in pdb_texts[k].split("\n") if l.startswith("ATOM  ") and len(l) >= 78 and l[16] == " "]
        pdb_texts[k + " (plain)"] = "\n".join(plain + ["TER", "END", ""])
    cif_texts = {"example.cif": open(os.path.join(DATA, "example.cif")).read()}
    cif_texts["example.cif (again)"] = cif_texts["example.cif"]
    for k, t in pdb_texts.items():
        cif_texts[k + " as loop"] = pdb_to_mmcif(t)
        if k.endswith("(plain)"):
            cif_texts[k + " as aligned loop"] = align_columns(pdb_to_mmcif(t))
    option_sets = [(), ("--include-hetatms", "--allow-vdw-fallback"), ("--include-hydrogens", "--allow-vdw-fallback"),
                   ("--read-radii-from-occupancy",)]
    stats = {"pdb": [0, 0, 0], "cif": [0, 0, 0]}   # mutants, short cut taken, different
    by_class = {}                                  # (format, mutation class) -> [mutants, short cut taken]
    bad = []

    def one(task):
        fmt, it, name, klass, text, level, opts, path = task
        with open(path, "w", encoding="utf-8", errors="replace") as f:
            f.write(text)
        out = []
        for mode in ("prepare-fast", "prepare-general"):
            r = subprocess.run([cli, mode, path, "--level", str(level), *opts], capture_output=True, timeout=300)
            if r.returncode < 0 or b"Sanitizer" in r.stderr or b"runtime error" in r.stderr:
                return (task, "crash", mode, r.returncode, r.stderr[-300:].decode(errors="replace"))
            out.append(r.stdout)
        os.remove(path)
        try:
            a, b = json.loads(out[0]), json.loads(out[1])
        except Exception:  # noqa: BLE001
            # (a mutated name with a quote or a control character in it: the driver's JSON is not escaped - the two
            # routes' outputs are compared as bytes behind the "fast" field)
            ra, rb = (re.sub(rb'^\{"fast":(true|false),', b"{", o) for o in out)
            took = 1 if out[0].startswith(b'{"fast":true') else 0
            return (task, "ok", took) if ra == rb else (task, "differ-bytes", took, len(ra), len(rb))
        took = a.pop("fast", 0)
        b.pop("fast", 0)
        if a != b:
            return (task, "differ", took, len(a.get("atoms", [])), len(b.get("atoms", [])), a.get("error"), b.get("error"))
        return (task, "ok", took)

    with tempfile.TemporaryDirectory(dir=workdir) as d:
        tasks = []
        for fmt, texts in (("pdb", pdb_texts), ("cif", cif_texts)):
            rng = random.Random(seed * 1000003 + (1 if fmt == "cif" else 0))
            names = sorted(texts)
            for it in range(n):
                name = rng.choice(names)
                lines = texts[name].split("\n")
                if fmt == "pdb":
                    klass = rng.choice(["structural", "structural", "bytes"])
                    mut = mutate_pdb_structural(lines, rng) if klass == "structural" else mutate_row_bytes(lines, rng, ("ATOM  ", "HETATM"))
                else:
                    klass = rng.choice(["splitter", "splitter", "structural", "bytes", "splitter+structural"])
                    mut = lines
                    if "splitter" in klass: mut = mutate_cif_splitter(mut, rng)
                    if "structural" in klass: mut = mutate_cif_structural(mut, rng)
                    if klass == "bytes": mut = mutate_row_bytes(mut, rng, ("ATOM ", "HETATM "))
                tasks.append((fmt, it, name, klass, "\n".join(mut), rng.randrange(4), rng.choice(option_sets),
                              os.path.join(d, f"{fmt}{it}.{'pdb' if fmt == 'pdb' else 'cif'}")))
        with ThreadPoolExecutor(jobs) as ex:
            for res in ex.map(one, tasks):
                task, verdict = res[0], res[1]
                st = stats[task[0]]
                st[0] += 1
                bc = by_class.setdefault((task[0], task[3]), [0, 0])
                bc[0] += 1
                if verdict == "ok":
                    st[1] += res[2]
                    bc[1] += res[2]
                else:
                    st[2] += 1
                    bad.append((task[0], task[1], task[2], task[3], task[5], task[6]) + tuple(res[1:]))
    stats["by_class"] = by_class
    return stats, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asan", action="store_true")
    ap.add_argument("-n", type=int, default=None)
    ap.add_argument("--differential", action="store_true")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--jobs", type=int, default=max(1, min(8, os.cpu_count() or 1)))
    args = ap.parse_args()
    if args.n is None:
        args.n = 2000 if args.differential else 400
    cli = os.path.join(ROOT, "rustsasa_amd", "lib", "sasa_host_cli")
    if args.asan:
        cli = os.path.join(tempfile.gettempdir(), "sasa_host_cli_asan")
        lib = os.path.join(ROOT, "rustsasa_amd", "lib")
        src = os.path.join(ROOT, "rustsasa_amd", "csrc", "host")
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                               "-I", os.path.join(ROOT, "include"), os.path.join(src, "host_api.cpp"),
                               os.path.join(src, "sasa_host_cli.cpp"), "-o", cli, "-L", lib, "-lrustsasa_amd",
                               "-Wl,-rpath," + lib, "-lpthread"])
        for name in ("1jcd.pdb", "example.cif"):  # the model's copy / move rules under the sanitizers
            subprocess.check_call([cli, "model-selftest", os.path.join(DATA, name)], stdout=subprocess.DEVNULL)
    if args.differential:
        stats, bad = differential(cli, args.n, args.seed, args.jobs)
        for f in bad[:40]:
            print("FAIL", *f)
        by_class = stats.pop("by_class")
        for (fmt, klass), (n_mut, n_fast) in sorted(by_class.items()):
            print(f"  {fmt} / {klass}: {n_mut} mutants, short cut taken on {n_fast}")
        for fmt, (n_mut, n_fast, n_bad) in stats.items():
            print(f"{fmt}: {n_mut} mutants (seed {args.seed}), short cut taken on {n_fast}, general reader on {n_mut - n_fast - n_bad}, "
                  f"{n_bad} differences or crashes" + (" under ASan + UBSan" if args.asan else ""))
        sys.exit(1 if bad else 0)
    bad = run_cases(cli, args.n)
    for f in bad:
        print("FAIL", *f)
    print(f"{args.n} mutated files x 3 modes, {len(bad)} crashes")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
