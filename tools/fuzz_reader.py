#!/usr/bin/env python3
"""Mutation fuzzing of the PDB / mmCIF reader and writer (no GPU needed).

    python tools/fuzz_reader.py [--asan] [-n 400]

Feeds mutated copies of the fixture files to `sasa_host_cli parse|rewrite|prepare-fast` and reports any run that
dies from a signal or prints a sanitizer report.  --asan first builds the host layer with
`g++ -fsanitize=address,undefined` (CPU only; GPU sanitizers are not available on this pool).
tests/test_reader_fuzz.py runs a short round of the same mutations with the regular build.
"""
import argparse
import os
import random
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asan", action="store_true")
    ap.add_argument("-n", type=int, default=400)
    args = ap.parse_args()
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
    bad = run_cases(cli, args.n)
    for f in bad:
        print("FAIL", *f)
    print(f"{args.n} mutated files x 3 modes, {len(bad)} crashes")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
