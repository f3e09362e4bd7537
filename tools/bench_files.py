#!/usr/bin/env python3
"""End-to-end directory-mode measurement (files -> per-residue SASA), C++ host API.

Writes N synthetic AF-proteome-like PDB files (residue-aligned fragments of the fixture
proteins, rigidly moved; same size distribution as bench_workloads.synthetic_proteome) to a
scratch directory, then runs `sasa_host_cli files residue` (SASAOptions::process_files):
multi-threaded parse + atom selection on the host, one GPU batch per chunk of files.
Reports files/s with the parse / compute split.  This is the parse- and PCIe-inclusive
number; bench.py's `value` is the HBM-resident hot path only.
"""
import argparse
import json
import os
os.environ.setdefault("RSASA_TUNING", "1")  # (the library reads its RSASA_* measurement switches only then)
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench_workloads as bw  # noqa: E402
import structio as sio  # noqa: E402

CLI = os.path.join(ROOT, "rustsasa_amd", "lib", "sasa_host_cli")


def load_domains():
    doms = []
    for name in bw.FIXTURES:
        recs = [a for a in sio.read_structure(sio.data_path(name)) if not a.hetero and a.element != "H"]
        xyz = np.array([[a.x, a.y, a.z] for a in recs])
        keys = [(a.chain, a.resseq, a.icode) for a in recs]
        starts = [0] + [i for i in range(1, len(keys)) if keys[i] != keys[i - 1]] + [len(keys)]
        doms.append((recs, xyz - xyz.mean(axis=0), np.array(starts)))
    return doms


CIF_HEAD = "data_synthetic\n#\nloop_\n" + "".join("_atom_site.%s\n" % c for c in (
    "group_PDB", "id", "type_symbol", "label_atom_id", "label_alt_id", "label_comp_id", "label_asym_id", "auth_asym_id",
    "label_entity_id", "label_seq_id", "auth_seq_id", "pdbx_PDB_ins_code", "Cartn_x", "Cartn_y", "Cartn_z", "occupancy",
    "B_iso_or_equiv", "pdbx_formal_charge", "pdbx_PDB_model_num"))


def write_structure(path, n_target, rng, doms, cif=False):
    """cif: the same atoms as an AlphaFold-style mmCIF file (the `_atom_site` loop of tests/golden/data/example.cif)."""
    lines, n, serial, resno, slot = [], 0, 1, 1, 0
    while n < n_target:
        recs, xyz, starts = doms[rng.integers(len(doms))]
        want = n_target - n
        n_res = len(starts) - 1
        if want >= starts[-1]:
            r0 = 0
        else:
            last = int(np.searchsorted(starts, starts[-1] - want, side="right") - 1)
            r0 = int(rng.integers(max(last, 0) + 1))
        r1 = min(max(int(np.searchsorted(starts, starts[r0] + want, side="right") - 1), r0 + 1), n_res)
        a0, a1 = starts[r0], starts[r1]
        frag = xyz[a0:a1] - xyz[a0:a1].mean(axis=0)
        frag = frag @ bw._random_rotation(rng).T + rng.normal(scale=0.05, size=frag.shape)
        frag = frag + np.array([slot % 4, (slot // 4) % 4, slot // 16]) * 90.0
        slot += 1
        chain = "ABCDEFGHIJKLMNOPQRSTUVWXYZ"[slot % 26]
        prev_key = None
        for k in range(a0, a1):
            a = recs[k]
            key = (a.chain, a.resseq, a.icode)
            if key != prev_key and prev_key is not None:
                resno += 1
            prev_key = key
            x, y, z = frag[k - a0]
            name = a.name if len(a.name) == 4 else " " + a.name
            if cif:
                lines.append("ATOM %-5d %s %-4s . %s %s %s 1 %-4d %-4d . %-8.3f %-8.3f %-8.3f 1.0 %-9.5f 0 1" %
                             (serial, a.element, a.name, a.resname, chain, chain, resno, resno, x, y, z, 50.0 + (resno % 40)))
            else:
                lines.append("ATOM  %5d %-4s %3s %1s%4d    %8.3f%8.3f%8.3f  1.00  0.00          %2s  " %
                             (serial % 100000, name, a.resname, chain, resno % 10000, x, y, z, a.element))
            serial += 1
        resno += 1
        n += a1 - a0
    with open(path, "w") as f:
        f.write(CIF_HEAD + "\n".join(lines) + "\n#\n" if cif else "\n".join(lines) + "\nEND\n")
    return n


def _write_one(job):
    d, i, n_t, fmt = job
    global _DOMS
    try:
        doms = _DOMS
    except NameError:
        doms = _DOMS = load_domains()
    p = os.path.join(d, f"s{i:05d}.{fmt}")
    return p, write_structure(p, int(n_t), np.random.default_rng(bw.PROTEOME_SEED + 7919 * i), doms, cif=fmt == "cif")


def write_set(d, n_files, fmt, procs=None):
    """n synthetic files of the proteome's size mix in directory d, every file from a seed of its own (so that a pool of
    processes writes them: 6 ms of Python per file); returns (paths, atoms).  This process must not have touched a GPU."""
    import multiprocessing as mp
    rng = np.random.default_rng(bw.PROTEOME_SEED)
    sizes = np.clip(rng.lognormal(np.log(2000.0), 0.75, n_files), 150, 25000).astype(int)
    jobs = [(d, i, int(n), fmt) for i, n in enumerate(sizes)]
    procs = procs or min(16, len(os.sched_getaffinity(0)))
    with mp.get_context("fork").Pool(procs) as pool:
        res = pool.map(_write_one, jobs, chunksize=32)
    return [p for p, _ in res], int(sum(a for _, a in res))


def end_to_end(args):
    """The reference's published benchmark end to end (paper/eval/benchmark.sh:1: hyperfine over the whole proteome
    directory; src/main.rs:203-226,342-480): N files in, one JSON file per input out (serde's shape), as ONE
    process_files call - the first call of a fresh process (HIP start-up inside, like the reference's process start) and
    a later call of the same process.  Input and output on /dev/shm."""
    import shutil
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="rsasa_e2e_", dir=base)
    try:
        t0 = time.time()
        paths, atoms = write_set(d, args.files, args.format)
        gen_s = time.time() - t0
        lst = os.path.join(d, "files.txt")
        open(lst, "w").write("\n".join(paths) + "\n")
        out_dir = os.path.join(d, "out")
        os.mkdir(out_dir)
        best = None
        for _ in range(args.repeat):
            p = subprocess.run([CLI, "files", "residue", lst, "--threads", str(args.threads), "--batch", str(args.batch), "--workers",
                                str(args.workers), "--devices", str(args.devices), "--calls", "2", "--out-dir", out_dir],
                               capture_output=True, text=True)
            assert p.returncode == 0, p.stderr[-500:]
            r = json.loads(p.stdout)
            r.pop("results")
            if best is None or r["calls_s"][0] < best["calls_s"][0]:
                best = r
        outs = sorted(os.listdir(out_dir))
        assert len(outs) == best["n_ok"] == args.files, (len(outs), best["n_ok"])
        sample = outs[:: max(1, len(outs) // 25)]
        n_res = 0
        for o in sample:  # a sample of the written files parses and has serde's shape
            rows = json.load(open(os.path.join(out_dir, o)))["Residue"]
            assert list(rows[0].keys()) == ["serial_number", "insertion_code", "value", "name", "is_polar", "chain_id"]
            n_res += len(rows)
        print(json.dumps({
            "files": args.files, "format": args.format, "atoms": atoms, "level": "ResidueLevel, one <stem>.json per input (serde's shape)",
            "seconds": round(best["calls_s"][0], 4), "files_per_s": round(args.files / best["calls_s"][0], 1),
            "seconds_later_call": round(best["calls_s"][1], 4), "files_per_s_later_call": round(args.files / best["calls_s"][1], 1),
            "bytes_read": sum(os.path.getsize(p) for p in paths), "bytes_written": best["bytes_written"],
            "outputs_checked": len(sample), "generation_s": round(gen_s, 1),
            "reference_published": {"seconds": 5.237, "files_per_s": 833.1, "hardware": "Apple M3, 8 cores (BASELINE.md: the reference's "
                                    "own paper figure for the E. coli proteome, 4 363 files; context, not a same-node comparison)"},
            "note": "seconds: a fresh process's first process_files call, HIP runtime start-up inside; /dev/shm in and out"}))
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--end-to-end", action="store_true", help="files in -> one JSON file per input out, one call (see end_to_end)")
    ap.add_argument("--files", type=int, default=1500)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--dir", default=None)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--workers", type=int, default=0, help="GPU worker contexts (0 = the default context)")
    ap.add_argument("--devices", type=int, default=1)
    ap.add_argument("--calls", type=int, default=3, help="process_files calls per process (the first includes HIP start-up)")
    ap.add_argument("--format", choices=("pdb", "cif"), default="pdb", help="the files' format (cif: AlphaFold-style mmCIF)")
    args = ap.parse_args()
    if args.end_to_end:
        return end_to_end(args)
    d = args.dir or tempfile.mkdtemp(prefix="rsasa_files_")
    rng = np.random.default_rng(bw.PROTEOME_SEED)
    sizes = np.clip(rng.lognormal(np.log(2000.0), 0.75, args.files), 150, 25000).astype(int)
    doms = load_domains()
    t0 = time.time()
    paths, atoms = [], 0
    for i, n_t in enumerate(sizes):
        p = os.path.join(d, f"s{i:05d}.{args.format}")
        atoms += write_structure(p, int(n_t), rng, doms, cif=args.format == "cif")
        paths.append(p)
    lst = os.path.join(d, "files.txt")
    open(lst, "w").write("\n".join(paths) + "\n")
    gen_s = time.time() - t0
    best = None
    for _ in range(args.repeat):
        cmd = [CLI, "files", "residue", lst, "--threads", str(args.threads), "--batch", str(args.batch),
               "--workers", str(args.workers), "--devices", str(args.devices), "--calls", str(args.calls)]
        p = subprocess.run(cmd, capture_output=True, text=True)
        assert p.returncode == 0, p.stderr[-500:]
        if os.environ.get("RSASA_FILES_TRACE"):
            sys.stderr.write(p.stderr[-3000:])
        r = json.loads(p.stdout)
        r.pop("results")
        r["first_call_s"] = r["calls_s"][0]
        r["later_calls_s"] = min(r["calls_s"][1:]) if len(r["calls_s"]) > 1 else None
        r["total_s"] = r["first_call_s"]
        if best is None or r["total_s"] < best["total_s"]:
            best = r
    best.update({"files_per_s": round(best["n_files"] / best["total_s"], 1),
                 "files_per_s_later_calls": round(best["n_files"] / best["later_calls_s"], 1) if best["later_calls_s"] else None,
                 "note": "files_per_s: a fresh process's first call (HIP runtime start-up inside); later calls of the same process: files_per_s_later_calls",
                 "atoms_per_file": round(atoms / args.files, 1), "generation_s": round(gen_s, 1),
                 "host_threads": args.threads or os.cpu_count(), "format": args.format,
                 "bytes_on_disk": sum(os.path.getsize(p) for p in paths)})
    print(json.dumps(best))


if __name__ == "__main__":
    main()
