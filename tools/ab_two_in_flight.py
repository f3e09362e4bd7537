#!/usr/bin/env python3
"""A tool of this directory for every library under rustsasa_amd/lib/variants/ (and the shipped one), a process each:
    python tools/ab_two_in_flight.py [--tool bench_h2h.py] [the tool's arguments]     (default: bench_two_in_flight.py 1)"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = ("import sys, runpy; sys.path.insert(0, %r); import rustsasa_amd._capi as c; c.LIB_PATH = %r; "
         "sys.argv = ['tool'] + %r; runpy.run_path(%r, run_name='__main__')")
args = sys.argv[1:]
tool = "bench_two_in_flight.py"
if args[:1] == ["--tool"]:
    tool, args = args[1], args[2:]
libs = {"base": os.path.join(ROOT, "rustsasa_amd", "lib", "librustsasa_amd.so")}
for p in sorted(glob.glob(os.path.join(ROOT, "rustsasa_amd", "lib", "variants", "*", "librustsasa_amd.so"))):
    libs[os.path.basename(os.path.dirname(p))] = p
for name, path in libs.items():
    p = subprocess.run([sys.executable, "-c", CHILD % (ROOT, path, args or ["1"], os.path.join(ROOT, "tools", tool))],
                       capture_output=True, text=True, cwd=ROOT)
    for line in p.stdout.splitlines():
        if line.startswith(("shard", "sub-batches")):
            print(f"{name:12s} {line}")
    if p.returncode:
        print(name, "FAILED", p.stderr[-300:])
