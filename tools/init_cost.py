#!/usr/bin/env python3
"""One-time costs on a fresh process: context creation (HIP runtime start-up), first tiny call, first and second
256-structure call through host buffers."""
import sys, time, numpy as np
sys.path.insert(0, '.')
t0=time.perf_counter()
import torch
import rustsasa_amd, bench_workloads as bw
t1=time.perf_counter()
b = bw.synthetic_proteome(256, seed=3)
small = bw.synthetic_proteome(1, seed=4)
t2=time.perf_counter()
ctx = rustsasa_amd.Context(0)
t3=time.perf_counter()
ctx.calculate_sasa_batch(small.x, small.y, small.z, small.radius, small.ids, small.structure_offsets, 1.4, 100)
t4=time.perf_counter()
ctx.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, 1.4, 100, residue_offsets=b.residue_offsets)
t5=time.perf_counter()
ctx.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, 1.4, 100, residue_offsets=b.residue_offsets)
t6=time.perf_counter()
print(f"import {t1-t0:.3f}s  context create {1e3*(t3-t2):.1f} ms  first tiny call {1e3*(t4-t3):.1f} ms  first 256-structure call ({b.n_atoms} atoms) {1e3*(t5-t4):.1f} ms  second {1e3*(t6-t5):.1f} ms")
