#!/usr/bin/env python3
"""Two batches in flight inside ONE context (two workspaces, two streams) against one batch at a time, for the whole
proteome batch and for one rank's shard of an M-way split.  usage: tools/bench_two_in_flight.py [M ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, bench_workloads as bw
import rustsasa_amd

dev = torch.device("cuda:0")
for m in [int(a) for a in sys.argv[1:]] or [1, 8]:
    batch, n_points, _ = bench.make_workload("proteome", None, None, 0, 1, "strong", m if m > 1 else 0)
    with rustsasa_amd.Context(0) as ctx:
        run = bench.DeviceRun(ctx, batch, n_points, dev, True, None)
        for _ in range(5):
            run.step()
        # (the second workspace and its stream are created by the first overlapped enqueue: not part of the timing)
        run.enqueue(k=0)
        for i in range(1, 4):
            run.enqueue(k=i % 2)
            ctx.wait()
        ctx.wait()
        for timing in (False, True):
            ctx.enable_timing(timing)
            steps = 200 if m > 1 else 40
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(steps):
                run.step()
            torch.cuda.synchronize(); seq = (time.perf_counter() - t0) / steps
            torch.cuda.synchronize(); t0 = time.perf_counter()
            run.enqueue(k=0)
            for i in range(1, steps):
                run.enqueue(k=i % 2)
                ctx.wait()
            ctx.wait()
            torch.cuda.synchronize(); two = (time.perf_counter() - t0) / steps
            print(f"shard 1/{m}: {batch.n_structures} structures, {batch.n_atoms} atoms, timing events {timing}: "
                  f"one at a time {seq * 1e3:.4f} ms/step, two in flight {two * 1e3:.4f} ms/step")
