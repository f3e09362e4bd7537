#!/usr/bin/env python3
"""Randomised parity soak on the GPU: batches of structures with random sizes, shapes (compact, elongated,
sparse, dense clusters, coincident atoms), radii, ids (some duplicated) and point counts, every atom compared
with the oracle.  usage: tools/soak_parity.py [seconds] [seed]"""
import os, sys, time
os.environ.setdefault("RSASA_TUNING", "1")  # (the library reads its RSASA_* measurement switches only then)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (one HIP runtime per process)
import rustsasa_amd
from oracle import pyoracle as po

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)


def structure(rng):
    kind = rng.integers(6)
    n = int(rng.choice([1, 2, 7, 60, 300, 1500, 4000, 9000, 20000, 70000], p=[.03, .03, .04, .1, .2, .3, .15, .1, .04, .01]))
    if kind == 0:    # compact blob at protein density
        side = (n / 0.05) ** (1 / 3)
        xyz = rng.uniform(0, side, (n, 3))
    elif kind == 1:  # elongated
        xyz = rng.uniform(0, 1, (n, 3)) * np.array([rng.uniform(200, 4000), rng.uniform(10, 60), rng.uniform(10, 60)])
    elif kind == 2:  # sparse: many windows of cells
        xyz = rng.uniform(0, 1, (n, 3)) * rng.uniform(300, 1500, 3)
    elif kind == 3:  # dense cluster(s): many atoms per cell
        centers = rng.uniform(0, 60, (max(1, n // 400), 3))
        xyz = centers[rng.integers(len(centers), size=n)] + rng.normal(scale=rng.uniform(0.3, 2.0), size=(n, 3))
    elif kind == 4:  # lattice with coincident points
        xyz = np.round(rng.uniform(0, 40, (n, 3)) / 2.0) * 2.0
    else:            # plane
        xyz = rng.uniform(0, 1, (n, 3)) * np.array([120.0, 120.0, 0.5])
    xyz += rng.uniform(-500, 500, 3)
    r = rng.uniform(1.0, 2.2, n) if rng.random() < 0.8 else np.full(n, rng.uniform(1.2, 2.0))
    if rng.random() < 0.08 and n > 3:  # radii the matrix-core kernel hands to the general one: negative, zero, huge, tiny
        k = rng.integers(n, size=max(1, n // 200))
        r[k] = rng.choice([-1.0, 0.0, 70.0, 63.9, 64.1, 0.05], size=len(k))
    return xyz.astype(np.float32), r.astype(np.float32)


t_end = time.time() + budget
it = atoms = 0
with rustsasa_amd.Context(0) as ctx:
    while time.time() < t_end:
        parts = [structure(rng) for _ in range(int(rng.integers(1, 40)))]
        if rng.random() < 0.2:
            parts.insert(int(rng.integers(len(parts) + 1)), (np.zeros((0, 3), np.float32), np.zeros(0, np.float32)))
        xyz = np.concatenate([p[0] for p in parts]); r = np.concatenate([p[1] for p in parts])
        so = np.concatenate([[0], np.cumsum([len(p[0]) for p in parts])]).astype(np.uint32)
        ids = np.arange(len(xyz), dtype=np.uint64)
        if rng.random() < 0.3 and len(ids) > 4:  # some duplicated ids
            k = rng.integers(len(ids), size=max(1, len(ids) // 50)); ids[k] = ids[(k + 1) % len(ids)]
        if rng.random() < 0.4:  # ids in no order (hashes: an odd multiplier keeps equal ids equal and different ones different)
            ids = ids * np.uint64(0x9E3779B97F4A7C15)
        elif rng.random() < 0.3:  # serials that start over in every structure
            ids = np.concatenate([np.arange(1, len(p[0]) + 1, dtype=np.uint64) for p in parts]) if len(xyz) else ids
        use_ids = ids if rng.random() < 0.8 else None
        n_points = int(rng.choice([1, 20, 64, 100, 100, 100, 103, 110, 128, 131, 200, 960, 1000, 1003]))
        probe = float(rng.choice([1.4, 1.4, 1.4, 0.0, 0.7, 2.5]))
        if len(xyz) < 3000 and rng.random() < 0.1:
            probe = 33.0  # (a probe the matrix-core kernel does not take; the oracle needs minutes for it on large inputs)
        if rng.random() < 0.06 and len(xyz) > 10:  # NaN coordinates (either sign) and radii: the reference's arithmetic, bit for bit
            k = rng.integers(len(xyz), size=int(rng.integers(1, 6)))
            xyz[k, rng.integers(3, size=len(k))] = np.array([0x7FC00000, 0xFFC00000], np.uint32).view(np.float32)[rng.integers(2, size=len(k))]
            if rng.random() < 0.5:
                r[rng.integers(len(r))] = np.nan
        x, y, z = (np.ascontiguousarray(xyz[:, k]) for k in range(3))
        # random residues: consecutive runs of atoms that never cross a structure boundary
        cuts = set(so.tolist())
        if len(xyz):
            cuts.update(rng.integers(0, len(xyz), size=max(1, len(xyz) // 9)).tolist())
        ro = np.array(sorted(cuts), np.uint32)
        mode = rng.integers(4)
        if mode == 0:    # host arrays in, host arrays out
            got, got_res = ctx.calculate_sasa_batch(x, y, z, r, use_ids, so, probe, n_points, residue_offsets=ro)
            got_k = None
        elif mode == 3:  # the same as a stream of host batches (two queued: this one twice)
            got, got_res = ctx.host_batch_enqueue(x, y, z, r, use_ids, so, probe, n_points, residue_offsets=ro)
            got2, _ = ctx.host_batch_enqueue(x, y, z, r, use_ids, so, probe, n_points, residue_offsets=ro)
            ctx.host_batch_wait(); ctx.host_batch_wait()
            if not np.array_equal(got, got2, equal_nan=True):
                print(f"STREAM MISMATCH between the two queued copies, iteration {it}"); sys.exit(1)
            got_k = None
        else:            # device-resident, with the per-atom candidate counts
            dev = torch.device("cuda:0")
            tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
            out = torch.full((len(xyz),), -1.0, dtype=torch.float32, device=dev)
            res = torch.full((len(ro) - 1,), -1.0, dtype=torch.float32, device=dev)
            kk = torch.zeros(len(xyz), dtype=torch.int32, device=dev)
            ctx.enqueue_device(tt(x), tt(y), tt(z), tt(r), None if use_ids is None else tt(use_ids.view(np.int64)), so, out,
                               tt(ro.view(np.int32)), res, kk, probe, n_points, stream=torch.cuda.current_stream().cuda_stream)
            ctx.wait()
            got, got_res, got_k = out.cpu().numpy(), res.cpu().numpy(), kk.cpu().numpy().view(np.uint32)
        want = po.calculate_sasa_batch(x, y, z, r, use_ids, so, probe, n_points, 8, threads=0)
        want_res = np.array([np.float32(0) if a == b else np.add.reduce(want[a:b], dtype=np.float32) for a, b in zip(ro[:-1], ro[1:])],
                            np.float32) if len(ro) > 1 else np.zeros(0, np.float32)
        # sequential f32 sums (options.rs:209-216): numpy's pairwise reduce differs for long runs, so compare run by run
        for k_, (a, b) in enumerate(zip(ro[:-1], ro[1:])):
            acc = np.float32(0)
            for v in want[a:b]:
                acc = np.float32(acc + v)
            want_res[k_] = acc
        if got_k is not None and len(xyz) <= 30000:
            for s0, s1 in zip(so[:-1], so[1:]):
                if s1 > s0:
                    _, _, wk = po.calculate_sasa_internal(x[s0:s1], y[s0:s1], z[s0:s1], r[s0:s1], None if use_ids is None else use_ids[s0:s1],
                                                          probe, n_points, 8, return_details=True)
                    if not np.array_equal(got_k[s0:s1], wk):
                        print(f"K MISMATCH iteration {it}"); sys.exit(1)
        if not np.array_equal(got_res, want_res, equal_nan=True):
            print(f"RESIDUE MISMATCH iteration {it}: {int((got_res != want_res).sum())} of {len(want_res)}"); sys.exit(1)
        if not np.array_equal(got, want, equal_nan=True):
            bad = np.flatnonzero(~((got == want) | (np.isnan(got) & np.isnan(want))))
            np.savez("soak_failure.npz", xyz=xyz, r=r, so=so, ids=ids, n_points=n_points, probe=probe)
            print(f"MISMATCH iteration {it}: {len(bad)} of {len(got)} atoms differ (first {bad[:5]}), saved soak_failure.npz")
            sys.exit(1)
        it += 1; atoms += len(got)
print(f"soak ok: {it} batches, {atoms} atoms, all equal to the oracle")
