#!/usr/bin/env python3
"""CPU simulation (numpy/scipy, no GPU): how many point tiles of phase A a patch-level test would remove.

The sphere points are reordered into spatially compact patches of 16 (recursive bisection); a patch with centre c and
chord radius eps is dead when ONE candidate j of the near tile(s) has  c.v_j + eps |v_j| < limit_j  (then every
point of the patch has s.v_j < limit_j).  Reported per workload: share of dead patches, share of patches that keep
at least one survivor of the per-point filter, survivors per atom.

    python tools/sim_patch_cull.py [uniform|proteome] [n_points] [sample]
"""
import sys
import os

import numpy as np
from scipy.spatial import cKDTree

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_workloads as bw  # noqa: E402


def lattice(n):
    i = np.arange(n, dtype=np.float32)
    t = i * np.float32(1.0 / n)
    inc = np.arccos(np.float32(1) - np.float32(2) * t)
    az = np.float32(10.166408) * i
    return np.stack([np.sin(inc) * np.cos(az), np.sin(inc) * np.sin(az), np.cos(inc)], 1).astype(np.float64)


def compact_order(pts, leaf=16):
    """indices of pts in an order whose consecutive runs of `leaf` are spatially compact (recursive bisection,
    left halves sized to multiples of leaf)"""
    def rec(idx):
        if len(idx) <= leaf:
            return [idx]
        p = pts[idx]
        ax = int(np.argmax(p.max(0) - p.min(0)))
        o = idx[np.argsort(p[:, ax], kind="stable")]
        n_leaf = (len(idx) + leaf - 1) // leaf
        left = (n_leaf // 2) * leaf
        return rec(o[:left]) + rec(o[left:])
    return np.concatenate(rec(np.arange(len(pts))))


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "uniform"
    n_points = int(sys.argv[2]) if len(sys.argv) > 2 else 960
    sample = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
    probe = 1.4
    rng = np.random.default_rng(1)
    if wl == "uniform":
        b = bw.synthetic_uniform(200000, seed=5)
        xyz = np.stack([b.x, b.y, b.z], 1).astype(np.float64)
        r = b.radius.astype(np.float64)
        # interior atoms only (the 1 M-atom cube is 99 % interior)
        lo, hi = xyz.min(0) + 8, xyz.max(0) - 8
        inner = np.where(np.all((xyz > lo) & (xyz < hi), 1))[0]
        pick = rng.choice(inner, sample, replace=False)
    else:
        b = bw.synthetic_proteome(12, seed=3)
        xyz = np.stack([b.x, b.y, b.z], 1).astype(np.float64)
        r = b.radius.astype(np.float64)
        pick = rng.choice(len(r), sample, replace=False)
    max_r = r.max()
    tree = cKDTree(xyz)
    pts = lattice(n_points)
    for leaf in (16, 32, 64):
        order = compact_order(pts, leaf)
        P = pts[order]
        n_patch = (n_points + leaf - 1) // leaf
        centres, eps = [], []
        for k in range(n_patch):
            q = P[k * leaf:(k + 1) * leaf]
            c = q.mean(0)
            c /= np.linalg.norm(c)
            centres.append(c)
            eps.append(np.linalg.norm(q - c, axis=1).max())
        centres, eps = np.array(centres), np.array(eps)
        stats = {nt: dict(dead=0, alive_after=0) for nt in (16, 32, 48)}
        tot_patches = 0
        surv16 = 0
        exposed_tot = 0
        patch_with_surv = 0
        for i in pick:
            R = r[i] + probe
            sr = r[i] + max_r + 2 * probe
            nb = [j for j in tree.query_ball_point(xyz[i], sr) if j != i]
            if not nb:
                continue
            nb = np.array(nb)
            rng.shuffle(nb)
            v = xyz[i] - xyz[nb]
            d2 = (v * v).sum(1)
            lim = ((r[nb] + probe) ** 2 - d2 - R * R) / (2 * R)
            near = d2 < 1.8 * R * R
            o = np.concatenate([np.where(near)[0], np.where(~near)[0]])
            v, lim, d2 = v[o], lim[o], d2[o]
            vn = np.sqrt(d2)
            tot_patches += n_patch
            dots_all = P @ v.T  # points x candidates
            occl_all = (dots_all < lim).any(1)
            exposed_tot += (~occl_all).sum()
            occl16 = (dots_all[:, :16] < lim[:16]).any(1)
            surv16 += (~occl16).sum()
            pw = np.add.reduceat((~occl16).astype(int), np.arange(0, n_points, leaf))
            patch_with_surv += (pw > 0).sum()
            for nt in stats:
                cd = centres @ v[:nt].T + eps[:, None] * vn[None, :nt]
                dead = (cd < lim[:nt]).any(1)
                stats[nt]["dead"] += dead.sum()
        print(f"{wl} N={n_points} patch={leaf}: patches/atom {n_patch}, eps max {eps.max():.3f} mean {eps.mean():.3f}; "
              f"survivors of tile 0 per atom {surv16 / len(pick):.1f}, exposed {exposed_tot / len(pick):.1f}, "
              f"patches with a tile-0 survivor {patch_with_surv / tot_patches:.3f}")
        for nt, s in stats.items():
            print(f"    one-candidate patch kill against the first {nt} candidates: dead share {s['dead'] / tot_patches:.3f}")


if __name__ == "__main__":
    main()
