#!/usr/bin/env python3
"""CPU simulation: how many union slots k_occlusion_mx would sweep with its 25 x-runs trimmed by row gaps.

A group's union is the 25 rows (dy, dz) of the 5x5 block around its cell row, cells [cx_first - 2, cx_last + 2].
A row whose y/z gap to the group's atoms leaves nothing of the search radius is dropped, the others shortened to the
x cells within reach.  Variants: x bounds from the cells (cheap) or from the atoms' positions.

    python tools/sim_union_trim.py [n_structures]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_workloads as bw  # noqa: E402


def main():
    n_s = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    b = bw.synthetic_proteome(n_s, seed=3)
    probe = np.float32(1.4)
    tot = dict(base=0, cellx=0, exact=0, groups=0, atoms=0)
    chunks = dict(base=0, cellx=0, exact=0)
    for s in range(b.n_structures):
        x, y, z, r, _ = b.structure(s)
        max_r = np.float32(max(0.0, r.max()))
        cs = probe + max_r
        inv = np.float32(1.0) / cs
        mn = np.array([x.min(), y.min(), z.min()], np.float32) - cs
        c = np.stack([((x - mn[0]) * inv).astype(np.int64), ((y - mn[1]) * inv).astype(np.int64), ((z - mn[2]) * inv).astype(np.int64)], 1)
        dims = c.max(0) + 3
        lin = c[:, 0] + dims[0] * (c[:, 1] + dims[1] * c[:, 2])
        order = np.argsort(lin, kind="stable")
        c, pos, rr = c[order], np.stack([x, y, z], 1)[order], r[order]
        # occupancy count per cell
        cnt = np.zeros(dims[0] * dims[1] * dims[2] + 1, np.int64)
        np.add.at(cnt, lin, 1)
        csum = np.concatenate([[0], np.cumsum(cnt)])

        def run_len(cy, cz, x0, x1):
            if cy < 0 or cz < 0 or cy >= dims[1] or cz >= dims[2] or x1 < x0:
                return 0
            x0, x1 = max(x0, 0), min(x1, dims[0] - 1)
            base = dims[0] * (cy + dims[1] * cz)
            return int(csum[base + x1 + 1] - csum[base + x0])

        key = (c[:, 0] >> 3) + 4096 * (c[:, 1] + 4096 * c[:, 2])
        starts = np.concatenate([[0], np.where(key[1:] != key[:-1])[0] + 1, [len(key)]])
        for g0, g1 in zip(starts[:-1], starts[1:]):
            gc = c[g0]
            cxf, cxl = c[g0:g1, 0].min(), c[g0:g1, 0].max()
            p = pos[g0:g1]
            sr = float(rr[g0:g1].max() + max_r + 2 * probe)
            U = dict(base=0, cellx=0, exact=0)
            for dz in range(-2, 3):
                for dy in range(-2, 3):
                    cy, cz = gc[1] + dy, gc[2] + dz
                    U["base"] += run_len(cy, cz, cxf - 2, cxl + 2)
                    # gaps between the group's atoms and the row's slab (in angstrom)
                    ylo, yhi = mn[1] + cy * cs, mn[1] + (cy + 1) * cs
                    zlo, zhi = mn[2] + cz * cs, mn[2] + (cz + 1) * cs
                    gy = max(0.0, ylo - p[:, 1].max(), p[:, 1].min() - yhi)
                    gz = max(0.0, zlo - p[:, 2].max(), p[:, 2].min() - zhi)
                    rem2 = sr * sr - gy * gy - gz * gz
                    if rem2 < 0:
                        continue
                    reach = np.sqrt(rem2) + 1e-3
                    # x bounds from the cells
                    nc = int(reach * inv) + 1
                    U["cellx"] += run_len(cy, cz, max(cxf - 2, cxf - nc), min(cxl + 2, cxl + nc))
                    x0 = int((p[:, 0].min() - reach - mn[0]) * inv)
                    x1 = int((p[:, 0].max() + reach - mn[0]) * inv)
                    U["exact"] += run_len(cy, cz, max(cxf - 2, x0), min(cxl + 2, x1))
            for k in U:
                tot[k] += U[k]
                chunks[k] += (min(U[k], 256) + 63) // 64 * (g1 - g0)
            tot["groups"] += 1
            tot["atoms"] += g1 - g0
    print(f"structures {b.n_structures} atoms {tot['atoms']} groups {tot['groups']} ({tot['atoms'] / tot['groups']:.2f} atoms each)")
    for k in ("base", "cellx", "exact"):
        print(f"  {k:6s}: union slots per group {tot[k] / tot['groups']:.1f}, swept chunks per atom {chunks[k] / tot['atoms']:.2f}")


if __name__ == "__main__":
    main()
