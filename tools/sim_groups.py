#!/usr/bin/env python3
"""CPU simulation of k_occlusion_mx's grouping: how many group prologues (run look-ups + scan + union gathers) a batch
costs under different rules, and how many union chunks of 64 slots every atom then sweeps.

    python tools/sim_groups.py [n_structures] [--real]

Rules simulated on the bench batch's first n structures (cell-sorted, concatenated as on the device):
  natural      groups = (structure, z cell, y cell, x cell >> shift), no wave boundaries, no union limit
  shipped      a wave owns 64 consecutive atoms and lane 0 starts a group; unions over `cap` slots halve the group
  aligned(A)   a wave owns the groups that START in its window of A atoms and follows the last one for at most 64 - A
               atoms past it; A = 64 is the ideal "no cut" rule with an unbounded tail
  two rows     groups = (structure, z cell, y cell >> 1, x block): two segments of lanes, 30 runs
For each: atoms per prologue, union slots per group, swept chunks per atom, share of groups halved.
--real: the 88 structures of the reference's quality set (tests/golden/freesasa_set.tar.xz) instead.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_workloads as bw  # noqa: E402


def group_shift(n_atoms, n_cells):
    per = n_atoms / max(n_cells, 1)
    return 0 if per >= 1.0 else 1 if per >= 0.6 else 2 if per >= 0.3 else 3


class Struct:
    def __init__(self, x, y, z, r, probe=np.float32(1.4)):
        max_r = np.float32(max(0.0, float(r.max())))
        cs = np.float32(probe + max_r)
        inv = np.float32(1.0) / cs
        mn = np.array([x.min(), y.min(), z.min()], np.float32) - cs
        mx = np.array([x.max(), y.max(), z.max()], np.float32) + cs
        self.dims = np.ceil((mx - mn) * inv).astype(np.int64) + 1
        c = np.stack([((x - mn[0]) * inv).astype(np.int64), ((y - mn[1]) * inv).astype(np.int64),
                      ((z - mn[2]) * inv).astype(np.int64)], 1)
        d = self.dims
        lin = c[:, 0] + d[0] * (c[:, 1] + d[1] * c[:, 2])
        order = np.argsort(lin, kind="stable")
        self.c = c[order]
        self.n = len(x)
        self.n_cells = int(d[0] * d[1] * d[2])
        cnt = np.bincount(lin, minlength=self.n_cells)
        self.csum = np.concatenate([[0], np.cumsum(cnt)])
        self.shift = group_shift(self.n, self.n_cells)

    def run_len(self, cy, cz, x0, x1):
        d = self.dims
        if cy < 0 or cz < 0 or cy >= d[1] or cz >= d[2]:
            return 0
        x0, x1 = max(x0, 0), min(x1, d[0] - 1)
        if x1 < x0:
            return 0
        base = d[0] * (cy + d[1] * cz)
        return int(self.csum[base + x1 + 1] - self.csum[base + x0])

    def union(self, cy0, cy1, cz, cxf, cxl):
        """slots of the union around rows cy0..cy1 of z row cz, x cells cxf..cxl"""
        u = 0
        for dz in range(-2, 3):
            for yy in range(cy0 - 2, cy1 + 3):
                u += self.run_len(yy, cz + dz, cxf - 2, cxl + 2)
        return u


class Tally:
    def __init__(self, name):
        self.name, self.groups, self.atoms, self.slots, self.chunks, self.halved, self.lookups = name, 0, 0, 0, 0, 0, 0

    def add(self, n_atoms, u, cap, lookups=1, halved=0):
        self.groups += 1
        self.atoms += n_atoms
        self.slots += u
        self.chunks += ((min(u, cap) + 63) // 64) * n_atoms
        self.lookups += lookups
        self.halved += halved

    def line(self):
        return (f"{self.name:34s} atoms/prologue {self.atoms / max(self.groups, 1):5.2f}  prologues/atom {self.groups / max(self.atoms, 1):.4f}"
                f"  run look-ups/atom {self.lookups / max(self.atoms, 1):.4f}  union slots/group {self.slots / max(self.groups, 1):6.1f}"
                f"  chunks/atom {self.chunks / max(self.atoms, 1):.2f}  halved {100.0 * self.halved / max(self.groups, 1):.1f} %")


def one_row_group(st, c, tally, cap):
    """group of consecutive atoms c (same row, same x block): halve until the union fits (as the kernel does)"""
    stack = [c]
    while stack:
        g = stack.pop()
        lookups = 0
        while True:
            cxf, cxl = int(g[0, 0]), int(g[-1, 0])
            u = st.union(int(g[0, 1]), int(g[0, 1]), int(g[0, 2]), cxf, cxl)
            lookups += 1
            if u <= cap or cxf == cxl:
                break
            mid = (cxf + cxl) >> 1
            k = int(np.searchsorted(g[:, 0], mid, side="right"))
            stack.append(g[k:])
            g = g[:k]
        tally.add(len(g), u, cap, lookups, 1 if lookups > 1 else 0)


def batch_atoms(b):
    """The batch's cell-sorted atoms as the device lays them out: rows (sid, cx, cy, cz, shift), and the structures' grids."""
    structs = []
    for s in range(b.n_structures):
        x, y, z, r, _ = b.structure(s)
        structs.append(Struct(x, y, z, r))
    rows = []
    for sid, st in enumerate(structs):
        rows.append(np.concatenate([np.full((st.n, 1), sid), st.c, np.full((st.n, 1), st.shift)], 1))
    return structs, np.concatenate(rows)


def shipped_grouping(b, cap=320, wave=64):
    """What k_occlusion_mx's grouping does to batch `b` (the shipped rule: a wave owns 64 consecutive atoms, lane 0 starts a
    group, a union over `cap` slots halves its group): bench.py's real_coords leg prints this beside the kernel's time."""
    structs, A = batch_atoms(b)
    N = len(A)
    sid, cx, cy, cz, sh = A.T
    bx = cx >> sh
    st_mask = np.ones(N, bool)
    st_mask[1:] = (sid[1:] != sid[:-1]) | (cy[1:] != cy[:-1]) | (cz[1:] != cz[:-1]) | (bx[1:] != bx[:-1])
    natural = int(st_mask.sum())
    st_mask[::wave] = True
    t = Tally("shipped")
    idx = np.flatnonzero(st_mask)
    for g0, g1 in zip(idx, np.append(idx[1:], N)):
        one_row_group(structs[sid[g0]], A[g0:g1, 1:4], t, cap)
    return {"atoms": N, "structures": b.n_structures, "grid_cells_per_atom": round(sum(s.n_cells for s in structs) / N, 2),
            "atoms_per_natural_group": round(N / natural, 2), "atoms_per_prologue": round(t.atoms / t.groups, 2),
            "union_slots_per_group": round(t.slots / t.groups, 1), "swept_chunks_per_atom": round(t.chunks / t.atoms, 2),
            "groups_halved_for_union_overflow": round(t.halved / t.groups, 4), "union_slots": cap,
            "method": "CPU simulation of the kernel's grouping rule (tools/sim_groups.py; reproduces the 4.79 atoms per prologue "
                      "that the device stamps counted on the synthetic proteome with 256 slots: 4.71)"}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n_s = int(args[0]) if args else 60
    if "--real" in sys.argv:
        import real_coords
        b = real_coords.quality_set_batch()
    else:
        b = bw.synthetic_proteome(bw.PROTEOME_STRUCTURES)
        sizes = np.diff(b.structure_offsets.astype(np.int64))
        order = np.argsort(-sizes, kind="stable")
        # every k-th structure of the size-sorted list: the bench's size mix
        b = bw.select(b, order[:: max(1, len(order) // n_s)][:n_s])
    structs, A = batch_atoms(b)  # the batch's cell-sorted atoms: (sid, cx, cy, cz, shift)
    N = len(A)
    sid, cx, cy, cz, sh = A.T
    bx = cx >> sh
    nat = np.ones(N, bool)
    nat[1:] = (sid[1:] != sid[:-1]) | (cy[1:] != cy[:-1]) | (cz[1:] != cz[:-1]) | (bx[1:] != bx[:-1])
    print(f"structures {b.n_structures}  atoms {N}  cells/atom {sum(s.n_cells for s in structs) / N:.1f}  natural groups {nat.sum()} ({N / nat.sum():.2f} atoms each)")

    def run(name, starts_mask, cap):
        t = Tally(name)
        idx = np.flatnonzero(starts_mask)
        ends = np.append(idx[1:], N)
        for g0, g1 in zip(idx, ends):
            one_row_group(structs[sid[g0]], A[g0:g1, 1:4], t, cap)
        print(t.line())
        return t

    for cap in (256, 320):
        run(f"natural, cap {cap}", nat, cap)
        cut = nat.copy()
        cut[::64] = True
        run(f"shipped (wave = 64 atoms), cap {cap}", cut, cap)
    # aligned windows: wave k owns the groups starting in [kA, (k+1)A); its last group may run to kA + 63
    for Awin in (64, 60, 56, 48):
        tail = 64 - Awin if Awin < 64 else 10 ** 9
        st_mask = nat.copy()
        # a forced start where a group that began in an earlier window would pass that window's 64 lanes
        last_start = 0
        for p in range(N):
            if st_mask[p]:
                last_start = p
            else:
                k = last_start // Awin  # the window that owns the running group
                if p >= k * Awin + Awin + tail:
                    st_mask[p] = True
                    last_start = p
        run(f"aligned windows A = {Awin}, cap 320", st_mask, 320)
    # two-row groups: rows (2k, 2k + 1) of one z row, same x block; both segments must lie in one wave's 64 atoms,
    # else each row is a group of its own
    for cap in (320, 384):
        for wave in (64, 10 ** 9):
            t = Tally(f"two rows, wave {wave if wave < 10 ** 9 else 'unbounded'}, cap {cap}")
            idx = np.flatnonzero(nat)
            ends = np.append(idx[1:], N)
            key2 = {}
            for g0, g1 in zip(idx, ends):
                key2.setdefault((sid[g0], cz[g0], cy[g0] >> 1, bx[g0]), []).append((g0, g1))
            for k, segs in key2.items():
                st = structs[k[0]]
                if len(segs) == 2 and segs[0][0] // wave == (segs[1][1] - 1) // wave:
                    (a0, a1), (b0, b1) = segs
                    cxf = int(min(cx[a0], cx[b0])); cxl = int(max(cx[a1 - 1], cx[b1 - 1]))
                    u = st.union(int(cy[a0]), int(cy[b0]), int(cz[a0]), cxf, cxl)
                    if u <= cap:
                        t.add((a1 - a0) + (b1 - b0), u, cap)
                        continue
                for g0, g1 in segs:
                    # a segment that straddles a wave boundary is cut there
                    cuts = [g0] + [p for p in range(g0 + 1, g1) if p % wave == 0] + [g1]
                    for q0, q1 in zip(cuts[:-1], cuts[1:]):
                        one_row_group(st, A[q0:q1, 1:4], t, cap)
            print(t.line())


if __name__ == "__main__":
    main()
