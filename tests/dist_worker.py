"""Worker of tests/test_distributed_cpu.py: one of WORLD_SIZE gloo ranks on the CPU.

Exercises the host logic of the multi-GPU path without a GPU: the per-rank
workloads of bench.py (weak scaling) and its largest-first equal-atom sharding of
one proteome (strong scaling, the default) are independent units with no data-path
collective; only the final MAX/SUM aggregation uses torch.distributed.  The
oracle stands in for the GPU engine as the per-shard compute.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench  # noqa: E402
import bench_workloads as bw  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")

    # weak scaling: every rank builds its own batch (seed + rank), as bench.py does
    batch, n_points, _ = bench.make_workload("proteome", 6, None, rank)
    assert n_points == 100
    el, structures, atoms = bench.aggregate(dist, dev, 1.0 + rank, batch.n_structures, batch.n_atoms)
    counts = [None] * world
    dist.all_gather_object(counts, (batch.n_structures, batch.n_atoms, float(batch.x[:16].sum())))
    assert el == float(world)                          # MAX over ranks of 1.0 + rank
    assert structures == sum(c[0] for c in counts) == 6 * world
    assert atoms == sum(c[1] for c in counts)
    assert len({c[2] for c in counts}) == world         # ranks really got different structures

    # strong scaling (bench.py's default): ONE proteome, largest structures first, equal-atom
    # shards from the same sharder bench.py uses; shards are disjoint, complete, balanced, and
    # computing them independently reproduces the unsharded result exactly
    n_total = 10
    mine, _, _ = bench.make_workload("proteome", n_total, None, rank, world, "strong")
    full = bw.synthetic_proteome(n_total, seed=bw.PROTEOME_SEED)
    sasa = po.calculate_sasa_batch(mine.x, mine.y, mine.z, mine.radius, mine.ids,
                                   mine.structure_offsets, 1.4, 100, 8, threads=1)
    res = po.residue_sums(sasa, mine.residue_offsets)
    parts = [None] * world
    dist.all_gather_object(parts, (rank, sasa, res, [int(i) for i in mine.shard_indices]))
    if rank == 0:
        ref = po.calculate_sasa_batch(full.x, full.y, full.z, full.radius, full.ids,
                                      full.structure_offsets, 1.4, 100, 8, threads=2)
        ref_res = po.residue_sums(ref, full.residue_offsets)
        assert sorted(i for p in parts for i in p[3]) == list(range(n_total))
        so = full.structure_offsets.astype(np.int64)
        ro = full.residue_offsets.astype(np.int64)
        sizes = np.diff(so)
        loads = [int(sizes[p[3]].sum()) for p in parts]
        assert max(loads) - min(loads) <= int(sizes.max())      # largest-first greedy bound
        for r, s_sasa, s_res, idx in parts:
            assert all(sizes[idx[k]] >= sizes[idx[k + 1]] for k in range(len(idx) - 1))
            pos = rpos = 0
            for s in idx:
                n = so[s + 1] - so[s]
                assert np.array_equal(s_sasa[pos:pos + n], ref[so[s]:so[s + 1]])
                r0, r1 = np.searchsorted(ro, so[s]), np.searchsorted(ro, so[s + 1])
                assert np.array_equal(s_res[rpos:rpos + (r1 - r0)], ref_res[r0:r1])
                pos += n
                rpos += r1 - r0
            assert pos == len(s_sasa) and rpos == len(s_res)
        print("DIST_OK", world)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
