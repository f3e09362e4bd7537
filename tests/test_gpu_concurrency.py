"""Re-entrancy of the C ABI on the GPU: the reference's hot path is called concurrently from
rayon workers in directory mode (src/main.rs:375), so the drop-in must give the same answers
when driven from several host threads -- one context per thread, or one shared context."""
import threading

import numpy as np
import pytest

import bench_workloads as bw
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

PROBE = 1.4


def _cases():
    out = []
    for name in ["1jcd.pdb", "151L_H3.pdb", "bad_seqadv_1A06.pdb", "example.cif"]:
        xyz, r, _, ids = bw.fixture_soa(name)
        x, y, z = (np.ascontiguousarray(xyz[:, k]).astype(np.float32) for k in range(3))
        out.append((x, y, z, r, ids, po.calculate_sasa_internal(x, y, z, r, ids, PROBE, 100, 8)))
    return out


def _hammer(contexts, cases, rounds):
    errors = []

    def work(tid):
        try:
            c = contexts[tid % len(contexts)]
            for it in range(rounds):
                x, y, z, r, ids, want = cases[(tid + it) % len(cases)]
                got = c.calculate_sasa_soa(x, y, z, r, ids, PROBE, 100)
                if not np.array_equal(got, want):
                    errors.append((tid, it, int(np.sum(got != want))))
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    return errors


def test_one_context_per_thread():
    import rustsasa_amd
    cases = _cases()
    contexts = [rustsasa_amd.Context(0) for _ in range(8)]
    try:
        assert _hammer(contexts, cases, 25) == []
    finally:
        for c in contexts:
            c.close()


def test_threads_sharing_one_context():
    import rustsasa_amd
    cases = _cases()
    with rustsasa_amd.Context(0) as c:
        assert _hammer([c], cases, 25) == []


def test_context_churn_leaves_results_stable():
    """Create / use / destroy many contexts: workspace and lattice caches are per context."""
    import rustsasa_amd
    x, y, z, r, ids, want = _cases()[0]
    for n_points in [100, 37, 100, 960, 100]:
        with rustsasa_amd.Context(0) as c:
            got = c.calculate_sasa_soa(x, y, z, r, ids, PROBE, n_points)
            if n_points == 100:
                assert np.array_equal(got, want)
            else:
                assert np.array_equal(got, po.calculate_sasa_internal(x, y, z, r, ids, PROBE, n_points, 8))
