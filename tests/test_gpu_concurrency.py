"""Re-entrancy of the C ABI on the GPU: the reference's hot path is called concurrently from
rayon workers in directory mode (src/main.rs:375), so the drop-in must give the same answers
when driven from several host threads -- one context per thread, or one shared context."""
import os
import threading

import numpy as np
import pytest

import bench_workloads as bw
from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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


def test_two_batches_in_flight_in_one_context():
    """rsasa_batch_enqueue keeps up to two batches in flight per context (two workspaces, two streams; a third
    enqueue first waits for the oldest); rsasa_batch_wait returns them oldest first.  Different batches, different
    point counts and batches small enough for the per-atom kernels, interleaved: every result equals the
    one-batch-at-a-time run, and a wait with nothing in flight returns at once."""
    import torch
    import rustsasa_amd
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    batches = [bw.synthetic_proteome(60, seed=3), bw.synthetic_proteome(25, seed=4), bw.synthetic_proteome(3, seed=5),
               bw.synthetic_proteome(40, seed=6)]
    points = [100, 96, 100, 1000]
    with rustsasa_amd.Context(0) as ctx:
        ctx.wait()  # nothing in flight
        dev_in, want = [], []
        for b, n_points in zip(batches, points):
            x, y, z, r, ids = t(b.x), t(b.y), t(b.z), t(b.radius), t(b.ids.view(np.int64))
            ro = t(b.residue_offsets.view(np.int32))
            dev_in.append((x, y, z, r, ids, ro))
            out = torch.empty(b.n_atoms, dtype=torch.float32, device=dev)
            res = torch.empty(b.n_residues, dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            ctx.enqueue_device(x, y, z, r, ids, b.structure_offsets, out, ro, res, None, 1.4, n_points)
            ctx.wait()
            want.append((out.cpu().numpy(), res.cpu().numpy()))
        # all four enqueued back to back (the third and fourth enqueue wait for the first and second themselves),
        # twice over, results collected oldest first
        for _ in range(2):
            outs = []
            for (x, y, z, r, ids, ro), b, n_points in zip(dev_in, batches, points):
                out = torch.full((b.n_atoms,), -1.0, dtype=torch.float32, device=dev)
                res = torch.full((b.n_residues,), -1.0, dtype=torch.float32, device=dev)
                outs.append((out, res))
                torch.cuda.synchronize()
                ctx.enqueue_device(x, y, z, r, ids, b.structure_offsets, out, ro, res, None, 1.4, n_points)
            ctx.wait_all()
            ctx.wait()
            for (out, res), (wa, wr) in zip(outs, want):
                assert np.array_equal(out.cpu().numpy(), wa) and np.array_equal(res.cpu().numpy(), wr)
        # enqueue k + 1, then wait for k
        x, y, z, r, ids, ro = dev_in[0]
        b = batches[0]
        o = [(torch.empty(b.n_atoms, dtype=torch.float32, device=dev), torch.empty(b.n_residues, dtype=torch.float32, device=dev))
             for _ in range(2)]
        ctx.enqueue_device(x, y, z, r, ids, b.structure_offsets, o[0][0], ro, o[0][1], None, 1.4, 100)
        for i in range(1, 7):
            ctx.enqueue_device(x, y, z, r, ids, b.structure_offsets, o[i % 2][0], ro, o[i % 2][1], None, 1.4, 100)
            ctx.wait()
            assert np.array_equal(o[(i - 1) % 2][0].cpu().numpy(), want[0][0])
            o[(i - 1) % 2][0].fill_(-1.0)
        ctx.wait()
        assert np.array_equal(o[0][0].cpu().numpy(), want[0][0]) and np.array_equal(o[0][1].cpu().numpy(), want[0][1])
    # the oracle on one of the batches, so that "equal" is not equally wrong
    b = batches[2]
    assert np.array_equal(want[2][0], po.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, 1.4, 100, 8, threads=4))


@pytest.mark.gpu
def test_two_ranks_share_one_gpu():
    """The N > 1 path with the REAL engine on every rank (reference src/main.rs:375,439: files dealt to workers that
    run the hot path by themselves): `bench.py --gpus 2` starts its two ranks itself, the strong-scaling sharder gives
    each its structures, each rank runs the timed stepping on its shard and compares it with the oracle.  RCCL cannot
    put two ranks on one device, and the test boxes have one: the process group is gloo (barrier, MAX / SUM on the host)
    and both ranks name device 0.  No scaling number is read off this."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo",
                        "--device", "0", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--h2h-steps", "2",
                        "--two-steps", "2", "--config5-steps", "0", "--files", "0", "--per-call-seconds", "0", "--weak-steps", "2",
                        "--structures", "96", "--verify-shards"], capture_output=True, text=True, env=env,
                       timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["steps"] == 3
    assert out["config"]["dist_backend"] == "gloo" and out["config"]["structures_total"] == 96
    assert out["shards_disjoint_and_complete"] is True
    sp = out["shard_parity"]
    assert [s["rank"] for s in sp] == [0, 1] and all(s["device"] == 0 for s in sp)
    assert sum(s["structures"] for s in sp) == 96 and all(s["atoms"] >= 32768 for s in sp)  # (the matrix-core kernel's batches)
    assert all(s["atoms_equal_oracle"] and s["residues_equal_oracle"] for s in sp), sp
    assert out["config"]["outputs_of_both_workspaces_equal"] is True
    # the secondary legs run at N > 1 too (host to host in both modes, one batch at a time, weak scaling)
    assert out["host_to_host"]["residues_equal_hbm_run"] is True and out["weak_scaling"]["structures_per_gpu"] == 96
    assert out["one_at_a_time"]["steps"] == 2


def test_stream_of_host_batches_matches_the_oracle(monkeypatch):
    """rsasa_host_batch_enqueue / _wait (ABI 3): host batches enqueued ahead of the waits - two computing at a time,
    their uploads taking turns on the link - give the oracle's values, in order, atoms and residues; a batch with bad
    arguments reports through ITS wait and leaves the stream usable; pageable and pinned outputs both work."""
    import torch
    import rustsasa_amd
    monkeypatch.setenv("RSASA_SUB_ATOMS", "100000")  # sub-batches (the pipelined path) from 200 k atoms on, not 3 M
    batches = [bw.synthetic_proteome(n, seed=s) for n, s in ((120, 11), (90, 12), (150, 13), (70, 14), (110, 15))]
    assert all(b.n_atoms >= 200000 for b in batches[:3])
    want = [po.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, PROBE, 100, 8, threads=0)
            for b in batches]

    def pin(a):
        return torch.from_numpy(np.ascontiguousarray(a)).pin_memory().numpy()

    with rustsasa_amd.Context(0) as ctx:
        assert ctx._lib.rsasa_host_batch_wait(ctx._h) == 0  # nothing enqueued: returns at once
        outs = []

        def enq(k, pinned):
            b = batches[k]
            cols = [b.x, b.y, b.z, b.radius, b.ids, b.residue_offsets]
            if pinned:
                cols = [pin(c) for c in cols]
            x, y, z, r, ids, ro = cols
            outs.append(ctx.host_batch_enqueue(x, y, z, r, ids, b.structure_offsets, PROBE, 100, residue_offsets=ro,
                                               atom_out=pin(np.zeros(b.n_atoms, np.float32)) if pinned else None,
                                               res_out=pin(np.zeros(b.n_residues, np.float32)) if pinned else None))

        enq(0, True); enq(1, False); enq(2, True)      # three queued: two compute, one waits for a worker
        ctx.host_batch_wait()                          # batch 0
        enq(3, False); enq(4, True)
        # a batch with decreasing residue offsets: rejected when a worker takes it, reported by the wait that returns it
        b = batches[3]
        bad_ro = b.residue_offsets.copy()
        bad_ro[3], bad_ro[4] = bad_ro[4] + 5, bad_ro[3]
        ctx.host_batch_enqueue(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, PROBE, 100, residue_offsets=bad_ro)
        for _ in range(4):
            ctx.host_batch_wait()                      # batches 1 .. 4
        with pytest.raises(rustsasa_amd.RsasaError):
            ctx.host_batch_wait()                      # the bad one
        enq(1, True)                                   # the stream goes on
        ctx.host_batch_wait_all()
        for k, (atom, res) in zip([0, 1, 2, 3, 4, 1], outs):
            assert np.array_equal(atom, want[k]), k
            assert np.array_equal(res, po.residue_sums(want[k], batches[k].residue_offsets)), k
        # the context itself still serves synchronous calls with the same settings
        b = batches[3]
        atom, _ = ctx.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, PROBE, 100)
        assert np.array_equal(atom, want[3])


def test_stream_of_host_batches_runs_with_the_callers_lane_count():
    """The stream's worker contexts take the caller's settings at every enqueue (rsasa_context_clone_settings): with
    pulp's lane count 16, 100 points have four remainder points, and the values must be that lane count's."""
    import rustsasa_amd
    b = bw.synthetic_proteome(30, seed=21)
    with rustsasa_amd.Context(0) as ctx:
        for w in (16, 8):
            ctx.set_simd_width(w)
            atom, _ = ctx.host_batch_enqueue(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, PROBE, 100)
            ctx.host_batch_wait()
            want = po.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, PROBE, 100, w, threads=0)
            assert np.array_equal(atom, want), w


def test_measurement_switches_are_read_only_under_rsasa_tuning():
    """The library's RSASA_* switches (kernel choice, sub-batch sizes, traces) are measurement aids: a production
    process that happens to have one in its environment computes as if it had not.  Two child processes run the same
    host batch with RSASA_SUB_ATOMS=100000 and RSASA_H2H_TRACE=1 set: only the one that also says RSASA_TUNING=1 cuts
    the batch into sub-batches and traces them."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import numpy as np; import bench_workloads as bw; import rustsasa_amd\n"
            "b = bw.synthetic_proteome(100, seed=3)\n"
            "with rustsasa_amd.Context(0) as c:\n"
            "    a, _ = c.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, 1.4, 100)\n"
            "print('TOTAL %%.3f' %% float(a.sum(dtype=np.float64)))\n") % ROOT
    totals = {}
    for tuning in ("0", "1"):
        env = {k: v for k, v in os.environ.items() if not k.startswith("RSASA_")}
        env.update(RSASA_SUB_ATOMS="100000", RSASA_H2H_TRACE="1", RSASA_TUNING=tuning)
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        totals[tuning] = [ln for ln in p.stdout.splitlines() if ln.startswith("TOTAL")][0]
        assert ("h2h ctx" in p.stderr) == (tuning == "1"), p.stderr[-500:]
    assert totals["0"] == totals["1"]


def test_call_combining_merges_concurrent_calls_and_keeps_every_value():
    """rsasa_context_set_call_combining (ABI 4): 32 host threads make the reference's own call - one structure per call
    (src/main.rs:375,439, src/lib.rs:249-254) - with mixed structure sizes, lane counts W, point counts and entry points
    (AoS records, columns with ids, columns without), through one shared context and through contexts of their own.
    Calls with different settings must never be merged, so every value must equal the oracle run with THAT call's
    settings; a call with unusable input fails alone, a call with NaN input (defined: the oracle's values) computes
    alone, and neither disturbs the calls merged around it; the counters say that merging did happen."""
    import rustsasa_amd
    from rustsasa_amd import RsasaError
    proteome = bw.synthetic_proteome(48, seed=11)
    sizes = np.diff(proteome.structure_offsets.astype(np.int64))
    pick = np.argsort(sizes)[[0, 5, 12, 20, 28, 36, 44, 47]]  # 150 .. ~10 000 atoms
    structs = [tuple(np.ascontiguousarray(a) for a in proteome.structure(int(s))) for s in pick]
    x0, y0, z0, r0, i0 = structs[2]
    nan_case = (np.where(np.arange(len(x0)) == 7, np.float32(np.nan), x0), y0, z0, r0, i0)
    inf_x = x0.copy()
    inf_x[3] = np.inf
    settings = [(8, 100), (16, 100), (4, 97), (8, 200)]  # (W, points): 100 % 16 = 4 and 97 % 4 = 1 remainder points
    want = {}
    for k, (w, n) in enumerate(settings):
        for s, (x, y, z, r, ids) in enumerate(structs):
            want[(k, s, True)] = po.calculate_sasa_internal(x, y, z, r, ids, PROBE, n, w)
            if k == 0:
                want[(k, s, False)] = po.calculate_sasa_internal(x, y, z, r, None, PROBE, n, w)
    want_nan = po.calculate_sasa_internal(*nan_case, PROBE, 100, 8)
    shared = rustsasa_amd.Context(0)
    shared.set_call_combining(0)
    own = {}
    errors, done = [], []
    b0, c0 = rustsasa_amd.Context.call_combining_stats(0)

    def work(tid):
        try:
            k = tid % 4
            w, n = settings[k]
            if k == 0:
                c = shared  # eight threads on one context
            else:
                c = own[tid] = rustsasa_amd.Context(0)
                c.set_simd_width(w)
                c.set_call_combining(20 if tid % 8 == k else 0)
            for it in range(24):
                s = (tid * 3 + it) % len(structs)
                x, y, z, r, ids = structs[s]
                if tid == 4 and it % 6 == 0:  # unusable input: this call's error, nobody else's
                    try:
                        c.calculate_sasa_soa(inf_x, y0, z0, r0, i0, PROBE, n)
                        errors.append((tid, it, "an infinite coordinate was accepted"))
                    except RsasaError as e:
                        if e.status != -5:
                            errors.append((tid, it, f"status {e.status}"))
                    continue
                if tid == 8 and it % 5 == 0:  # NaN input: defined, computed by the call alone
                    got = c.calculate_sasa_soa(*nan_case, PROBE, 100)
                    if not np.array_equal(got, want_nan, equal_nan=True):
                        errors.append((tid, it, "NaN case"))
                    continue
                with_ids = not (k == 0 and tid % 8 == 4)
                if tid % 2 == 0 and with_ids:
                    got = c.calculate_sasa_internal(rustsasa_amd.make_atoms(x, y, z, r, ids), PROBE, n)
                else:
                    got = c.calculate_sasa_soa(x, y, z, r, ids if with_ids else None, PROBE, n)
                if not np.array_equal(got, want[(k, s, with_ids)]):
                    errors.append((tid, it, s, int(np.sum(got != want[(k, s, with_ids)]))))
            done.append(tid)
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(32)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a combined call hangs"
    b1, c1 = rustsasa_amd.Context.call_combining_stats(0)
    for c in own.values():
        c.close()
    shared.close()
    assert errors == [] and len(done) == 32
    assert c1 - c0 > 0 and b1 - b0 < c1 - c0, f"{c1 - c0} calls in {b1 - b0} batches: nothing was merged"
    # switched off again, a context's calls run by themselves
    with rustsasa_amd.Context(0) as c:
        c.set_call_combining(0)
        c.set_call_combining(-1)
        x, y, z, r, ids = structs[1]
        assert np.array_equal(c.calculate_sasa_soa(x, y, z, r, ids, PROBE, 100), want[(0, 1, True)])
        assert rustsasa_amd.Context.call_combining_stats(0) == (b1, c1)


def test_ninth_host_batch_is_refused_at_once():
    """Eight host batches may be queued and not yet waited for; a ninth rsasa_host_batch_enqueue returns
    RSASA_ERR_QUEUE_FULL without blocking and without touching the queue (ABI 4; it used to wait for the oldest batch
    and then fail), and after one wait there is room again."""
    import time
    import rustsasa_amd
    from rustsasa_amd import RsasaError
    b = bw.synthetic_proteome(6, seed=9)
    want = po.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, PROBE, 100, 8, threads=0)
    with rustsasa_amd.Context(0) as ctx:
        outs = [ctx.host_batch_enqueue(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, PROBE, 100)[0] for _ in range(8)]
        time.sleep(0.5)  # (all eight are computed by now: the queue stays full until a wait takes one out)
        t0 = time.perf_counter()
        with pytest.raises(RsasaError) as e:
            ctx.host_batch_enqueue(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, PROBE, 100)
        assert e.value.status == -7 and time.perf_counter() - t0 < 0.2
        ctx.host_batch_wait()
        outs.append(ctx.host_batch_enqueue(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, PROBE, 100)[0])
        ctx.host_batch_wait_all()
        for o in outs:
            assert np.array_equal(o, want)


def test_host_stream_under_stress():
    """The stream of host batches (HostStream / LinkGate / LinkTurn / FoldPool behind rsasa_host_batch_enqueue) with two
    caller threads on ONE context: 2 000 batches - single structures, small batches, 160 k-atom batches and one that is
    cut into sub-batches - enqueued, waited for one by one, waited for all at once and mixed with synchronous batch and
    per-structure calls, in an order drawn at random; every tenth batch is one the engine must refuse (an infinite
    coordinate), wherever it falls in the queue; a full queue is the caller's to drain.  Every batch's values against the
    oracle, every refused batch reported exactly once, nothing hangs; then a context is destroyed with batches still
    queued (they finish first), and the device's memory is back where it was."""
    import random
    import time
    import torch
    import rustsasa_amd
    from rustsasa_amd._capi import ptr
    torch.cuda.synchronize()
    with rustsasa_amd.Context(0) as warm:  # (the runtime's own start-up allocations are not this test's)
        warm.calculate_sasa_soa(*_cases()[0][:5], PROBE, 100)
    free0 = torch.cuda.mem_get_info()[0]
    pool = []
    for n, seed in ((1, 21), (1, 22), (3, 23), (6, 24), (60, 25), (1300, 26)):
        b = bw.synthetic_proteome(n, seed=seed)
        want = po.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, PROBE, 100, 8, threads=0)
        pool.append((b, want, po.residue_sums(want, b.residue_offsets)))
    assert pool[-1][0].n_atoms > 3_000_000  # (large enough to be cut into sub-batches)
    bad = bw.synthetic_proteome(2, seed=27)
    bad_x = bad.x.copy()
    bad_x[5] = np.inf
    weights = [30, 30, 20, 12, 7, 1]
    records, lock = [], threading.Lock()
    errors, refused_seen, refused_sent = [], [0], [0]
    ctx = rustsasa_amd.Context(0)
    lib, h = ctx._lib, ctx._h
    N_BATCHES = 2000
    sent = [0]

    def enqueue(rng):
        with lock:
            if sent[0] >= N_BATCHES:
                return False
            sent[0] += 1
            k = sent[0]
        if k % 10 == 0:
            b, x = bad, bad_x
            rec = None
        else:
            b, want, want_res = pool[rng.choices(range(len(pool)), weights)[0]]
            x = b.x
            rec = (np.full(b.n_atoms, np.nan, np.float32), np.full(b.n_residues, np.nan, np.float32), want, want_res)
        out_a = rec[0] if rec else np.empty(b.n_atoms, np.float32)
        out_r = rec[1] if rec else np.empty(b.n_residues, np.float32)
        for _ in range(10000):
            rc = lib.rsasa_host_batch_enqueue(h, ptr(x), ptr(b.y), ptr(b.z), ptr(b.radius), ptr(b.ids), ptr(b.structure_offsets),
                                              b.n_structures, PROBE, 100, ptr(out_a), ptr(b.residue_offsets), b.n_residues, ptr(out_r))
            if rc != -7:
                break
            wait_one()  # (queue full: somebody has to take a batch out)
        if rc != 0:
            errors.append(("enqueue", k, rc))
            return True
        with lock:
            records.append(rec if rec else (out_a, out_r, x, None))  # (all arrays stay alive until the end)
            if rec is None:
                refused_sent[0] += 1
        return True

    def wait_one():
        rc = lib.rsasa_host_batch_wait(h)
        if rc == -5:
            with lock:
                refused_seen[0] += 1
        elif rc != 0:
            errors.append(("wait", rc))

    def work(tid):
        rng = random.Random(100 + tid)
        try:
            more = True
            while more:
                op = rng.random()
                if op < 0.55:
                    more = enqueue(rng)
                elif op < 0.80:
                    wait_one()
                elif op < 0.84:
                    while True:  # rsasa_host_batch_wait_all returns the FIRST error: count every refused batch through single waits
                        with lock:
                            pending = len(records) - done_count()
                        if pending <= 0:
                            break
                        wait_one()
                elif op < 0.92:
                    b, want, _ = pool[rng.choice((0, 1, 2, 3))]
                    got, _ = ctx.calculate_sasa_batch(b.x, b.y, b.z, b.radius, b.ids, b.structure_offsets, PROBE, 100)
                    if not np.array_equal(got, want):
                        errors.append(("sync batch", tid))
                else:
                    b, want, _ = pool[rng.choice((0, 1))]
                    if not np.array_equal(ctx.calculate_sasa_soa(b.x, b.y, b.z, b.radius, b.ids, PROBE, 100), want):
                        errors.append(("per-structure call", tid))
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    def done_count():
        return sum(1 for r in records if r[3] is None or not np.isnan(r[0][-1]))

    t0 = time.time()
    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in threads), "the stream of host batches hangs"
    for _ in range(len(records) + 8):  # whatever is still queued (a wait with nothing queued returns at once)
        wait_one()
    assert errors == [], errors[:5]
    assert sent[0] == N_BATCHES and refused_sent[0] == N_BATCHES // 10 == refused_seen[0]
    n_ok = 0
    for out_a, out_r, want, want_res in records:
        if want_res is None:
            continue
        assert np.array_equal(out_a, want) and np.array_equal(out_r, want_res)
        n_ok += 1
    assert n_ok == N_BATCHES - N_BATCHES // 10
    # destroyed with batches still queued: they finish, their results are complete
    b, want, want_res = pool[4]
    outs = [(np.full(b.n_atoms, np.nan, np.float32), np.full(b.n_residues, np.nan, np.float32)) for _ in range(5)]
    for a, r in outs:
        assert lib.rsasa_host_batch_enqueue(h, ptr(b.x), ptr(b.y), ptr(b.z), ptr(b.radius), ptr(b.ids), ptr(b.structure_offsets),
                                            b.n_structures, PROBE, 100, ptr(a), ptr(b.residue_offsets), b.n_residues, ptr(r)) == 0
    ctx.close()
    for a, r in outs:
        assert np.array_equal(a, want) and np.array_equal(r, want_res)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert abs(free1 - free0) < 96 * 2 ** 20, f"device memory {free0 >> 20} -> {free1 >> 20} MB"
    print(f"stress: {N_BATCHES} host batches from two threads in {time.time() - t0:.1f} s")


def test_call_combining_opens_a_second_block_when_one_is_full():
    """A combining block holds 196 608 atoms: fourteen threads that each bring a structure of 30 000 atoms (a call's
    largest share is 32 768) cannot share one - the calls that do not fit open the next block, every block is led by the
    call that opened it, and every value is the oracle's.  A call with more atoms than a share runs by itself."""
    import rustsasa_amd
    rng = np.random.default_rng(3)
    structs = []
    for k in range(3):
        b = bw.synthetic_uniform(30_000 + 500 * k, seed=40 + k)
        structs.append((b.x, b.y, b.z, b.radius, b.ids, po.calculate_sasa_internal(b.x, b.y, b.z, b.radius, b.ids, PROBE, 100, 8, threads=0)))
    big = bw.synthetic_uniform(40_000, seed=50)
    want_big = po.calculate_sasa_internal(big.x, big.y, big.z, big.radius, big.ids, PROBE, 100, 8, threads=0)
    errors = []
    b0, c0 = rustsasa_amd.Context.call_combining_stats(0)
    with rustsasa_amd.Context(0) as c:
        c.set_call_combining(50)

        def work(tid):
            try:
                for it in range(6):
                    x, y, z, r, ids, want = structs[(tid + it) % 3]
                    if not np.array_equal(c.calculate_sasa_soa(x, y, z, r, ids, PROBE, 100), want):
                        errors.append((tid, it))
                if tid == 0 and not np.array_equal(c.calculate_sasa_soa(big.x, big.y, big.z, big.radius, big.ids, PROBE, 100), want_big):
                    errors.append("the call that runs alone")
            except Exception as e:  # noqa: BLE001
                errors.append((tid, repr(e)))

        threads = [threading.Thread(target=work, args=(t,)) for t in range(14)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
        assert not any(t.is_alive() for t in threads)
    b1, c1 = rustsasa_amd.Context.call_combining_stats(0)
    assert errors == []
    assert c1 - c0 == 14 * 6 and b1 - b0 >= (14 * 6 * 30_000) // 196_608  # (no block held more than it can)
