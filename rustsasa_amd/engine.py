"""Python host wrapper over the C ABI: one `Context` per GPU.

Mirrors the reference's free function `calculate_sasa_internal`
(src/lib.rs:249-254) and its directory-mode batching (src/main.rs:375,439).
All compute happens in the HIP library; this module only marshals buffers.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _capi
from ._capi import ATOM_DTYPE, DeviceBatch, RsasaError, Timings, check, ptr


def device_count() -> int:
    n = C.c_int(0)
    check(_capi.load().rsasa_device_count(C.byref(n)))
    return n.value


def sphere_points(n_points: int):
    """Golden-section-spiral lattice as uploaded to the GPU (src/lib.rs:43-66)."""
    x = np.empty(n_points, np.float32)
    y = np.empty(n_points, np.float32)
    z = np.empty(n_points, np.float32)
    check(_capi.load().rsasa_sphere_points(n_points, ptr(x), ptr(y), ptr(z)))
    return x, y, z


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _columns(x, y, z, radius, ids, n_expected=None):
    """The SoA columns as contiguous arrays of one common length (the C side sizes every copy from
    the offsets it is given, so a short column would be read past its end)."""
    x, y, z, radius = map(_f32, (x, y, z, radius))
    ids = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint64)
    n = x.shape[0]
    for name, a in (("x", x), ("y", y), ("z", z), ("radius", radius), ("ids", ids)):
        if a is not None and (a.ndim != 1 or a.shape[0] != n):
            raise ValueError(f"{name} must be a 1-D array of {n} entries (the length of x)")
    if n_expected is not None and n != n_expected:
        raise ValueError(f"the offsets cover {n_expected} atoms but the columns hold {n}")
    return x, y, z, radius, ids


def _offsets(name, off):
    off = np.ascontiguousarray(off, dtype=np.uint32)
    if off.ndim != 1 or off.shape[0] < 1:
        raise ValueError(f"{name} must be a 1-D array of at least one entry")
    return off


def _out_buffer(name, buf, n):
    """A caller-provided output array (e.g. pinned host memory) or a fresh one."""
    if buf is None:
        return np.zeros(n, np.float32)
    if not (isinstance(buf, np.ndarray) and buf.dtype == np.float32 and buf.ndim == 1 and
            buf.shape[0] == n and buf.flags["C_CONTIGUOUS"] and buf.flags["WRITEABLE"]):
        raise ValueError(f"{name} must be a writeable contiguous float32 array of {n} entries")
    return buf


class Context:
    """One GPU, one HIP stream, one growable HBM workspace (rsasa_context_t)."""

    def __init__(self, device: int = 0, simd_width: int = 8):
        self._lib = _capi.load()
        h = C.c_void_p()
        check(self._lib.rsasa_context_create(device, C.byref(h)))
        self._h = h
        self.device = device
        if simd_width != 8:
            self.set_simd_width(simd_width)
        self._keepalive = []  # buffers of the batches in flight, oldest first
        self._host_keepalive = []  # the same for host batches (host_batch_enqueue)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rsasa_context_destroy(self._h)  # (waits for queued host batches)
            self._h = None
            self._host_keepalive = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, status):
        check(status, self._h)

    def set_simd_width(self, w: int):
        self._check(self._lib.rsasa_context_set_simd_width(self._h, w))

    def set_call_combining(self, max_wait_us: int = 0):
        """rsasa_context_set_call_combining: per-structure calls of several host threads (ctypes releases the interpreter
        lock during a call) are merged into batch launches; max_wait_us < 0 switches it off."""
        self._check(self._lib.rsasa_context_set_call_combining(self._h, int(max_wait_us)))

    @staticmethod
    def call_combining_stats(device: int = 0):
        """(batches launched, calls merged into them) on `device` since the process started."""
        b, c = C.c_uint64(0), C.c_uint64(0)
        check(_capi.load().rsasa_call_combining_stats(device, C.byref(b), C.byref(c)), None)
        return int(b.value), int(c.value)

    # ---- single structure -------------------------------------------------
    def calculate_sasa_internal(self, atoms: np.ndarray, probe_radius: float = 1.4,
                                n_points: int = 100, threads: int = -1) -> np.ndarray:
        """Drop-in for calculate_sasa_internal(&[Atom], probe, n_points, threads)."""
        atoms = np.ascontiguousarray(atoms, dtype=ATOM_DTYPE)
        out = np.zeros(atoms.shape[0], np.float32)
        self._check(self._lib.rsasa_calculate_sasa_internal(
            self._h, ptr(atoms), atoms.shape[0], probe_radius, n_points, threads, ptr(out)))
        return out

    def calculate_sasa_soa(self, x, y, z, radius, ids=None, probe_radius: float = 1.4,
                           n_points: int = 100) -> np.ndarray:
        x, y, z, radius, ids = _columns(x, y, z, radius, ids)
        out = np.zeros(x.shape[0], np.float32)
        self._check(self._lib.rsasa_calculate_sasa_soa(
            self._h, ptr(x), ptr(y), ptr(z), ptr(radius), ptr(ids), x.shape[0], probe_radius,
            n_points, ptr(out)))
        return out

    # ---- many structures, host buffers -----------------------------------
    def calculate_sasa_batch(self, x, y, z, radius, ids, structure_offsets,
                             probe_radius: float = 1.4, n_points: int = 100,
                             residue_offsets=None, want_atoms: bool = True, atom_out=None,
                             res_out=None):
        """Host buffers in, host buffers out.  `atom_out` / `res_out` may be preallocated float32
        arrays (pinned host memory makes both copy directions asynchronous)."""
        so = _offsets("structure_offsets", structure_offsets)
        n_struct = so.shape[0] - 1
        x, y, z, radius, ids = _columns(x, y, z, radius, ids, int(so[-1]) if n_struct else 0)
        atom_out = _out_buffer("atom_out", atom_out, x.shape[0]) if want_atoms else None
        ro = None
        n_res = 0
        if residue_offsets is not None:
            ro = _offsets("residue_offsets", residue_offsets)
            n_res = ro.shape[0] - 1
            res_out = _out_buffer("res_out", res_out, n_res)
        else:
            res_out = None
        self._check(self._lib.rsasa_calculate_sasa_batch(
            self._h, ptr(x), ptr(y), ptr(z), ptr(radius), ptr(ids), ptr(so), n_struct,
            probe_radius, n_points, ptr(atom_out), ptr(ro), n_res, ptr(res_out)))
        return atom_out, res_out

    # ---- a stream of host batches ------------------------------------------
    def host_batch_enqueue(self, x, y, z, radius, ids, structure_offsets,
                           probe_radius: float = 1.4, n_points: int = 100,
                           residue_offsets=None, want_atoms: bool = True, atom_out=None,
                           res_out=None):
        """rsasa_host_batch_enqueue: as calculate_sasa_batch, but returns once the batch is queued; the results are
        in (atom_out, res_out) - returned here - after the host_batch_wait() that returns this batch.  All arrays
        are kept alive until then; do not touch them in between."""
        so = _offsets("structure_offsets", structure_offsets)
        n_struct = so.shape[0] - 1
        x, y, z, radius, ids = _columns(x, y, z, radius, ids, int(so[-1]) if n_struct else 0)
        atom_out = _out_buffer("atom_out", atom_out, x.shape[0]) if want_atoms else None
        ro = None
        n_res = 0
        if residue_offsets is not None:
            ro = _offsets("residue_offsets", residue_offsets)
            n_res = ro.shape[0] - 1
            res_out = _out_buffer("res_out", res_out, n_res)
        else:
            res_out = None
        self._check(self._lib.rsasa_host_batch_enqueue(
            self._h, ptr(x), ptr(y), ptr(z), ptr(radius), ptr(ids), ptr(so), n_struct,
            probe_radius, n_points, ptr(atom_out), ptr(ro), n_res, ptr(res_out)))
        self._host_keepalive.append((x, y, z, radius, ids, so, ro, atom_out, res_out))
        return atom_out, res_out

    def host_batch_wait(self):
        """Waits for the oldest enqueued host batch and raises its error, if any."""
        try:
            self._check(self._lib.rsasa_host_batch_wait(self._h))
        finally:
            if self._host_keepalive:
                self._host_keepalive.pop(0)

    def host_batch_wait_all(self):
        while self._host_keepalive:
            self.host_batch_wait()

    # ---- MD trajectory: one topology, many frames --------------------------
    def calculate_sasa_trajectory(self, xyz, radius, ids=None, probe_radius: float = 1.4,
                                  n_points: int = 100, residue_offsets=None, want_atoms: bool = True):
        """xyz: [n_frames, n_atoms, 3] float32 (frame-major); returns ([F, N] atom values or None,
        [F, R] residue sums or None)."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        if xyz.ndim != 3 or xyz.shape[2] != 3:
            raise ValueError("xyz must have shape [n_frames, n_atoms, 3]")
        n_frames, n_atoms = xyz.shape[0], xyz.shape[1]
        radius = _f32(radius)
        ids = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint64)
        for name, a in (("radius", radius), ("ids", ids)):
            if a is not None and (a.ndim != 1 or a.shape[0] != n_atoms):
                raise ValueError(f"{name} must be a 1-D array of {n_atoms} entries")
        atom_out = np.zeros((n_frames, n_atoms), np.float32) if want_atoms else None
        ro = res_out = None
        n_res = 0
        if residue_offsets is not None:
            ro = np.ascontiguousarray(residue_offsets, dtype=np.uint32)
            n_res = ro.shape[0] - 1
            res_out = np.zeros((n_frames, n_res), np.float32)
        self._check(self._lib.rsasa_calculate_sasa_trajectory(
            self._h, ptr(xyz), n_frames, n_atoms, ptr(radius), ptr(ids), probe_radius, n_points,
            ptr(atom_out), ptr(ro), n_res, ptr(res_out)))
        return atom_out, res_out

    # ---- many structures, buffers already in HBM --------------------------
    def enqueue_device(self, x, y, z, radius, ids, structure_offsets_host: np.ndarray,
                       out_atom_sasa=None, residue_offsets=None, out_residue_sasa=None,
                       out_neighbor_counts=None, probe_radius: float = 1.4, n_points: int = 100,
                       stream: Optional[int] = None):
        """Enqueue one batch whose arrays are torch CUDA(=HIP) tensors.

        `stream` is a raw hipStream_t value (e.g. torch.cuda.current_stream().cuda_stream);
        None uses the context's own streams.  Up to two batches may be in flight (the library gives
        each its own workspace; a third enqueue first waits for the oldest): wait() waits for the
        OLDEST batch in flight, so enqueue(k + 1) followed by wait() returns batch k's results.
        """
        so = np.ascontiguousarray(structure_offsets_host, dtype=np.uint32)

        def dp(t):
            return None if t is None else t.data_ptr()

        b = DeviceBatch()
        b.x, b.y, b.z, b.radius = dp(x), dp(y), dp(z), dp(radius)
        b.id = dp(ids)
        b.structure_offsets_host = so.ctypes.data
        b.n_structures = so.shape[0] - 1
        b.n_atoms = int(x.shape[0])
        b.residue_offsets = dp(residue_offsets)
        b.n_residues = 0 if residue_offsets is None else int(residue_offsets.shape[0]) - 1
        b.out_atom_sasa = dp(out_atom_sasa)
        b.out_residue_sasa = dp(out_residue_sasa)
        b.out_neighbor_counts = dp(out_neighbor_counts)
        # The library may re-run a batch from its wait(): a batch's buffers stay alive until it has been
        # waited for.  With two batches already in flight rsasa_batch_enqueue waits for the oldest itself.
        keep = (so, x, y, z, radius, ids, residue_offsets, out_atom_sasa, out_residue_sasa,
                out_neighbor_counts)
        full = len(self._keepalive) >= 2
        try:
            self._check(self._lib.rsasa_batch_enqueue(self._h, C.byref(b), probe_radius, n_points,
                                                      C.c_void_p(stream) if stream else None))
        finally:
            if full:
                self._keepalive.pop(0)
        self._keepalive.append(keep)

    def wait(self):
        """Waits for the oldest batch in flight (no batch in flight: returns at once)."""
        try:
            self._check(self._lib.rsasa_batch_wait(self._h))
        finally:
            if self._keepalive:
                self._keepalive.pop(0)

    def wait_all(self):
        while self._keepalive:
            self.wait()

    # ---- measurement -------------------------------------------------------
    def enable_timing(self, enable: bool = True):
        self._check(self._lib.rsasa_context_enable_timing(self._h, 1 if enable else 0))

    def timings(self) -> dict:
        t = Timings()
        self._check(self._lib.rsasa_context_get_timings(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in Timings._fields_}

    def ids_kept(self) -> int:
        """Structures of the context's last checked (sub-)batch that kept their ids (rsasa_context_ids_kept)."""
        n = C.c_uint64(0)
        self._check(self._lib.rsasa_context_ids_kept(self._h, C.byref(n)))
        return int(n.value)

    def ids_dropped(self) -> int:
        """(Sub-)batches that ran without their ids because the ids of every structure increased strictly
        (rsasa_context_ids_dropped)."""
        n = C.c_uint64(0)
        self._check(self._lib.rsasa_context_ids_dropped(self._h, C.byref(n)))
        return int(n.value)


def make_atoms(x, y, z, radius, ids) -> np.ndarray:
    """Packs SoA columns into rsasa_atom_t records."""
    a = np.zeros(len(x), ATOM_DTYPE)
    a["position"][:, 0] = x
    a["position"][:, 1] = y
    a["position"][:, 2] = z
    a["radius"] = radius
    a["id"] = ids
    return a


__all__ = ["Context", "RsasaError", "device_count", "sphere_points", "make_atoms", "ATOM_DTYPE"]
