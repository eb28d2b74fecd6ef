"""Row-partitioned CG across the GPUs of one node: one process per GPU, torch.distributed over
RCCL/xGMI (backend "nccl" on ROCm), gloo on CPU for the tests.

The reference is a serial library (no parallelism of any kind), so this layer
is new design: rank p owns a contiguous block of rows of A and the matching
slices of x, b, r, c, u; per iteration the only exchanges are
  * one halo exchange of u before the SpMV (for the 3D Laplacian cut into z-slabs: N^2 doubles to
    each neighbour), and
  * two all-gathers of ONE scalar per rank -- the local dot(u, c) and the local |r|^2 -- which
    every rank then sums in rank order 0..P-1, so all ranks hold bit-identical alpha / residual.
All arithmetic runs in libmik.so (``mik_cgd_*`` phases); this module only orchestrates phases and
collectives, and never synchronises the host except to read the residual.

The compute "engine" is pluggable so that the orchestration, the partitioning and the exchange
plans can be exercised without a GPU: ``HipEngine`` is the product; the CPU tests inject a numpy
test double with the same phase interface (tests/dist_double.py).
"""
from __future__ import annotations

import ctypes as C
import json
import os
import sys
import time
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

_vp = C.c_void_p

# phases of mik_cgd_phase (include/mik.h)
INIT_A, INIT_B, INIT_C = 10, 11, 12
STEP_A, STEP_B, STEP_C, STEP_D = 0, 1, 2, 3
STEP_B_INTERIOR, STEP_B_REST = 4, 5      # step B split so that the halo exchange overlaps the interior rows


# ==============================================================================================
# partitioning (host, numpy)
# ==============================================================================================
def partition_rows(n: int, nranks: int, align: int = 1) -> np.ndarray:
    """Contiguous row blocks of (almost) equal size, cut on multiples of ``align`` (e.g. N^2 so that
    a 3D Laplacian is cut into z-slabs).  Returns nranks + 1 offsets."""
    units = -(-n // align)
    cuts = [min(n, align * ((units * p) // nranks)) for p in range(nranks + 1)]
    cuts[-1] = n
    return np.asarray(cuts, np.int64)


@dataclass
class HaloPlan:
    """What rank `rank` receives into / sends from its extended vector."""
    rank: int
    nranks: int
    n_loc: int
    ghost_gids: np.ndarray                                   # sorted global ids of the halo entries
    recv: List[tuple] = field(default_factory=list)          # (peer, offset into ghost region, count)
    send: List[tuple] = field(default_factory=list)          # (peer, offset into send buffer, count)
    send_idx: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))   # local indices, concatenated per peer

    @property
    def n_ghost(self) -> int:
        return int(self.ghost_gids.size)

    @property
    def n_send(self) -> int:
        return int(self.send_idx.size)


def localize_block(ptr, idx, offsets, rank):
    """Rows offsets[rank]:offsets[rank+1] of a CSR matrix with GLOBAL 0-based column ids -> local
    column ids ([0, n_loc) owned, [n_loc, n_loc + n_ghost) halo, halo sorted by global id) and the
    receive side of the halo plan."""
    r0, r1 = int(offsets[rank]), int(offsets[rank + 1])
    n_loc = r1 - r0
    idx = np.asarray(idx, np.int64)
    owned = (idx >= r0) & (idx < r1)
    ghost_gids = np.unique(idx[~owned])
    local = np.empty(idx.shape, np.int64)
    local[owned] = idx[owned] - r0
    local[~owned] = n_loc + np.searchsorted(ghost_gids, idx[~owned])
    plan = HaloPlan(rank, len(offsets) - 1, n_loc, ghost_gids)
    owner = np.searchsorted(offsets, ghost_gids, side="right") - 1
    for peer in np.unique(owner):
        sel = np.nonzero(owner == peer)[0]
        plan.recv.append((int(peer), int(sel[0]), int(sel.size)))      # contiguous: ghost ids are sorted
    return local, plan


def complete_plan(plan: HaloPlan, offsets, needs_of_peers):
    """``needs_of_peers[q]`` = sorted global ids rank q wants (its ghost_gids).  Fills the send side."""
    r0, r1 = int(offsets[plan.rank]), int(offsets[plan.rank + 1])
    chunks, off = [], 0
    plan.send = []
    for q, gids in enumerate(needs_of_peers):
        if q == plan.rank or gids is None:
            continue
        mine = gids[(gids >= r0) & (gids < r1)]
        if mine.size:
            plan.send.append((q, off, int(mine.size)))
            chunks.append((mine - r0).astype(np.int32))
            off += int(mine.size)
    plan.send_idx = np.concatenate(chunks) if chunks else np.zeros(0, np.int32)
    return plan


def interior_row_blocks(ptr, local_idx, n_loc: int, block: int = 256):
    """Row-blocks [a, b) (of `block` rows) that contain no row referencing a halo column (local index >= n_loc),
    when the blocks that do form a prefix and a suffix of the rank's rows (slab partitions); else None."""
    nb = (n_loc + block - 1) // block
    if not isinstance(ptr, np.ndarray) and hasattr(ptr, "data_ptr"):           # torch tensors on the device: same steps there
        import torch
        ghost_entries = torch.nonzero(local_idx >= n_loc).flatten()
        if ghost_entries.numel() == 0:
            return (0, nb) if nb else None
        rows = torch.searchsorted(ptr, ghost_entries, right=True) - 1
        blocks = torch.unique(torch.div(rows, block, rounding_mode="floor")).cpu().numpy()
    else:
        ptr = np.asarray(ptr)
        ghost_entries = np.nonzero(np.asarray(local_idx) >= n_loc)[0]
        if ghost_entries.size == 0:
            return (0, nb) if nb else None
        rows = np.searchsorted(ptr, ghost_entries, side="right") - 1
        blocks = np.unique(rows // block)
    a = 0
    while a < blocks.size and blocks[a] == a:
        a += 1
    rest = blocks[a:]
    b = nb - rest.size
    if rest.size and not np.array_equal(rest, np.arange(b, nb)):
        return None
    return (a, b) if b > a else None


# ==============================================================================================
# communicator
# ==============================================================================================
class TorchComm:
    """torch.distributed: "nccl" (= RCCL over xGMI) for device tensors, "gloo" for the CPU tests."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.size = dist.get_rank(), dist.get_world_size()
        # MIK_DIST_FORCE_COLLECTIVES=1: issue the collectives even in a world of one (exercises the
        # backend's call paths on a single-GPU box)
        self.force = os.environ.get("MIK_DIST_FORCE_COLLECTIVES", "0") == "1"
        # gloo cannot move device tensors: stage them through host copies (slow; used to run several
        # ranks on ONE GPU for verification, where RCCL refuses duplicate devices)
        self.staged = dist.get_backend() == "gloo"

    def all_gather_objects(self, obj):
        out = [None] * self.size
        self.dist.all_gather_object(out, obj)
        return out

    def exchange(self, plan: HaloPlan, send_buf, ghost_view):
        """send_buf / ghost_view: 1-D tensors (packed halo values; tail of the extended vector)."""
        if (self.size == 1 and not self.force) or (not plan.send and not plan.recv):
            return
        dist = self.dist
        staged = self.staged and send_buf.is_cuda
        if staged:
            import torch
            torch.cuda.current_stream().synchronize()
            dev_ghost, send_buf, ghost_view = ghost_view, send_buf.cpu(), ghost_view.cpu()
        ops = []
        for peer, off, cnt in plan.recv:
            ops.append(dist.P2POp(dist.irecv, ghost_view[off:off + cnt], peer))
        for peer, off, cnt in plan.send:
            ops.append(dist.P2POp(dist.isend, send_buf[off:off + cnt], peer))
        for req in dist.batch_isend_irecv(ops):
            req.wait()             # nccl: makes the current stream wait; gloo: blocks the host
        if staged:
            dev_ghost.copy_(ghost_view)

    def exchange_begin(self, plan: HaloPlan, send_buf, ghost_view):
        """Start the halo exchange and return a handle for ``exchange_end``.  With RCCL the transfers run on the
        backend's own stream (ordered after what is already enqueued on the current stream) and the current stream
        is NOT made to wait yet -- kernels enqueued before ``exchange_end`` overlap the transfer.  Host-staged
        modes complete the exchange right here."""
        if (self.size == 1 and not self.force) or (not plan.send and not plan.recv):
            return None
        if self.staged and send_buf.is_cuda or not send_buf.is_cuda:
            self.exchange(plan, send_buf, ghost_view)
            return None
        dist = self.dist
        ops = [dist.P2POp(dist.irecv, ghost_view[off:off + cnt], peer) for peer, off, cnt in plan.recv]
        ops += [dist.P2POp(dist.isend, send_buf[off:off + cnt], peer) for peer, off, cnt in plan.send]
        return dist.batch_isend_irecv(ops)

    def exchange_end(self, handle):
        for req in handle or ():
            req.wait()             # the current stream waits for the transfers

    def all_gather_scalar(self, all_t, rank_slot):
        """all_t[p] <- rank p's all_t[p] (in-place all-gather of one scalar per rank)."""
        if self.size == 1 and not self.force:
            return
        if all_t.is_cuda and self.staged:
            import torch
            torch.cuda.current_stream().synchronize()
            host = all_t.cpu()
            self.dist.all_gather(list(host.chunk(self.size)), host[self.rank:self.rank + 1].clone())
            all_t.copy_(host)
        elif all_t.is_cuda:
            self.dist.all_gather_into_tensor(all_t, rank_slot)       # in-place form: slot [rank] is the input
        else:
            self.dist.all_gather(list(all_t.chunk(self.size)), rank_slot.clone())

    def all_gather_host(self, values: np.ndarray) -> np.ndarray:
        """(P, count) array of every rank's host scalars, rank order (blocking)."""
        values = np.ascontiguousarray(values)
        if self.size == 1 and not self.force:
            return values[None, :].copy()
        import torch
        if self.staged:
            out = [torch.empty(values.size, dtype=torch.from_numpy(values).dtype) for _ in range(self.size)]
            self.dist.all_gather(out, torch.from_numpy(values.copy()))
            return np.stack([o.numpy() for o in out])
        dev = torch.device("cuda", torch.cuda.current_device())
        mine = torch.from_numpy(values.copy()).to(dev)
        out = torch.empty(self.size * values.size, dtype=mine.dtype, device=dev)
        self.dist.all_gather_into_tensor(out, mine)
        return out.cpu().numpy().reshape(self.size, values.size)

    def barrier(self):
        if self.size > 1:
            self.dist.barrier()


class SelfComm:
    """World of one process (no torch.distributed needed)."""
    rank, size = 0, 1

    def all_gather_objects(self, obj):
        return [obj]

    def exchange(self, plan, send_buf, ghost_view):
        pass

    def exchange_begin(self, plan, send_buf, ghost_view):
        return None

    def exchange_end(self, handle):
        pass

    def all_gather_scalar(self, all_t, rank_slot):
        pass

    def all_gather_host(self, values):
        return np.ascontiguousarray(values)[None, :].copy()

    def barrier(self):
        pass


class ThreadComm:
    """P virtual ranks = P host threads of ONE process sharing one GPU (each with its own ctx and
    stream).  The blocking callbacks of the partitioned GMRES handle meet at a barrier -- this is how
    that path is verified on a single-GPU box, where RCCL refuses two ranks on one device.
    ``ThreadComm.world(P)`` returns the P per-rank communicators."""

    def __init__(self, shared, rank):
        self.shared, self.rank, self.size = shared, rank, shared["P"]

    @staticmethod
    def world(P: int):
        import threading
        shared = {"P": P, "barrier": threading.Barrier(P, timeout=120), "slots": [None] * P, "send": [None] * P, "plans": [None] * P}
        return [ThreadComm(shared, r) for r in range(P)]

    def all_gather_objects(self, obj):
        return self._gather(obj)

    def _gather(self, obj):
        sh = self.shared
        sh["slots"][self.rank] = obj
        sh["barrier"].wait()
        out = list(sh["slots"])
        sh["barrier"].wait()
        return out

    def all_gather_host(self, values):
        return np.stack(self._gather(np.ascontiguousarray(values).copy()))

    def exchange(self, plan, send_buf, ghost_view):
        import torch
        sh = self.shared
        torch.cuda.current_stream().synchronize()          # my packed halo is complete
        sh["send"][self.rank], sh["plans"][self.rank] = send_buf, plan
        sh["barrier"].wait()
        for peer, off, cnt in plan.recv:
            soff = next(o for (q, o, c) in sh["plans"][peer].send if q == self.rank)
            ghost_view[off:off + cnt].copy_(sh["send"][peer][soff:soff + cnt])
        torch.cuda.current_stream().synchronize()          # my reads of the peers' buffers are complete
        sh["barrier"].wait()

    def barrier(self):
        self.shared["barrier"].wait()


def rank_ordered_sum(parts: np.ndarray) -> np.ndarray:
    """((p_0 + p_1) + p_2) + ... along axis 0 in the array's own dtype: the order every rank uses, so all
    ranks obtain identical bits (include/mik.h: mik_reduce_fn)."""
    tot = parts[0].copy()
    for q in range(1, parts.shape[0]):
        tot = tot + parts[q]
    return tot


# ==============================================================================================
# product engine: libmik.so
# ==============================================================================================
class HipEngine:
    """Owns the device state of one rank and enqueues ``mik_cgd`` phases on the current torch stream."""

    def __init__(self, pkg, ptr, local_idx, val, plan: HaloPlan, b_loc, x_loc=None, *, abstol, reltol, maxiter, device=0,
                 stream=None, layout=None):
        import torch
        self.pkg, self.torch, self.plan = pkg, torch, plan
        L = pkg.lib()
        self.L = L
        dtype = np.dtype({torch.float64: np.float64, torch.float32: np.float32}[val.dtype]) if isinstance(val, torch.Tensor) else np.dtype(val.dtype)
        tdt = {np.dtype(np.float64): torch.float64, np.dtype(np.float32): torch.float32}[dtype]
        dev = torch.device("cuda", device)
        self.ctx = pkg.HipContext(device)
        self.stream = stream if stream is not None else torch.cuda.Stream(device=dev)   # loopback ranks share one stream
        self.ctx.set_stream(self.stream.cuda_stream)
        n_loc, n_ext = plan.n_loc, plan.n_loc + plan.n_ghost
        if isinstance(val, torch.Tensor):               # the rank's block was generated on the device (build_rank_problem)
            torch.cuda.synchronize(dev)                 # the generator ran on torch's stream, the upload runs on the ctx stream
            self.A = pkg.HipCSR.from_device(n_loc, n_ext, int(val.numel()), ptr.data_ptr(), local_idx.data_ptr(), val.data_ptr(), dtype,
                                            index_base=0, is_csc=False, ctx=self.ctx)
        else:
            self.A = pkg.HipCSR(n_loc, n_ext, ptr, local_idx, val, index_base=0, is_csc=False, ctx=self.ctx)
        if layout not in (None, "auto"):
            self.A.set_layout(layout)                   # "csr": the iterable runs on the plain CSR arrays (bench.py's contract loop)
        with torch.cuda.stream(self.stream):
            self.u_ext = torch.zeros(max(n_ext, 1), dtype=tdt, device=dev)           # receives the halo in place
            self.send_buf = torch.zeros(max(plan.n_send, 1), dtype=tdt, device=dev)
            self.dot_all = torch.zeros(plan.nranks, dtype=tdt, device=dev)
            self.rr_all = torch.zeros(plan.nranks, dtype=tdt, device=dev)
            self.send_idx = torch.from_numpy(plan.send_idx.astype(np.int32)).to(dev) if plan.n_send else torch.zeros(1, dtype=torch.int32, device=dev)
        self.b = pkg.HipVector.from_numpy(np.ascontiguousarray(b_loc, dtype), self.ctx)
        self.x = pkg.HipVector.from_numpy(np.ascontiguousarray(x_loc, dtype), self.ctx) if x_loc is not None else pkg.HipVector(n_loc, dtype, self.ctx).fill_(0)
        self.r = pkg.HipVector(n_loc, dtype, self.ctx)
        self.c = pkg.HipVector(n_loc, dtype, self.ctx)
        h = _vp()
        pkg._lib.check(L.mik_cgd_create(self.ctx.handle, self.A.handle, _vp(self.x.ptr), _vp(self.b.ptr), _vp(self.u_ext.data_ptr()),
                                        _vp(self.r.ptr), _vp(self.c.ptr), _vp(self.send_idx.data_ptr()), plan.n_send,
                                        _vp(self.send_buf.data_ptr()), _vp(self.dot_all.data_ptr()), _vp(self.rr_all.data_ptr()),
                                        plan.rank, plan.nranks, float(abstol), float(reltol), int(maxiter), int(x_loc is None),
                                        C.byref(h)), "mik_cgd_create", self.ctx.handle)
        self.handle = h
        # overlap of the halo exchange with the rows that need no halo (MIK_DIST_OVERLAP=0 switches it off)
        self.overlap = False
        rng = interior_row_blocks(ptr, local_idx, n_loc) if os.environ.get("MIK_DIST_OVERLAP", "1") != "0" else None
        if rng is not None and L.mik_cgd_set_interior(self.handle, int(rng[0]), int(rng[1])) == 0:
            self.overlap = True
            self.interior = rng
        self.ctx.synchronize()

    # tensors the communicator works on
    def ghost_view(self):
        return self.u_ext[self.plan.n_loc:self.plan.n_loc + self.plan.n_ghost]

    def dot_slot(self):
        return self.dot_all[self.plan.rank:self.plan.rank + 1]

    def rr_slot(self):
        return self.rr_all[self.plan.rank:self.plan.rank + 1]

    def phase(self, ph: int, iteration: int = 0):
        self.pkg._lib.check(self.L.mik_cgd_phase(self.handle, ph, int(iteration)), "mik_cgd_phase", self.ctx.handle)

    def wait(self, cap: int = 1024):
        res, tol = C.c_double(), C.c_double()
        done = C.c_int()
        steps = C.c_int64()
        hist = np.empty(cap, np.float64)
        self.pkg._lib.check(self.L.mik_cgd_wait(self.handle, C.byref(res), C.byref(tol), C.byref(done),
                                                hist.ctypes.data_as(C.POINTER(C.c_double)), cap, C.byref(steps)), "mik_cgd_wait", self.ctx.handle)
        return res.value, tol.value, bool(done.value), hist[:steps.value].copy()

    def stream_ctx(self):
        return self.torch.cuda.stream(self.stream)

    def solution(self) -> np.ndarray:
        return self.x.to_numpy()

    def close(self):
        if getattr(self, "handle", None):
            self.L.mik_cgd_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ==============================================================================================
# the distributed iterable
# ==============================================================================================
class DistCGIterable:
    """``CGIterable`` (src/cg.jl:5-16) over a row partition.  ``iterate`` follows src/cg.jl:43-66;
    construction follows ``cg_iterator!`` (src/cg.jl:120-155)."""

    def __init__(self, engine, comm, *, maxiter):
        self.e, self.comm, self.maxiter = engine, comm, int(maxiter)
        self.mv_products = 0
        e = engine
        with e.stream_ctx():
            e.phase(INIT_A)
            comm.exchange(e.plan, e.send_buf, e.ghost_view())
            e.phase(INIT_B)
            comm.all_gather_scalar(e.rr_all, e.rr_slot())
            e.phase(INIT_C)
        self.residual, self.tol, _, _ = e.wait()
        self.prev_residual = 1.0

    def converged(self) -> bool:
        return self.residual <= self.tol

    def done(self, iteration: int) -> bool:
        return iteration >= self.maxiter or self.converged()

    def _enqueue_step(self, iteration: int):
        e, comm = self.e, self.comm
        e.phase(STEP_A)
        if getattr(e, "overlap", False):
            pending = comm.exchange_begin(e.plan, e.send_buf, e.ghost_view())
            e.phase(STEP_B_INTERIOR)                    # runs while the halo is in flight
            comm.exchange_end(pending)
            e.phase(STEP_B_REST)
        else:
            comm.exchange(e.plan, e.send_buf, e.ghost_view())
            e.phase(STEP_B)
        comm.all_gather_scalar(e.dot_all, e.dot_slot())
        e.phase(STEP_C)
        comm.all_gather_scalar(e.rr_all, e.rr_slot())
        e.phase(STEP_D, iteration)

    def iterate_many(self, iteration: int, max_steps: int) -> np.ndarray:
        """Up to max_steps iterate() calls with one host wait; the stopping test of src/cg.jl:36 runs on
        the device after every step (identically on every rank) and turns later steps into no-ops."""
        if max_steps <= 0 or self.done(iteration):
            return np.zeros(0)
        max_steps = min(int(max_steps), self.maxiter - iteration, 1024)
        with self.e.stream_ctx():
            for j in range(max_steps):
                self._enqueue_step(iteration + j)
        res, _, _, hist = self.e.wait()
        if hist.size:
            self.prev_residual = hist[-2] if hist.size > 1 else self.residual
            self.residual = res
            self.mv_products += hist.size
        return hist

    def iterate(self, iteration: int = 0):
        h = self.iterate_many(iteration, 1)
        return None if h.size == 0 else (self.residual, iteration + 1)

    def __iter__(self):
        iteration = 0
        while (nxt := self.iterate(iteration)) is not None:
            _, iteration = nxt
            yield self.residual


def _plan_arrays(plan: HaloPlan):
    """(n, peer int32[], off int64[], cnt int64[]) x 2 for mik_cgd_set_halo_plan / mik_comm_halo (kept alive by the caller)."""
    def pack(segs):
        peer = np.asarray([s[0] for s in segs], np.int32)
        off = np.asarray([s[1] for s in segs], np.int64)
        cnt = np.asarray([s[2] for s in segs], np.int64)
        return peer, off, cnt
    return pack(plan.recv), pack(plan.send)


def register_halo_plan(pkg, engine):
    """Hand the rank's halo plan to the library (``mik_cgd_set_halo_plan``): from then on libmik.so runs the exchanges."""
    (rp, ro, rc), (sp, so, sc) = _plan_arrays(engine.plan)
    ip, lp = C.POINTER(C.c_int), C.POINTER(C.c_int64)
    pkg._lib.check(engine.L.mik_cgd_set_halo_plan(engine.handle, rp.size, rp.ctypes.data_as(ip), ro.ctypes.data_as(lp), rc.ctypes.data_as(lp),
                                                  sp.size, sp.ctypes.data_as(ip), so.ctypes.data_as(lp), sc.ctypes.data_as(lp)),
                   "mik_cgd_set_halo_plan", engine.ctx.handle)


class NativeComm:
    """``mik_comm``: the transports INSIDE libmik.so (include/mik.h "Transport 1" and "Transport 3").  ``bootstrap`` is any
    communicator of this module (TorchComm over gloo or nccl, SelfComm): it only carries small host objects between the ranks --
    rank 0's 128-byte ncclUniqueId, the 64-byte HIP IPC handles of the mailboxes and landing buffers -- what MPI.jl's ``bcast`` /
    ``Allgather`` would do for a Julia host.

    ``transport``: "rccl" (halo by ncclSend / ncclRecv, scalars by ncclAllGather), "rccl+mailbox" (halo by RCCL, the two scalars of a
    step through peer-mapped mailboxes), "mailbox" (no RCCL at all: scalars through the mailboxes, the halo pushed into the
    neighbours' landing buffers -- the only transport that lets several ranks share one GPU).  ``force_rccl`` creates a real RCCL communicator even
    in a world of one (exercises the library's RCCL call path on a single-GPU box)."""

    def __init__(self, pkg, ctx, bootstrap, *, force_rccl=False, transport="rccl"):
        self.pkg, self.ctx, self.L = pkg, ctx, pkg.lib()
        self.boot = bootstrap
        self.rank, self.size = bootstrap.rank, bootstrap.size
        self.transport = transport
        if transport not in ("rccl", "rccl+mailbox", "mailbox"):
            raise ValueError(f"NativeComm: unknown transport {transport!r}")
        ident = None
        if transport != "mailbox" and (self.size > 1 or force_rccl):
            buf = C.create_string_buffer(128)
            payload = bytes(buf.raw)
            if self.rank == 0:                        # a failure on rank 0 is told to everybody instead of leaving them in the gather
                try:
                    pkg._lib.check(self.L.mik_comm_unique_id(buf), "mik_comm_unique_id")
                    payload = bytes(buf.raw)
                except Exception as exc:              # noqa: BLE001
                    payload = f"mik_comm_unique_id failed: {exc}"
            ident = bootstrap.all_gather_objects(payload)[0]
            if isinstance(ident, str):
                raise RuntimeError(ident)
        h = _vp()
        pkg._lib.check(self.L.mik_comm_create(ctx.handle, ident, self.rank, self.size, C.byref(h)), "mik_comm_create", ctx.handle)
        self.handle = h
        if transport != "rccl":
            mine = C.create_string_buffer(64)
            pkg._lib.check(self.L.mik_comm_mailbox_export(self.handle, mine), "mik_comm_mailbox_export", ctx.handle)
            handles = b"".join(bootstrap.all_gather_objects(bytes(mine.raw)))
            pkg._lib.check(self.L.mik_comm_mailbox_connect(self.handle, handles), "mik_comm_mailbox_connect", ctx.handle)
            bootstrap.barrier()

    def uses_rccl(self) -> bool:
        out = C.c_int()
        self.L.mik_comm_info(self.handle, None, None, C.byref(out))
        return bool(out.value)

    def mailbox(self):
        """(connected, fine-grained) of the communicator's mailbox"""
        a, b = C.c_int(), C.c_int()
        self.L.mik_comm_mailbox_info(self.handle, C.byref(a), C.byref(b))
        return bool(a.value), bool(b.value)

    def _landing_targets(self, plan, info):
        """per SEND segment of `plan`: where it lands in the receiver's ghost region (the offset of the matching receive segment there).
        info[q] = (..., recv segments of rank q)"""
        dst, taken = [], {}
        for peer, _off, cnt in plan.send:
            cands = [sg for sg in info[peer][-1] if sg[0] == self.rank]
            k = taken.get(peer, 0)
            taken[peer] = k + 1
            if k >= len(cands) or cands[k][2] != cnt:
                raise RuntimeError(f"halo plans disagree: rank {self.rank} sends {cnt} entries to rank {peer}, which expects {cands}")
            dst.append(cands[k][1])
        return np.array(dst if dst else [0], np.int64)

    def connect_ghosts(self, engine):
        """transport "mailbox": every rank allocates the landing buffer of its halo and exports it (mik_cgd_ghost_export), every sender learns
        where its segments land (mik_cgd_connect_ghosts).  Collective; after mik_cgd_set_halo_plan + mik_cgd_set_comm."""
        plan, L, ctx = engine.plan, self.L, engine.ctx
        hbuf = C.create_string_buffer(64)
        self.pkg._lib.check(L.mik_cgd_ghost_export(engine.handle, hbuf), "mik_cgd_ghost_export", ctx.handle)
        info = self.boot.all_gather_objects((bytes(hbuf.raw), int(plan.n_ghost), [tuple(int(v) for v in sg) for sg in plan.recv]))
        handles = b"".join(i[0] for i in info)
        counts = np.array([i[1] for i in info], np.int64)
        dst = self._landing_targets(plan, info)
        self.pkg._lib.check(L.mik_cgd_connect_ghosts(engine.handle, handles, counts.ctypes.data_as(C.POINTER(C.c_int64)),
                                                     dst.ctypes.data_as(C.POINTER(C.c_int64))), "mik_cgd_connect_ghosts", ctx.handle)
        self.boot.barrier()

    def make_link(self, plan: HaloPlan, dtype):
        """A connected ``mik_plink`` for `plan` on this communicator (collective): what ``mik_partition.link`` of a row-partitioned GMRES
        iterable takes -- halo and rank-ordered sums then run on the device, without host callbacks."""
        L, ctx = self.L, self.ctx
        (rp, ro, rc), (sp, so, sc) = _plan_arrays(plan)
        ip, lp = C.POINTER(C.c_int), C.POINTER(C.c_int64)
        h = _vp()
        self.pkg._lib.check(L.mik_plink_create(self.handle, self.pkg._lib.MIK_F64 if np.dtype(dtype) == np.float64 else self.pkg._lib.MIK_F32, int(plan.n_ghost),
                                               rp.size, rp.ctypes.data_as(ip), ro.ctypes.data_as(lp), rc.ctypes.data_as(lp),
                                               sp.size, sp.ctypes.data_as(ip), so.ctypes.data_as(lp), sc.ctypes.data_as(lp), C.byref(h)), "mik_plink_create", ctx.handle)
        hbuf = C.create_string_buffer(64)
        self.pkg._lib.check(L.mik_plink_export(h, hbuf), "mik_plink_export", ctx.handle)
        info = self.boot.all_gather_objects((bytes(hbuf.raw), int(plan.n_ghost), [tuple(int(v) for v in sg) for sg in plan.recv]))
        handles = b"".join(i[0] for i in info)
        counts = np.array([i[1] for i in info], np.int64)
        dst = self._landing_targets(plan, info)
        self.pkg._lib.check(L.mik_plink_connect(h, handles, counts.ctypes.data_as(lp), dst.ctypes.data_as(lp)), "mik_plink_connect", ctx.handle)
        self.boot.barrier()
        return h

    def close(self):
        if getattr(self, "handle", None):
            self.L.mik_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NativeDistCGIterable:
    """``CGIterable`` over a row partition with the exchanges run by libmik.so itself: ``iterate_many`` is ONE C call
    (``mik_cgd_iterate_many``: pack, ncclSend/ncclRecv halo on a side stream overlapped with the interior rows, SpMV,
    ncclAllGather of one scalar per rank, update, ncclAllGather, stopping test -- per step, no host code in between)."""

    def __init__(self, pkg, engine, native_comm: NativeComm, *, maxiter):
        self.pkg, self.e, self.comm, self.maxiter = pkg, engine, native_comm, int(maxiter)
        self.mv_products = 0
        register_halo_plan(pkg, engine)
        pkg._lib.check(engine.L.mik_cgd_set_comm(engine.handle, native_comm.handle), "mik_cgd_set_comm", engine.ctx.handle)
        if native_comm.transport == "mailbox":
            native_comm.connect_ghosts(engine)
        res, tol = C.c_double(), C.c_double()
        with engine.stream_ctx():
            pkg._lib.check(engine.L.mik_cgd_init(engine.handle, C.byref(res), C.byref(tol)), "mik_cgd_init", engine.ctx.handle)
        self.residual, self.tol, self.prev_residual = res.value, tol.value, 1.0

    def converged(self) -> bool:
        return self.residual <= self.tol

    def done(self, iteration: int) -> bool:
        return iteration >= self.maxiter or self.converged()

    def iterate_many(self, iteration: int, max_steps: int) -> np.ndarray:
        if max_steps <= 0 or self.done(iteration):
            return np.zeros(0)
        max_steps = min(int(max_steps), self.maxiter - iteration, 1024)
        hist = np.empty(max_steps, np.float64)
        steps = C.c_int64()
        e = self.e
        with e.stream_ctx():
            self.pkg._lib.check(e.L.mik_cgd_iterate_many(e.handle, int(iteration), max_steps, hist.ctypes.data_as(C.POINTER(C.c_double)),
                                                         C.byref(steps)), "mik_cgd_iterate_many", e.ctx.handle)
        hist = hist[:steps.value].copy()
        if hist.size:
            self.prev_residual = hist[-2] if hist.size > 1 else self.residual
            self.residual = float(hist[-1])
            self.mv_products += hist.size
        return hist

    def iterate(self, iteration: int = 0):
        h = self.iterate_many(iteration, 1)
        return None if h.size == 0 else (self.residual, iteration + 1)


class GroupCG:
    """P ranks driven by ONE host thread through ``mik_cgd_group_*`` (include/mik.h "Transport 2"): every engine has its own
    ctx (and may sit on its own GPU: peer copies over xGMI), the library orders halos and scalar gathers with events.
    On a single-GPU box this runs P virtual ranks on one device -- the check that the library's step routine equals the
    partition-aware oracle bit for bit."""

    def __init__(self, pkg, engines: Sequence, *, maxiter):
        self.pkg, self.engines, self.maxiter = pkg, list(engines), int(maxiter)
        for e in self.engines:
            register_halo_plan(pkg, e)
        self.P = len(self.engines)
        self.handles = (_vp * self.P)(*[e.handle for e in self.engines])
        self.L = self.engines[0].L
        res, tol = C.c_double(), C.c_double()
        pkg._lib.check(self.L.mik_cgd_group_init(self.handles, self.P, C.byref(res), C.byref(tol)), "mik_cgd_group_init", self.engines[0].ctx.handle)
        self.residual, self.tol = res.value, tol.value

    def done(self, iteration):
        return iteration >= self.maxiter or self.residual <= self.tol

    def halo_early(self):
        """per rank (runs, rows, merged) of ``mik_cgd_halo_early``: the rows a rank updates and packs ahead of the sweep"""
        out = []
        for e in self.engines:
            runs, rows, merged = C.c_int(), C.c_int64(), C.c_int()
            self.pkg._lib.check(self.L.mik_cgd_halo_early(e.handle, C.byref(runs), C.byref(rows), C.byref(merged)), "mik_cgd_halo_early", e.ctx.handle)
            out.append((runs.value, rows.value, bool(merged.value)))
        return out

    def iterate_many(self, iteration: int, max_steps: int) -> np.ndarray:
        if max_steps <= 0 or self.done(iteration):
            return np.zeros(0)
        max_steps = min(int(max_steps), self.maxiter - iteration, 1024)
        hist = np.empty(max_steps, np.float64)
        steps = C.c_int64()
        self.pkg._lib.check(self.L.mik_cgd_group_iterate_many(self.handles, self.P, int(iteration), max_steps,
                                                              hist.ctypes.data_as(C.POINTER(C.c_double)), C.byref(steps)),
                            "mik_cgd_group_iterate_many", self.engines[0].ctx.handle)
        hist = hist[:steps.value].copy()
        if hist.size:
            self.residual = float(hist[-1])
        return hist

    def solve(self) -> np.ndarray:
        out, iteration = [], 0
        while True:
            h = self.iterate_many(iteration, 64)
            if h.size == 0:
                break
            out.append(h)
            iteration += h.size
        return np.concatenate(out) if out else np.zeros(0)

    def solution(self) -> np.ndarray:
        return np.concatenate([e.solution() for e in self.engines])

    def close(self):
        if getattr(self, "handles", None) is not None:
            self.L.mik_cgd_group_release(self.handles, self.P)
            self.handles = None


class PartitionLinks:
    """The two points where the ranks of a row-partitioned iterable couple (include/mik.h: mik_halo_fn,
    mik_reduce_fn), on top of a communicator.  ``send_buf`` / ``x_ext`` are 1-D tensors (device tensors for
    the product, CPU tensors for the gloo test double)."""

    def __init__(self, comm, plan: HaloPlan, send_buf, x_ext):
        self.comm, self.plan, self.send_buf, self.x_ext = comm, plan, send_buf, x_ext

    def halo(self):
        p = self.plan
        self.comm.exchange(p, self.send_buf, self.x_ext[p.n_loc:p.n_loc + p.n_ghost])

    def reduce(self, values: np.ndarray):
        """values: this rank's partial sums -> in place, the sums over ranks 0..P-1 in rank order."""
        values[:] = rank_ordered_sum(self.comm.all_gather_host(values))


class DistGMRESIterable:
    """``GMRESIterable`` (src/gmres.jl:31-49) over a row partition: this rank's block of the Arnoldi basis
    lives in ``mik_gmres_create_partitioned``; the halo exchange before every SpMV and the rank-ordered
    sums of the projections / norms (src/orthogonalize.jl:71,75; src/gmres.jl:252) come back here as
    callbacks and go through ``comm``.  ``iterate`` follows src/gmres.jl:57-106 on every rank identically."""

    def __init__(self, pkg, comm, ptr, local_idx, val, plan: HaloPlan, b_loc, x_loc=None, *, abstol=0.0, reltol=None, restart=20,
                 maxiter=None, orth_meth=None, pl_diag=None, pr_diag=None, device=0, n_global=None, native=None):
        """``native``: None -- the exchanges are host callbacks over ``comm`` (any communicator of this module); "mailbox" -- a ``mik_comm``
        without RCCL is created on this iterable's context, its mailboxes and a ``mik_plink`` for `plan` are connected over ``comm`` (which then
        only bootstraps, as in NativeComm), and the library couples the ranks on the device: no callback, no host round trip inside an
        Arnoldi column."""
        import torch
        self.pkg, self.comm, self.plan, self.torch = pkg, comm, plan, torch
        L = pkg.lib()
        self.L = L
        dtype = np.dtype(val.dtype)
        self.dtype = dtype
        tdt = {np.dtype(np.float64): torch.float64, np.dtype(np.float32): torch.float32}[dtype]
        dev = torch.device("cuda", device)
        self.ctx = pkg.HipContext(device)
        self.stream = torch.cuda.Stream(device=dev)
        self.ctx.set_stream(self.stream.cuda_stream)
        n_loc, n_ext = plan.n_loc, plan.n_loc + plan.n_ghost
        self.A = pkg.HipCSR(n_loc, n_ext, ptr, local_idx, val, index_base=0, is_csc=False, ctx=self.ctx)
        with torch.cuda.stream(self.stream):
            self.x_ext = torch.zeros(max(n_ext, 1), dtype=tdt, device=dev)
            self.send_buf = torch.zeros(max(plan.n_send, 1), dtype=tdt, device=dev)
            self.send_idx = (torch.from_numpy(plan.send_idx.astype(np.int32)).to(dev) if plan.n_send
                             else torch.zeros(1, dtype=torch.int32, device=dev))
        self.b = pkg.HipVector.from_numpy(np.ascontiguousarray(b_loc, dtype), self.ctx)
        initially_zero = x_loc is None
        self.x = pkg.HipVector(n_loc, dtype, self.ctx).fill_(0) if initially_zero else pkg.HipVector.from_numpy(np.ascontiguousarray(x_loc, dtype), self.ctx)
        self.pl = pkg.HipVector.from_numpy(np.ascontiguousarray(pl_diag, dtype), self.ctx) if pl_diag is not None else None
        self.pr = pkg.HipVector.from_numpy(np.ascontiguousarray(pr_diag, dtype), self.ctx) if pr_diag is not None else None
        n_glob = int(n_global) if n_global is not None else int(sum(comm.all_gather_objects(n_loc)))
        self.restart = int(min(20, n_glob) if restart is None else restart)                 # src/gmres.jl:113
        self.maxiter = int(n_glob if maxiter is None else maxiter)                           # :114
        reltol = float(np.sqrt(np.finfo(dtype).eps)) if reltol is None else float(reltol)    # :112
        self.orth_meth = orth_meth if orth_meth is not None else pkg.ModifiedGramSchmidt()   # :116
        self.callback_error = None
        ctype = C.c_double if dtype == np.float64 else C.c_float

        self.links = PartitionLinks(comm, plan, self.send_buf, self.x_ext)

        def _halo(_user):
            try:
                self.links.halo()
                return 0
            except BaseException as e:                      # never let an exception cross the C frame
                self.callback_error = e
                return 1

        def _reduce(_user, _dtype, count, values):
            try:
                self.links.reduce(np.ctypeslib.as_array(C.cast(values, C.POINTER(ctype)), shape=(count,)))
                return 0
            except BaseException as e:
                self.callback_error = e
                return 1

        self._halo_cb, self._reduce_cb = pkg._lib.HALO_FN(_halo), pkg._lib.REDUCE_FN(_reduce)   # keep alive
        self.ncomm, self.link = None, None
        if native is not None:
            self.ncomm = NativeComm(pkg, self.ctx, comm, transport=native)
            self.link = self.ncomm.make_link(plan, dtype)
        self.part = pkg._lib.MikPartition(plan.rank, plan.nranks, n_ext, self.x_ext.data_ptr(), self.send_idx.data_ptr(), plan.n_send,
                                          self.send_buf.data_ptr(), self._halo_cb, self._reduce_cb, None, self.link)
        h = _vp()
        with torch.cuda.stream(self.stream):
            self._check(L.mik_gmres_create_partitioned(
                self.ctx.handle, self.A.handle, _vp(self.x.ptr), _vp(self.b.ptr), _vp(self.pl.ptr if self.pl else None),
                _vp(self.pr.ptr if self.pr else None), float(abstol), reltol, self.restart, self.maxiter, int(initially_zero),
                self.orth_meth.code, C.byref(self.part), C.byref(h)), "mik_gmres_create_partitioned")
        self.handle = h
        self._refresh()

    def _check(self, status, where):
        if status and self.callback_error is not None:
            err, self.callback_error = self.callback_error, None
            raise err
        self.pkg._lib.check(status, where, self.ctx.handle)

    def _refresh(self):
        res, tol, beta = C.c_double(), C.c_double(), C.c_double()
        k, mv, conv = C.c_int(), C.c_int64(), C.c_int()
        self._check(self.L.mik_gmres_state(self.handle, C.byref(res), C.byref(tol), C.byref(beta), C.byref(k), C.byref(mv), C.byref(conv)),
                    "mik_gmres_state")
        self.residual_current, self.tol, self.beta, self.k, self.mv_products = res.value, tol.value, beta.value, k.value, mv.value

    def converged(self) -> bool:                                         # src/gmres.jl:51
        return self.residual_current <= self.tol

    def done(self, iteration: int) -> bool:                              # src/gmres.jl:55
        return iteration >= self.maxiter or self.converged()

    def iterate(self, iteration: int = 0):
        res, done = C.c_double(), C.c_int()
        with self.torch.cuda.stream(self.stream):
            self._check(self.L.mik_gmres_iterate(self.handle, int(iteration), C.byref(res), C.byref(done)), "mik_gmres_iterate")
        if done.value:
            return None
        self._refresh()
        return self.residual_current, iteration + 1

    def __iter__(self):
        iteration = 0
        while (nxt := self.iterate(iteration)) is not None:
            _, iteration = nxt
            yield self.residual_current

    def solve(self) -> np.ndarray:
        """The loop of gmres! (src/gmres.jl:207-214): residual history of this solve."""
        return np.asarray(list(self))

    def solution(self) -> np.ndarray:
        return self.x.to_numpy()

    def iterate_many(self, iteration: int, max_steps: int) -> np.ndarray:
        """Up to max_steps iterate() calls inside the library (mik_gmres_iterate_many): the loop of gmres! without a trip through the host
        language per inner iteration."""
        hist = np.empty(max(int(max_steps), 1), np.float64)
        steps = C.c_int64()
        with self.torch.cuda.stream(self.stream):
            self._check(self.L.mik_gmres_iterate_many(self.handle, int(iteration), int(max_steps), hist.ctypes.data_as(C.POINTER(C.c_double)), C.byref(steps)),
                        "mik_gmres_iterate_many")
        self._refresh()
        return hist[:steps.value].copy()

    def close(self):
        if getattr(self, "handle", None):
            self.L.mik_gmres_destroy(self.handle)
            self.handle = None
        if getattr(self, "link", None):
            self.L.mik_plink_destroy(self.link)
            self.link = None
        if getattr(self, "ncomm", None):
            self.ncomm.close()
            self.ncomm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LoopbackCG:
    """P virtual ranks in ONE process (all engines on one device / one stream), stepped in lockstep:
    the same phases and exchange plans as the multi-process path, with the collectives replaced by
    tensor copies.  This is how the partitioned path is verified on a single GPU."""

    def __init__(self, engines: Sequence, *, maxiter):
        self.engines, self.maxiter = list(engines), int(maxiter)
        self._all(INIT_A)
        self._exchange()
        self._all(INIT_B)
        self._gather("rr_all")
        self._all(INIT_C)
        waits = [e.wait() for e in self.engines]
        self.residual, self.tol = waits[0][0], waits[0][1]
        assert all(w[0] == self.residual and w[1] == self.tol for w in waits), "ranks disagree on the initial residual"

    def _all(self, ph, iteration=0):
        for e in self.engines:
            with e.stream_ctx():
                e.phase(ph, iteration)

    def _exchange(self):
        for p, e in enumerate(self.engines):
            with e.stream_ctx():
                for peer, off, cnt in e.plan.recv:
                    src = self.engines[peer]
                    soff = next(o for (q, o, c) in src.plan.send if q == p)
                    e.ghost_view()[off:off + cnt].copy_(src.send_buf[soff:soff + cnt])

    def _gather(self, name):
        for e in self.engines:
            with e.stream_ctx():
                for q, src in enumerate(self.engines):
                    if src is not e:
                        getattr(e, name)[q:q + 1].copy_(getattr(src, name)[q:q + 1])

    def done(self, iteration):
        return iteration >= self.maxiter or self.residual <= self.tol

    def iterate_many(self, iteration: int, max_steps: int) -> np.ndarray:
        if max_steps <= 0 or self.done(iteration):
            return np.zeros(0)
        max_steps = min(int(max_steps), self.maxiter - iteration, 1024)
        split = all(getattr(e, "overlap", False) for e in self.engines)
        for j in range(max_steps):
            self._all(STEP_A)
            if split:                                   # same phase order as DistCGIterable with an overlapping halo
                self._all(STEP_B_INTERIOR)
                self._exchange()
                self._all(STEP_B_REST)
            else:
                self._exchange()
                self._all(STEP_B)
            self._gather("dot_all")
            self._all(STEP_C)
            self._gather("rr_all")
            self._all(STEP_D, iteration + j)
        waits = [e.wait() for e in self.engines]
        hist = waits[0][3]
        assert all(np.array_equal(w[3], hist) for w in waits), "ranks disagree on the residual history"
        if hist.size:
            self.residual = waits[0][0]
        return hist

    def solve(self) -> np.ndarray:
        out, iteration = [], 0
        while True:
            h = self.iterate_many(iteration, 64)
            if h.size == 0:
                break
            out.append(h)
            iteration += h.size
        return np.concatenate(out) if out else np.zeros(0)

    def solution(self) -> np.ndarray:
        return np.concatenate([e.solution() for e in self.engines])


def build_rank_problem(pkg, comm, N: int, nz_per_rank: Optional[int] = None, dtype=np.float64, device=None, note=None):
    """z-slab of the 3D Laplacian on an N x N x (nz_per_rank * P) grid -- or of the cubic N^3 grid when
    nz_per_rank is None -- plus the hashed rhs; returns (ptr, local_idx, val, plan, b_loc, n_global)."""
    P, rank = comm.size, comm.rank
    rows = _laplace_rows
    localize = localize_block
    if device is not None:                          # generate and localise the slab on the GPU: no host pass over its entries
        rows = lambda pkg_, N_, NZ_, a, b, dt: _laplace_rows_torch(N_, NZ_, a, b, dt, device)
        localize = localize_block_torch
    if nz_per_rank is None:
        n = N ** 3
        offsets = partition_rows(n, P, align=N * N)
        n_glob, ptr, idx, val = rows(pkg, N, N, offsets[rank], offsets[rank + 1], dtype)
    else:
        NZ = nz_per_rank * P
        n = N * N * NZ
        offsets = np.arange(P + 1, dtype=np.int64) * (N * N * nz_per_rank)
        n_glob, ptr, idx, val = rows(pkg, N, NZ, offsets[rank], offsets[rank + 1], dtype)
    note = note or (lambda msg: None)
    note("slab rows generated")
    local_idx, plan = localize(ptr, idx, offsets, rank)
    note("columns localised")
    needs = comm.all_gather_objects(plan.ghost_gids)
    note("halo needs gathered")
    complete_plan(plan, offsets, needs)
    b_loc = pkg.fixtures.hashed_rhs(n, int(offsets[rank]), int(offsets[rank + 1]), dtype)
    note("halo plan complete, right-hand side generated")
    return ptr, local_idx, val, plan, b_loc, n, offsets


def _laplace_rows(pkg, N, NZ, r0, r1, dtype):
    """Rows r0:r1 of the 7-point Laplacian on an N x N x NZ grid (x fastest), CSR with global columns."""
    j = np.arange(r0, r1, dtype=np.int64)
    dims = [(1, N), (N, N), (N * N, NZ)]
    cand = []
    for stride, ext in reversed(dims):
        c = (j // stride) % ext
        cand.append((j - stride, c > 0, -1.0))
    cand.append((j, np.ones(j.shape, bool), 6.0))
    for stride, ext in dims:
        c = (j // stride) % ext
        cand.append((j + stride, c < ext - 1, -1.0))
    idx = np.stack([c[0] for c in cand], axis=1)
    mask = np.stack([c[1] for c in cand], axis=1)
    val = np.broadcast_to(np.asarray([c[2] for c in cand], dtype=dtype), idx.shape)
    ptr = np.zeros(j.size + 1, np.int64)
    np.cumsum(mask.sum(axis=1), out=ptr[1:])
    return N * N * NZ, ptr, idx[mask], np.ascontiguousarray(val[mask])


def build_self_halo_problem(pkg, N: int, NZ: int, device, dtype=np.float64):
    """ONE slab of an N x N x NZ grid that is periodic in z: the rank is its own lower and upper neighbour, so its halo
    (bottom and top planes, 2 N^2 entries) is exchanged with itself -- over RCCL when the native transport is used.  A
    single-GPU box then runs and times every call of the P-rank step (development / measurement aid: ``MIK_DIST_SELF_HALO=1``).
    Returns what build_rank_problem returns."""
    import torch
    dev = torch.device("cuda", device) if isinstance(device, int) else device
    tdt = {np.dtype(np.float64): torch.float64, np.dtype(np.float32): torch.float32}[np.dtype(dtype)]
    n, plane = N * N * NZ, N * N
    j = torch.arange(n, dtype=torch.int64, device=dev)
    big = torch.iinfo(torch.int64).max
    cols, vals = [], []
    for stride, ext, periodic in ((1, N, False), (N, N, False), (plane, NZ, True)):
        c = torch.div(j, stride, rounding_mode="floor") % ext
        lo = torch.where(c > 0, j - stride, j + stride * (ext - 1) if periodic else torch.full_like(j, big))
        hi = torch.where(c < ext - 1, j + stride, j - stride * (ext - 1) if periodic else torch.full_like(j, big))
        cols += [lo, hi]
        vals += [-1.0, -1.0]
    cols.append(j)
    vals.append(6.0)
    idx = torch.stack(cols, dim=1)
    val = torch.tensor(vals, dtype=tdt, device=dev).expand(idx.shape)
    idx, order = torch.sort(idx, dim=1)                       # a row's entries in ascending global column, absent ones last
    val = torch.gather(val, 1, order)
    mask = idx != big
    ptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(mask.sum(dim=1), dim=0, out=ptr[1:])
    rows = j.unsqueeze(1).expand(idx.shape)[mask]
    gi, gv = idx[mask].contiguous(), val[mask].contiguous()
    wrap = (gi - rows).abs() == plane * (NZ - 1)              # the periodic neighbours: served through the halo
    ghost = torch.unique(gi[wrap])
    li = torch.where(wrap, n + torch.searchsorted(ghost, gi), gi).contiguous()
    ghost_gids = ghost.cpu().numpy()
    plan = HaloPlan(0, 1, n, ghost_gids)
    plan.recv = [(0, 0, int(ghost_gids.size))]
    plan.send = [(0, 0, int(ghost_gids.size))]
    plan.send_idx = ghost_gids.astype(np.int32)
    b_loc = pkg.fixtures.hashed_rhs(n, 0, n, dtype)
    return ptr, li, gv, plan, b_loc, n, np.array([0, n], np.int64)


def _laplace_rows_torch(N, NZ, r0, r1, dtype, device):
    """_laplace_rows on the GPU (torch tensors): the rank's slab never exists on the host."""
    import torch
    dev = torch.device("cuda", device) if isinstance(device, int) else device
    tdt = {np.dtype(np.float64): torch.float64, np.dtype(np.float32): torch.float32}[np.dtype(dtype)]
    j = torch.arange(int(r0), int(r1), dtype=torch.int64, device=dev)
    dims = [(1, N), (N, N), (N * N, NZ)]
    cand = []
    for stride, ext in reversed(dims):
        c = torch.div(j, stride, rounding_mode="floor") % ext
        cand.append((j - stride, c > 0, -1.0))
    cand.append((j, torch.ones_like(j, dtype=torch.bool), 6.0))
    for stride, ext in dims:
        c = torch.div(j, stride, rounding_mode="floor") % ext
        cand.append((j + stride, c < ext - 1, -1.0))
    idx = torch.stack([c[0] for c in cand], dim=1)
    mask = torch.stack([c[1] for c in cand], dim=1)
    val = torch.tensor([c[2] for c in cand], dtype=tdt, device=dev).expand(idx.shape)
    ptr = torch.zeros(j.numel() + 1, dtype=torch.int64, device=dev)
    torch.cumsum(mask.sum(dim=1), dim=0, out=ptr[1:])
    return N * N * NZ, ptr, idx[mask].contiguous(), val[mask].contiguous()


def localize_block_torch(ptr, idx, offsets, rank):
    """localize_block for device tensors: local column ids stay on the device, the (small) halo plan comes to the host."""
    import torch
    r0, r1 = int(offsets[rank]), int(offsets[rank + 1])
    n_loc = r1 - r0
    owned = (idx >= r0) & (idx < r1)
    ghost = torch.unique(idx[~owned])                                   # sorted
    local = torch.where(owned, idx - r0, n_loc + torch.searchsorted(ghost, idx))
    ghost_gids = ghost.cpu().numpy()
    plan = HaloPlan(rank, len(offsets) - 1, n_loc, ghost_gids)
    owner = np.searchsorted(offsets, ghost_gids, side="right") - 1
    for peer in np.unique(owner):
        sel = np.nonzero(owner == peer)[0]
        plan.recv.append((int(peer), int(sel[0]), int(sel.size)))
    return local.contiguous(), plan
