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



# ==============================================================================================
# first contact with the machine: transport self-test (child processes) and the in-process group
# ==============================================================================================
def transport_selftest(boot, rank, world, device, wanted, *, timeout=None, simulate_failure=()):
    """Run iterativesolvers.jl_amd/selftest.py for every transport in `wanted` ("mailbox", "rccl") as a CHILD process of every rank,
    before anything is timed: sequence-numbered scalars through the mailbox slots, 4 MB payloads through the landing buffers,
    ncclAllGather of one double and 4 MB ncclSend / ncclRecv, every word checked, every check timed.  A child that does not return in
    `timeout` seconds is killed -- the parent never touches a transport whose self-test failed.  Collective over `boot` (which only
    carries the verdicts).  Returns {"mailbox": {...}, "rccl": {...}, "usable": [...]} -- identical on every rank."""
    import shutil
    import subprocess
    import tempfile
    timeout = float(os.environ.get("MIK_SELFTEST_TIMEOUT_S", "75")) if timeout is None else float(timeout)
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "selftest.py")
    base = boot.all_gather_objects(tempfile.mkdtemp(prefix="mik_selftest_") if rank == 0 else None)[0]
    devices = boot.all_gather_objects((os.uname().nodename, int(device)))
    shared = len(set(devices)) < len(devices)
    report = {"what": "child processes (one per rank and transport) before any timed leg: iterativesolvers.jl_amd/selftest.py", "world": world,
              "ranks_share_a_device": bool(shared), "timeout_seconds": timeout}
    for name in wanted:
        t0 = time.perf_counter()
        if name in simulate_failure:
            mine = {"pass": False, "failure": f"failure simulated by MIK_SELFTEST_FAIL={name} (development)"}
        elif name == "rccl" and shared and world > 1:
            mine = {"pass": False, "skipped": True, "failure": "RCCL needs distinct devices: two ranks of this run share one GPU (ncclCommInitRank refuses duplicate devices)"}
        elif name == "rccl" and world == 1:
            mine = {"pass": False, "skipped": True, "failure": "a world of one has nothing to exchange over RCCL"}
        else:
            cmd = [sys.executable, script, "--transport", name, "--rank", str(rank), "--world", str(world), "--device", str(device),
                   "--dir", os.path.join(base, name), "--timeout", str(max(10.0, timeout - 10.0))]
            env = dict(os.environ)
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):       # the child meets its peers through files, not through the launcher's store
                env.pop(k, None)
            try:
                proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True)
                try:
                    so, se = proc.communicate(timeout=timeout)
                    lines = [ln for ln in so.strip().splitlines() if ln.startswith("{")]
                    mine = json.loads(lines[-1]) if lines else {"pass": False, "failure": f"no result line (exit code {proc.returncode}): {se.strip()[-300:]}"}
                except subprocess.TimeoutExpired:
                    proc.kill()
                    proc.communicate()
                    mine = {"pass": False, "failure": f"the self-test did not return within {timeout:.0f} s and was killed"}
            except Exception as exc:       # noqa: BLE001 -- cannot spawn: the transport is not usable from this container
                mine = {"pass": False, "failure": f"could not start the self-test child: {type(exc).__name__}: {exc}"[:300]}
        mine["wall_seconds"] = time.perf_counter() - t0
        every = boot.all_gather_objects(mine)
        rec = {"pass": all(bool(e.get("pass")) for e in every), "ranks": every}
        if not rec["pass"]:
            rec["failure"] = next((f"rank {q}: {e.get('failure') or 'a check failed'}" for q, e in enumerate(every) if not e.get("pass")), None)
            rec["skipped"] = all(bool(e.get("skipped")) for e in every if not e.get("pass"))
        else:
            def med(check, key):
                vals = [e["checks"][check][key] for e in every if check in e.get("checks", {}) and key in e["checks"][check]]
                return float(np.median(vals)) if vals else None
            rec["summary"] = ({"mailbox_scalars_us": med("mailbox_scalars", "us_per_round_median"), "landing_4MB_us": med("landing_4MB", "us_per_exchange_median"),
                               "landing_4MB_gbs_received": med("landing_4MB", "gbs_received")} if name == "mailbox" else
                              {"rccl_allgather_us": med("rccl_allgather", "us_per_round_median"), "rccl_halo_4MB_us": med("rccl_halo_4MB", "us_per_exchange_median"),
                               "rccl_halo_4MB_gbs_received": med("rccl_halo_4MB", "gbs_received")})
        report[name] = rec
    boot.barrier()
    if rank == 0:
        shutil.rmtree(base, ignore_errors=True)
    ok = {n for n in wanted if report.get(n, {}).get("pass")}
    report["usable"] = [t for t, needs in (("mailbox", {"mailbox"}), ("rccl+mailbox", {"rccl", "mailbox"}), ("rccl", {"rccl"})) if needs <= ok]
    return report


class _OneOf:
    """rank `rank` of a world of `size` whose plans are completed by the caller (build_group_problem)"""

    def __init__(self, rank, size):
        self.rank, self.size = rank, size

    def all_gather_objects(self, obj):
        self.mine = obj
        return [np.zeros(0, np.int64)] * self.size          # completed later, once every rank's needs are known


def build_group_problem(pkg, N: int, nz_per_rank: int, P: int, devices, dtype=np.float64):
    """build_rank_problem for all P ranks in ONE process (the in-process group, include/mik.h "Transport 2"): rank p's slab is generated
    on devices[p].  Returns (list of per-rank tuples as build_rank_problem returns them)."""
    probs, needs = [], []
    for p in range(P):
        fake = _OneOf(p, P)
        probs.append(build_rank_problem(pkg, fake, N, nz_per_rank=nz_per_rank, dtype=dtype, device=devices[p]))
        needs.append(fake.mine)
    for p in range(P):
        complete_plan(probs[p][3], probs[p][6], needs)
    return probs

# ==============================================================================================
# bench entry (bench.py --gpus N: one rank per GPU, started by torch.distributed.run or by bench.py itself)
# ==============================================================================================
def bench_main(args):
    """BASELINE.json configs[3]: cg! on the z-slab partition of the 512 x 512 x 64 P Laplacian (P = 8: the 512^3 grid;
    64 planes and two 512^2-double halos per rank), exchanges over RCCL issued from inside libmik.so.  P = 1
    (--force-dist) runs the same code path on the 256^3 grid of configs[1]."""
    import math
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.load_package()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if "MIK_FORCE_DEVICE" in os.environ:          # development: several ranks on one GPU (if the backend allows it)
        local_rank = int(os.environ["MIK_FORCE_DEVICE"])
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} needs device {local_rank}, {torch.cuda.device_count()} device(s) visible")
    torch.cuda.set_device(local_rank)
    transport = os.environ.get("MIK_DIST_TRANSPORT", "native")   # "native": RCCL inside libmik.so; "torch": phases driven from Python
    t_bench0 = time.perf_counter()
    group_only, boot_failure = False, None

    def note(msg):
        """progress on stderr with the time since start (rank 0): where a first run on new hardware spends its time, or stops, is visible in the log"""
        if rank == 0:
            print(f"bench.py [{time.perf_counter() - t_bench0:7.1f} s] {msg}", file=sys.stderr, flush=True)
    if world > 1 or "RANK" in os.environ:
        # the process group only bootstraps (ncclUniqueId, barriers, max over ranks of the timings): gloo suffices for the
        # native transport; the legacy transport needs torch's own RCCL communicator
        try:
            import datetime
            if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost", "::1"):
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # one node: no hostname lookup (a container's hostname may not resolve, or resolve slowly)
            if os.environ.get("MIK_BOOT_FAIL") == "1":
                raise RuntimeError("bootstrap failure simulated by MIK_BOOT_FAIL=1 (development)")
            if transport == "torch" and os.environ.get("MIK_DIST_BACKEND", "nccl") == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(os.environ.get("MIK_DIST_BACKEND", "gloo") if transport == "torch" else "gloo", rank=rank, world_size=world,
                                        timeout=datetime.timedelta(seconds=float(os.environ.get("MIK_BOOT_TIMEOUT_S", "180"))))
            boot = TorchComm()
            assert dist.get_world_size() == world
            note("process group (gloo) up")
        except Exception as exc:       # noqa: BLE001
            # The ranks cannot even meet (rendezvous refused, store unreachable): rank 0 measures the partitioned system alone through the
            # in-process group (include/mik.h "Transport 2": one host thread, every rank's slab on its own device, peer copies) -- the
            # driver still gets a contract-complete line; the other ranks leave quietly.
            boot_failure = f"{type(exc).__name__}: {exc}"[:300]
            print(f"bench.py: rank {rank}: process-group bootstrap failed ({boot_failure})", file=sys.stderr)
            if rank != 0:
                sys.exit(0)
            group_only = True
            boot = SelfComm()
    else:
        boot = SelfComm()
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but {world} rank(s) are running")
    K, Wm = args.steps, args.warmup
    if args.n is not None:
        N, nz = args.n, max(1, args.n // world)              # --grid G: the G^3 cube cut into `world` slabs
    elif world == 1:
        N, nz = 256, 256                                     # configs[1] through the partitioned code path
    else:
        N, nz = 512, 64                                      # configs[3]: 512 x 512 x 64 P (P = 8: 512^3)
    if "MIK_DIST_NZ" in os.environ:                      # development: planes per rank (e.g. --grid 512 with 64 planes = one rank's slab of configs[3])
        nz = int(os.environ["MIK_DIST_NZ"])
    t_up = time.perf_counter()
    # The slab is generated on the HOST (numpy, ~1.5 s for 16.7 M rows) and uploaded like any SparseMatrixCSC.  MIK_DIST_HOST_BUILD=0 generates it with
    # PyTorch on the device instead -- measured in round 6 with 3 / 4 processes on one GPU: 1 s on a fresh box, then 30 ... 500 s on the same box in
    # later runs (inside torch's indexing / scan ops; libmik's own allocations and kernels stayed at their usual times), so it is not the default.
    on_host = os.environ.get("MIK_DIST_HOST_BUILD", "1") == "1"
    self_halo = world == 1 and os.environ.get("MIK_DIST_SELF_HALO", "0") == "1"      # z-periodic slab: the rank exchanges its halo with itself
    torch.cuda.synchronize()
    note("HIP runtime up on the rank's device")
    group_devices = [int(os.environ["MIK_FORCE_DEVICE"])] * world if "MIK_FORCE_DEVICE" in os.environ else list(range(world))
    group_probs = None
    if group_only:
        if max(group_devices) >= torch.cuda.device_count():
            raise SystemExit(f"bench.py: the in-process group needs devices {group_devices}, {torch.cuda.device_count()} visible")
        group_probs = build_group_problem(pkg, N, nz, world, [None] * world if on_host else group_devices)
        ptr, local_idx, val, plan, b_loc, n, offsets = group_probs[0]
    elif self_halo:
        ptr, local_idx, val, plan, b_loc, n, offsets = build_self_halo_problem(pkg, N, nz, local_rank)
    else:
        ptr, local_idx, val, plan, b_loc, n, offsets = build_rank_problem(pkg, boot, N, nz_per_rank=nz, device=None if on_host else local_rank, note=note)
    nnz_loc = int(val.numel() if hasattr(val, "numel") else val.size)
    note(f"{world} rank(s) met, slabs of {N}x{N}x{nz} generated ({plan.n_loc} rows, {nnz_loc} entries, {plan.n_ghost} halo entries on rank 0)")
    ptr_keep = True
    state = {"k": 0, "it": None}

    def run_steps(count, batch, keep=None):
        done = 0
        while done < count:
            h = state["it"].iterate_many(state["k"], min(batch, count - done))
            assert h.size > 0
            if keep is not None:
                keep.extend(h.tolist())
            done += h.size
            state["k"] += h.size

    def sync_devices():
        if state.get("solo"):                       # the in-process group: rank 0 drives every device
            for dv in sorted(set(group_devices)):
                torch.cuda.synchronize(dv)
        else:
            torch.cuda.synchronize()

    def region(count, batch):
        if not state.get("solo"):
            boot.barrier()
        sync_devices()
        t0 = time.perf_counter()
        run_steps(count, batch)
        sync_devices()
        if not state.get("solo"):
            boot.barrier()
        return time.perf_counter() - t0

    def max_over_ranks(values):
        if world == 1 or state.get("solo"):
            return list(values)
        t = torch.tensor(list(values), dtype=torch.float64)
        if dist.get_backend() != "gloo":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.cpu().tolist()

    def timed(batch, count):
        """regions of exactly `count` steps until >= 0.25 s have been measured; every rank runs the same number"""
        first = max_over_ranks([region(count, batch)])[0]
        more = max(0, min(199, math.ceil(0.25 / max(first, 1e-6)) - 1))
        times = [first] + max_over_ranks([region(count, batch) for _ in range(more)])
        return times

    HBM_PEAK = 8000.0
    sqrt_eps = float(np.sqrt(np.finfo(np.float64).eps))
    big = dict(ptr=ptr, local_idx=local_idx, val=val, plan=plan, b_loc=b_loc)

    def bring_up(name, layout, prob, reltol, maxiter):
        """One transport on an engine of its own over `prob` (operator layout "auto" = mik_csr_create's choice, "csr" = the plain Int32 CSR
        arrays: the contract loop).  Collective: returns (engine, comm, iterable, None) or (None, None, None, first failure of any rank)."""
        e2 = c2 = i2 = None
        failure = None
        try:
            e2 = HipEngine(pkg, prob["ptr"], prob["local_idx"], prob["val"], prob["plan"], prob["b_loc"], abstol=0.0, reltol=reltol, maxiter=maxiter,
                           device=local_rank, layout=layout)
            c2 = NativeComm(pkg, e2.ctx, boot, force_rccl=force and name != "mailbox", transport=name)
            i2 = NativeDistCGIterable(pkg, e2, c2, maxiter=maxiter)
        except Exception as exc:       # noqa: BLE001
            failure = f"{type(exc).__name__}: {exc}"
        failures = [f for f in boot.all_gather_objects(failure) if f]
        if failures:
            for o in (e2, c2):
                try:
                    o and o.close()
                except Exception:      # noqa: BLE001
                    pass
            return None, None, None, failures[0][:300]
        return e2, c2, i2, None

    def tear_down(e2, c2):
        boot.barrier()
        for o in (e2, c2):
            try:
                o and o.close()
            except Exception:          # noqa: BLE001
                pass

    # The transports inside libmik.so (include/mik.h "Transport 1" / "Transport 3"), each on an engine of its own over the same slab:
    #   rccl          halo by ncclSend / ncclRecv on the side stream, the two scalars of a step by ncclAllGather
    #   rccl+mailbox  halo by RCCL, scalars as stores into peer-mapped mailboxes (no collective launch on the compute stream)
    #   mailbox       no RCCL at all: scalars by mailbox, halo pushed into the neighbours' IPC-mapped landing buffers
    # (2) every transport that came up runs the warm-up and the timed regions in the operator's default layout; their first residuals must
    #     agree bit for bit; the fastest of the largest agreeing group is `transport_chosen`.
    # (3) the CONTRACT loop: the chosen transport on the plain CSR arrays of the slab (mik_csr_set_layout(A_loc, 0), k_spmv_rowgather) --
    #     `value`, `ms_per_step` and `roofline` describe this loop, exactly as at N = 1.
    # (1) parity, last (a hang in it cannot cost the timed line): every transport solves a SMALL global system (64 x 64 x 8 P) to the default
    #     tolerance in both operator layouts and rank 0 compares history and solution with the partition-aware oracle (bench.py hands the
    #     checker in; this module never imports oracle/).
    # MIK_NATIVE_TRANSPORTS narrows / reorders the list.
    transports = {}
    chosen = None
    eng = it = ncomm = None
    import threading
    watchdog = {"timer": None}

    def emergency_line():
        """Something measured AFTER a good transport hangs (no device-to-device transfer of any kind could be tried before the driver's own
        multi-GPU run): every rank leaves, rank 0 first prints the line of what has been measured so far -- the complete line without
        parity_vs_oracle if the hang is in the parity leg, the default-layout line of the best transport if it is in a later transport or in the
        contract loop."""
        note = "a leg that ran after this measurement did not return in time; the process left with the line it had"
        if rank == 0 and state.get("line_ready"):
            line = make_line(note)
            if line.get("parity_vs_oracle") is None:
                line["parity_vs_oracle"] = {"reached": False, "note": "the parity leg runs last and did not finish"}
            print(json.dumps(line), flush=True)
        elif rank == 0 and chosen is not None:
            ms = transports[chosen]["ms_per_step"]
            print(json.dumps({
                "metric": "cg_iters_per_sec", "value": 1e3 / ms, "unit": "iters/s", "n_gpus": world, "world_size_checked": world, "steps": K, "warmup": Wm,
                "value_is_contract": False, "aggregate_slab_iters_per_sec": world * 1e3 / ms, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"cg! on the {N}x{N}x{nz * world} 3D 7-point Laplacian row-partitioned into {world} z-slab(s) of {N}x{N}x{nz} rows",
                           "n": int(n), "n_per_gpu": plan.n_loc, "host_sync_per_step": 1, "transport_chosen": chosen, "transports_measured": transports,
                           "operator_layout_of_the_timed_loop": "default (slice-constant); the CSR contract loop was not reached", "watchdog": note},
                "roofline": None}), flush=True)
        os._exit(0 if chosen is not None else 3)

    def arm_watchdog():
        if watchdog["timer"] is not None:
            watchdog["timer"].cancel()
        watchdog["timer"] = None
        if world > 1 and chosen is not None:
            watchdog["timer"] = threading.Timer(float(os.environ.get("MIK_BENCH_WATCHDOG_S", "150")), emergency_line)
            watchdog["timer"].daemon = True
            watchdog["timer"].start()

    parity = None
    contract = None
    names, force = [], False
    def run_parity():
        """every transport solves a SMALL global system (64 x 64 x 8 P) to the default tolerance in both operator layouts; rank 0 compares history
        and solution with the partition-aware oracle (bench.py hands the checker in; this module never imports oracle/)"""
        nonlocal parity
        check = getattr(args, "partition_oracle_fn", None)
        if check is not None and not self_halo and not getattr(args, "no_parity", False):
            Ns, nzs = 64, 8
            sp, sli, sv, splan, sb, sn, soff = build_rank_problem(pkg, boot, Ns, nz_per_rank=nzs, device=None if on_host else local_rank)
            small = dict(ptr=sp, local_idx=sli, val=sv, plan=splan, b_loc=sb)
            parity = {"workload": f"cg! to reltol = sqrt(eps) on the {Ns}x{Ns}x{nzs * world} Laplacian, {world} z-slab(s) of {nzs} planes, hashed rhs, x0 = 0",
                      "oracle": "oracle/mik_oracle.c cg, TREE mode with the same row partition (rank-ordered sums of the per-rank trees)", "transports": {}}
            for name in names:
                for layout in ("auto", "csr"):
                    key = f"{name}/{layout}"
                    note(f"parity: {key} on the {Ns}x{Ns}x{nzs * world} system")
                    e2, c2, i2, failure = bring_up(name, layout, small, sqrt_eps, 10 ** 6)
                    if failure:
                        parity["transports"][key] = {"came_up": False, "failure": failure}
                        continue
                    hist, k2, failure = [], 0, None
                    try:
                        while True:
                            h = i2.iterate_many(k2, 1 if k2 < 2 else 25)       # single steps, then batches: both host protocols
                            if h.size == 0:
                                break
                            hist.extend(h.tolist())
                            k2 += h.size
                        xs = e2.solution()
                    except Exception as exc:      # noqa: BLE001
                        failure, xs = f"{type(exc).__name__}: {exc}", None
                    shape = e2.ctx.cg_shape(np.float64)
                    gathered = boot.all_gather_objects((failure, [float(v).hex() for v in hist], xs))
                    tear_down(e2, c2)
                    if any(g[0] for g in gathered):
                        parity["transports"][key] = {"came_up": True, "failure": next(g[0] for g in gathered if g[0])[:300]}
                        continue
                    rec = {"came_up": True, "iters": len(hist), "ranks_agree": all(g[1] == gathered[0][1] for g in gathered)}
                    if rank == 0:
                        ref = check(Ns, nzs * world, soff, shape)
                        rec.update(oracle_iters=int(ref["iters"]), same_iters_isconverged=bool(len(hist) == ref["iters"] and ref["isconverged"]),
                                   history_bit_identical=bool(np.array_equal(np.asarray(hist), ref["resnorm"])),
                                   solution_bit_identical=bool(np.array_equal(np.concatenate([g[2] for g in gathered]), ref["x"])))
                        rec["bit_identical"] = bool(rec["ranks_agree"] and rec["history_bit_identical"] and rec["solution_bit_identical"] and rec["same_iters_isconverged"])
                    parity["transports"][key] = rec
            if rank == 0:
                ok = [k2 for k2, v in parity["transports"].items() if v.get("bit_identical")]
                parity["bit_identical"] = bool(ok) and all(v.get("bit_identical") for v in parity["transports"].values() if v.get("came_up"))
                parity["transports_bit_identical"] = ok


    selftest = None
    if transport == "native":
        # (order: the transport whose waits are all bounded first -- once it has been measured, a hang of a later one is survivable)
        default = "mailbox,rccl+mailbox,rccl" if (world > 1 or self_halo) else "rccl"
        names = [t for t in os.environ.get("MIK_NATIVE_TRANSPORTS", default).split(",") if t and t != "group"]
        force = self_halo or os.environ.get("MIK_DIST_FORCE_COLLECTIVES", "0") == "1"
        if group_only:
            selftest = {"reached": False, "failure": f"the ranks could not meet: {boot_failure}", "usable": []}
            names = []
        elif world > 1 and os.environ.get("MIK_SELFTEST", "1") != "0":
            # ---- (0) FIRST CONTACT: every transport proves itself in child processes before anything of this process touches it ------------
            wanted = [t for t in ("mailbox", "rccl") if any(t in nm.split("+") for nm in names)]
            selftest = transport_selftest(boot, rank, world, local_rank, wanted, simulate_failure=[f for f in os.environ.get("MIK_SELFTEST_FAIL", "").split(",") if f])
            selftest["candidates"] = list(names)
            selftest["dropped_from_candidates"] = [nm for nm in names if nm not in selftest["usable"]]
            names = [nm for nm in names if nm in selftest["usable"]]
            if rank == 0:      # on stderr at once: visible even if a later leg takes the process down
                brief = {k: ({"pass": v.get("pass"), **({"summary": v["summary"]} if "summary" in v else {"failure": v.get("failure")})} if isinstance(v, dict) and "pass" in v else v)
                         for k, v in selftest.items() if k != "what"}
                print("bench.py: transport_selftest " + json.dumps(brief), file=sys.stderr, flush=True)
        if world == 1:
            pkg.lib().mik_set_tuning(6, int(os.environ.get("MIK_KNOB6", "4")))     # a world of one still sends its scalars through the mailbox (development)

        # ---- (2) every transport in the operator's default layout -----------------------------------------------------------------
        alive = {}
        for name in names:
            arm_watchdog()                      # (only once a transport has been measured: then a hang of the next one is survivable)
            t_up = time.perf_counter()
            note(f"transport {name}: bring-up (default layout)")
            e2, c2, i2, failure = bring_up(name, "auto", big, 0.0, 10 ** 9)
            rec = {"came_up": failure is None}
            if failure:
                rec["failure"] = failure
                transports[name] = rec
                continue
            rec["operator_build_and_upload_seconds"] = time.perf_counter() - t_up
            state.update(k=0, it=i2)
            first = []
            try:
                run_steps(max(Wm, 8), 1, keep=first)
                tms = timed(1, K)
                failure = None
            except Exception as exc:       # noqa: BLE001
                failure = f"{type(exc).__name__}: {exc}"
            failures = [f for f in boot.all_gather_objects(failure) if f]
            if failures:
                rec.update(came_up=False, failure=failures[0][:300])
                transports[name] = rec
                continue
            note(f"transport {name}: {float(np.median(tms)) / K * 1e3:.4f} ms per step over {len(tms)} timed region(s) of {K} steps")
            rec.update(operator_layout=e2.A.layout(), ms_per_step=float(np.median(tms)) / K * 1e3, iters_per_sec=K / float(np.median(tms)), timed_regions=len(tms),
                       first_residuals=[float(v).hex() for v in first[:8]], uses_rccl=c2.uses_rccl())
            transports[name] = rec
            alive[name] = (e2, c2, i2, tms, state["k"])
            # provisional choice (what the watchdog would print): the fastest of the LARGEST group of transports with identical bits
            groups = {}
            for nm in alive:
                groups.setdefault(tuple(transports[nm]["first_residuals"]), []).append(nm)
            best = max(groups.values(), key=lambda g2: (len(g2), "rccl" in g2))
            for nm in alive:
                transports[nm]["same_bits_as_the_majority"] = nm in best
            chosen = min(best, key=lambda nm: transports[nm]["ms_per_step"])
        for nm, (e2, c2, i2, tms, kk) in alive.items():
            if nm == chosen:
                eng, ncomm, it, chosen_times, chosen_k = e2, c2, i2, tms, kk
            else:
                e2.close()
                c2.close()
        if chosen is None:
            if watchdog["timer"] is not None:
                watchdog["timer"].cancel()
            if rank == 0:
                print(f"bench.py: no transport between processes is usable ({ {k: v.get('failure') for k, v in transports.items()} }); "
                      f"measuring through the in-process group (one host thread, every slab on its own device, peer copies)", file=sys.stderr)
            transport = "group"
    del ptr_keep
    group = None
    if transport == "group":
        # ---- last resort (include/mik.h "Transport 2"): rank 0 drives all `world` slabs itself; needs no IPC handle, no RCCL, no second process ----
        chosen = "group"
        if rank != 0:                                         # rank 0 goes on alone; leaving with status 0 is not a failure for the launcher
            if dist.is_initialized():
                dist.destroy_process_group()
            return
        state["solo"] = True
        note(f"in-process group: {world} slabs on devices {group_devices}")
        if max(group_devices) >= torch.cuda.device_count():
            raise SystemExit(f"bench.py: the in-process group needs devices {group_devices}, {torch.cuda.device_count()} visible")
        del ptr, local_idx, val
        big = None
        if group_probs is None:
            group_probs = build_group_problem(pkg, N, nz, world, [None] * world if on_host else group_devices)
        plan = group_probs[0][3]

        def group_up(probs, layout, reltol, maxiter):
            engs = [HipEngine(pkg, q[0], q[1], q[2], q[3], q[4], abstol=0.0, reltol=reltol, maxiter=maxiter, device=group_devices[p2], layout=layout)
                    for p2, q in enumerate(probs)]
            return engs, GroupCG(pkg, engs, maxiter=maxiter)

        def group_down(engs, grp):
            grp.close()
            for e2 in engs:
                e2.close()
        t_up = time.perf_counter()
        engs, it = group_up(group_probs, "auto", 0.0, 10 ** 9)
        upload_seconds = time.perf_counter() - t_up
        eng = engs[0]
        uses_rccl = False
        state.update(k=0, it=it)
        first = []
        run_steps(max(Wm, 8), 1, keep=first)
        times = timed(1, K)
        transports["group"] = {"came_up": True, "operator_layout": eng.A.layout(), "ms_per_step": float(np.median(times)) / K * 1e3, "iters_per_sec": K / float(np.median(times)),
                               "timed_regions": len(times), "first_residuals": [float(v).hex() for v in first[:8]], "uses_rccl": False,
                               "devices": group_devices, "operator_build_and_upload_seconds": upload_seconds}
        group = {"engs": engs, "up": group_up, "down": group_down, "first": transports["group"]["first_residuals"]}
    if transport == "group":
        pass
    elif transport == "native":
        state.update(k=chosen_k, it=it)
        uses_rccl = ncomm.uses_rccl()
        upload_seconds = transports[chosen]["operator_build_and_upload_seconds"]
        times = chosen_times
    else:
        t_up = time.perf_counter()
        eng = HipEngine(pkg, ptr, local_idx, val, plan, b_loc, abstol=0.0, reltol=0.0, maxiter=10 ** 9, device=local_rank)
        upload_seconds = time.perf_counter() - t_up
        it = DistCGIterable(eng, boot, maxiter=10 ** 9)
        uses_rccl = world > 1
        state.update(k=0, it=it)
        run_steps(Wm, 1)
        times = timed(1, K)                  # one host-visible residual per step: the reference's protocol, as at N = 1
    kb = max(1, K // 25) * 25
    times_b = timed(25, kb)                  # one host wait per 25 steps
    dt = float(np.median(times))
    default_layout = eng.A.layout()
    default_kernel = eng.A.spmv_kernel()
    stored_bytes = eng.A.spmv_stored_bytes()
    alg_bytes = eng.A.spmv_algorithmic_bytes()          # SURVEY.md 8d on the rank's n_loc x n_ext block: nnz (s + 4) + (n_loc + 1) 4 + n_ext s + n_loc s
    u = pkg.HipVector.wrap(eng.u_ext.data_ptr(), plan.n_loc + plan.n_ghost, np.float64, eng.ctx, owner=eng.u_ext)
    d_b2b_ms = eng.A.time_spmv(u, eng.c, reps=20, fused_dot=True)
    default_first = transports[chosen]["first_residuals"] if transport == "native" else None

    # ---- (3) the contract loop: the chosen transport on the plain CSR arrays ---------------------------------------------------
    if transport == "group" and default_layout != "csr-rowblock" and not getattr(args, "no_csr", False):
        group["down"](group["engs"], it)
        e3s, i3 = group["up"](group_probs, "csr", 0.0, 10 ** 9)
        e3 = e3s[0]
        state.update(k=0, it=i3)
        first = []
        run_steps(max(Wm, 8), 1, keep=first)
        ms0, cnt0 = C.c_double(), C.c_int64()
        for q in e3s:
            pkg._lib.check(q.L.mik_cgd_profile(q.handle, 1, None, None), "mik_cgd_profile", q.ctx.handle)
        tms = timed(1, K)
        steps_timed = state["k"] - max(Wm, 8)
        per = []
        for q in e3s:
            pkg._lib.check(q.L.mik_cgd_profile(q.handle, 0, C.byref(ms0), C.byref(cnt0)), "mik_cgd_profile", q.ctx.handle)
            per.append((ms0.value / max(steps_timed, 1), int(cnt0.value)))
        tb3 = timed(25, kb)
        dt3 = float(np.median(tms))
        u3 = pkg.HipVector.wrap(e3.u_ext.data_ptr(), plan.n_loc + plan.n_ghost, np.float64, e3.ctx, owner=e3.u_ext)
        b2b = e3.A.time_spmv(u3, e3.c, reps=20, fused_dot=True)
        contract = {"came_up": True, "kernel": e3.A.spmv_kernel(), "operator_layout": e3.A.layout(), "iters_per_sec": K / dt3, "ms_per_step": dt3 / K * 1e3,
                    "spmv_in_loop_ms": per[0][0], "spmv_launches_timed": per[0][1], "steps_timed": int(steps_timed),
                    "spmv_in_loop_ms_per_rank": [q[0] for q in per], "spmv_back_to_back_ms": b2b,
                    "batched_25_steps_per_sync_iters_per_sec": kb / float(np.median(tb3)), "timed_regions": len(tms), "final_residual": i3.residual,
                    "first_residuals_equal_the_default_layout_bit_for_bit": bool([float(v).hex() for v in first[:8]] == group["first"])}
        group["down"](e3s, i3)
        # parity of the group against the partition-aware oracle on the small global system, both layouts
        check = getattr(args, "partition_oracle_fn", None)
        if check is not None and not getattr(args, "no_parity", False):
            Ns, nzs = 64, 8
            small = build_group_problem(pkg, Ns, nzs, world, [None] * world if on_host else group_devices)
            parity = {"workload": f"cg! to reltol = sqrt(eps) on the {Ns}x{Ns}x{nzs * world} Laplacian, {world} z-slab(s) of {nzs} planes, hashed rhs, x0 = 0",
                      "oracle": "oracle/mik_oracle.c cg, TREE mode with the same row partition (rank-ordered sums of the per-rank trees)", "transports": {}}
            for layout in ("auto", "csr"):
                es, g2 = group["up"](small, layout, sqrt_eps, 10 ** 6)
                hist, k2 = [], 0
                while True:
                    h = g2.iterate_many(k2, 1 if k2 < 2 else 25)
                    if h.size == 0:
                        break
                    hist.extend(h.tolist())
                    k2 += h.size
                xs = g2.solution()
                ref = check(Ns, nzs * world, small[0][6], es[0].ctx.cg_shape(np.float64))
                rec = {"came_up": True, "iters": len(hist), "ranks_agree": True, "oracle_iters": int(ref["iters"]),
                       "same_iters_isconverged": bool(len(hist) == ref["iters"] and ref["isconverged"]),
                       "history_bit_identical": bool(np.array_equal(np.asarray(hist), ref["resnorm"])), "solution_bit_identical": bool(np.array_equal(xs, ref["x"]))}
                rec["bit_identical"] = bool(rec["history_bit_identical"] and rec["solution_bit_identical"] and rec["same_iters_isconverged"])
                parity["transports"][f"group/{layout}"] = rec
                group["down"](es, g2)
            ok = [k2 for k2, v in parity["transports"].items() if v.get("bit_identical")]
            parity["bit_identical"] = bool(ok) and all(v.get("bit_identical") for v in parity["transports"].values())
            parity["transports_bit_identical"] = ok
    if transport == "native" and default_layout != "csr-rowblock" and not getattr(args, "no_csr", False):
        arm_watchdog()
        note(f"contract loop: transport {chosen} on the plain CSR arrays")
        e3, c3, i3, failure = bring_up(chosen, "csr", big, 0.0, 10 ** 9)
        if failure:
            contract = {"came_up": False, "failure": failure}
        else:
            state.update(k=0, it=i3)
            first = []
            run_steps(max(Wm, 8), 1, keep=first)
            ms0, cnt0 = C.c_double(), C.c_int64()
            pkg._lib.check(e3.L.mik_cgd_profile(e3.handle, 1, None, None), "mik_cgd_profile", e3.ctx.handle)     # HIP events around every SpMV launch of the loop
            tms = timed(1, K)
            steps_timed = state["k"] - max(Wm, 8)
            pkg._lib.check(e3.L.mik_cgd_profile(e3.handle, 0, C.byref(ms0), C.byref(cnt0)), "mik_cgd_profile", e3.ctx.handle)
            tb3 = timed(25, kb)
            dt3 = float(np.median(tms))
            spmv_ms = ms0.value / max(steps_timed, 1)                   # per STEP (a step whose halo is ordered by events launches its SpMV in two parts)
            u3 = pkg.HipVector.wrap(e3.u_ext.data_ptr(), plan.n_loc + plan.n_ghost, np.float64, e3.ctx, owner=e3.u_ext)
            b2b = e3.A.time_spmv(u3, e3.c, reps=20, fused_dot=True)
            per_rank = boot.all_gather_objects((spmv_ms, int(cnt0.value), [float(v).hex() for v in first[:8]]))
            contract = {"came_up": True, "kernel": e3.A.spmv_kernel(), "operator_layout": e3.A.layout(), "iters_per_sec": K / dt3, "ms_per_step": dt3 / K * 1e3,
                        "spmv_in_loop_ms": spmv_ms, "spmv_launches_timed": int(cnt0.value), "steps_timed": int(steps_timed),
                        "spmv_in_loop_ms_per_rank": [q[0] for q in per_rank], "spmv_back_to_back_ms": b2b,
                        "batched_25_steps_per_sync_iters_per_sec": kb / float(np.median(tb3)), "timed_regions": len(tms), "final_residual": i3.residual,
                        "first_residuals_equal_the_default_layout_bit_for_bit": bool(all(q[2] == default_first for q in per_rank))}
            tear_down(e3, c3)
        if watchdog["timer"] is not None:
            watchdog["timer"].cancel()
    elif watchdog["timer"] is not None:
        watchdog["timer"].cancel()

    def make_line(note=None):
        """the JSON line from whatever has been measured so far (the watchdog prints it too, with `note`)"""
        halo = int(plan.n_ghost)
        s8 = 8
        iter_alg = alg_bytes + 9 * plan.n_loc * s8               # B_cg = B_spmv + 9 n s, on this rank's slab
        iter_moved = stored_bytes + 8 * plan.n_loc * s8          # the default layout's SpMV + the two fused sweeps (x update rides on the u sweep)
        is_contract = bool(contract and contract.get("came_up"))
        v_ms = contract["ms_per_step"] if is_contract else dt / K * 1e3
        v_ips = 1e3 / v_ms
        pmc = getattr(args, "pmc_traffic", None) or (lambda k, with_source=False: (None, None, None) if with_source else None)
        if is_contract:
            c_ms = contract["spmv_in_loop_ms"]
            c_traffic, c_src, c_ok = pmc(contract["kernel"], with_source=True)
            roofline = {"bound": "hbm", "kernel": contract["kernel"] + "<double, fused dot>", "loop": "contract_csr_loop (rank 0's slab; every rank runs the same loop)",
                        "achieved": alg_bytes / (c_ms * 1e-3) / 1e9, "peak": HBM_PEAK, "unit": "GB/s", "frac": alg_bytes / (c_ms * 1e-3) / 1e9 / HBM_PEAK,
                        "traffic": c_traffic, "traffic_source": c_src, "traffic_binary_matches": c_ok,
                        "traffic_is": "committed constant from separate rocprofv3 --pmc passes of the single-GPU command (same kernel, same rows per GPU), not measured in this run",
                        "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": c_ms, "launches_timed": contract["spmv_launches_timed"],
                        "avg_launch_ms_per_rank": contract["spmv_in_loop_ms_per_rank"], "back_to_back_ms": contract["spmv_back_to_back_ms"],
                        "loop_ms_per_step": contract["ms_per_step"], "loop_iters_per_sec": contract["iters_per_sec"],
                        "loop_algorithmic_bytes_per_step_per_gpu": iter_alg, "loop_gbs_per_gpu": iter_alg / (contract["ms_per_step"] * 1e-3) / 1e9,
                        "loop_frac_per_gpu": iter_alg / (contract["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK, "target": 0.60,
                        "note": "SURVEY.md 8d on one rank's slab: algorithmic bytes of the Int32 CSR SpMV of its n_loc x n_ext block over the average HIP-event time of the "
                                "SpMV launch INSIDE the partitioned cg! loop (mik_cgd_profile; mik_csr_set_layout(A_loc, 0), k_spmv_rowgather).  loop_* = that loop: "
                                "every GPU moves loop_algorithmic_bytes_per_step_per_gpu per step of the ONE global system, so bytes / ms_per_step <= 8 TB/s per GPU."}
        else:
            moved = stored_bytes / (d_b2b_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": f"{default_kernel}<double, fused dot> (operator layout {default_layout}; rank 0, back-to-back on the live u)",
                        "loop": "default layout (the CSR contract loop did not run)", "achieved": moved, "peak": HBM_PEAK, "unit": "GB/s", "frac": moved / HBM_PEAK,
                        "traffic": pmc(default_kernel), "bytes_moved_per_launch": stored_bytes, "avg_launch_ms": d_b2b_ms,
                        "note": "bytes this layout moves per launch over the HIP-event time; NOT the CSR-algorithmic figure"}
        out = {
            "metric": "cg_iters_per_sec", "value": v_ips, "unit": "iters/s", "n_gpus": world, "world_size_checked": world, "steps": K, "warmup": Wm,
            "ms_per_step": v_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "value_is_contract": is_contract,
            "value_semantics": "cg! iterations per second of the ONE global system (K / dt, max over ranks; the absolute number the north star quotes at 1 / 2 / 4 / 8 "
                               "GPUs), on the plain CSR arrays of every rank's slab.  Weak scaling: 16.7 M rows per GPU at every N, so the ideal is value(N) = value(1); "
                               "the whole-job aggregate in slab-iterations (N * K / dt) is `aggregate_slab_iters_per_sec`, in row updates `aggregate_row_updates_per_sec`",
            "value_bytes_per_step_per_gpu": iter_alg if is_contract else iter_moved,
            "value_gbs_per_gpu": (iter_alg if is_contract else iter_moved) / (v_ms * 1e-3) / 1e9,
            "aggregate_slab_iters_per_sec": world * v_ips, "aggregate_row_updates_per_sec": v_ips * n,
            "default_layout_iters_per_sec": K / dt, "default_layout_ms_per_step": dt / K * 1e3,
            "config": {"workload": f"cg! on the {N}x{N}x{nz * world} 3D 7-point Laplacian row-partitioned into {world} z-slab(s) of {N}x{N}x{nz} rows"
                                   + (" (BASELINE.json configs[3]: the 512^3 grid on 8 GPUs)" if (N, nz, world) == (512, 64, 8) else
                                      " (BASELINE.json configs[3] layout, weak-scaled: 16.7 M rows per GPU)" if world > 1 else
                                      " (BASELINE.json configs[1] through the row-partitioned code path)")
                                   + (" -- z-PERIODIC variant: the slab exchanges its 2 N^2 halo entries with itself over RCCL (MIK_DIST_SELF_HALO)" if self_halo else ""),
                       "n": int(n), "n_per_gpu": plan.n_loc, "nnz_per_gpu": nnz_loc, "halo_doubles_received_per_rank": halo,
                       "host_sync_per_step": 1, "reltol_in_timed_loop": 0.0,
                       "operator_layout_of_the_timed_loop": "csr (mik_csr_set_layout(A_loc, 0): Int32 rowptr / col / val, k_spmv_rowgather)" if is_contract else default_layout,
                       "timed_regions": contract["timed_regions"] if is_contract else len(times),
                       "transport": ({"rccl": "RCCL inside libmik.so (mik_cgd_iterate_many: ncclSend/ncclRecv halo on a side stream underneath the sweep over u "
                                              "+ 2 ncclAllGather of one double per rank per step)",
                                      "rccl+mailbox": "halo by ncclSend/ncclRecv on a side stream; the two scalars of a step as stores into peer-mapped "
                                                      "mailboxes, summed inside the finalising kernels (no collective launch on the compute stream)",
                                      "mailbox": "peer-mapped mailbox (no RCCL): scalars as stores into IPC-mapped slots, halo pushed into the neighbours' IPC-mapped landing buffers and copied into the ghost tail by the receiver"}[chosen]
                                     if transport == "native" and (uses_rccl or world > 1 or self_halo) else
                                     "none (world of one)" if transport == "native" else
                                     "in-process group (include/mik.h Transport 2): rank 0's host thread drives every slab on its own device, halos and the two scalars "
                                     "of a step as event-ordered peer copies -- the last resort when no transport between processes passed its self-test"
                                     if transport == "group" else "torch.distributed driven from Python (legacy)"),
                       "transport_chosen": chosen, "transports_measured": transports,
                       "halo_overlap": bool(getattr(eng, "overlap", False)),
                       "operator_build_and_upload_seconds": upload_seconds, "final_residual": contract["final_residual"] if is_contract else it.residual,
                       "default_layout": {"operator_layout": default_layout, "kernel": default_kernel, "iters_per_sec": K / dt, "ms_per_step": dt / K * 1e3,
                                          "bytes_per_step_per_gpu": iter_moved, "gbs_per_gpu": iter_moved / (dt / K) / 1e9, "spmv_back_to_back_ms": d_b2b_ms,
                                          "batched_25_steps_per_sync_iters_per_sec": kb / float(np.median(times_b)), "timed_regions": len(times),
                                          "timed_seconds_total": float(sum(times)), "final_residual": it.residual,
                                          "note": "the same partitioned iteration with every slab in the layout mik_csr_create picks for this constant-coefficient operator "
                                                  "(one mask byte per row instead of the CSR arrays); same residuals bit for bit; NOT the contract figure"}},
            "contract_csr_loop": contract,
            "parity_vs_oracle": parity,
            "roofline": roofline,
            "transport_selftest": selftest,
            "wall": {"seconds_so_far": time.perf_counter() - t_bench0,
                     "expected_seconds_of_the_whole_command_at_8_gpus": "60 - 90: launcher + imports ~8 s, slab generation ~2 s, self-test children ~3 s per transport "
                                                   "(RCCL bootstrap of 8 ranks: up to ~15 s), ~2 s per measured transport, contract loop ~2 s, parity (3 transports x 2 "
                                                   "layouts, small system) ~6 s, CPU baseline on rank 0 ~14 s.  Measured with every rank on ONE GPU (one transport): "
                                                   "2 / 3 / 4 ranks = 21 / 13 / 15 s incl. launcher (profiles/r06_first_contact_wall.json); hard limits: self-test child "
                                                   "75 s, mailbox waits 10 s, watchdog 150 s per leg",
                     "limit_seconds": 600},
        }
        if boot_failure:
            out["config"]["bootstrap_failure"] = boot_failure
        if note:
            out["config"]["watchdog"] = note
        return out

    state["line_ready"] = True
    if transport == "native":
        # ---- (1) parity, LAST: whatever happens in it, the timed line exists (the watchdog prints it with parity_vs_oracle = "not reached") ----------
        arm_watchdog()
        run_parity()
        if watchdog["timer"] is not None:
            watchdog["timer"].cancel()
    note("all GPU legs done" + ("; CPU baseline on rank 0" if not getattr(args, "no_cpu_baseline", False) else ""))
    if rank == 0:
        out = make_line()
        fn = getattr(args, "cpu_baseline_fn", None)
        if fn is not None and not getattr(args, "no_cpu_baseline", False):
            # the reference-shaped CPU restatement on this box's host cores, in the same run (rank 0 only; the other ranks wait at the
            # teardown): one rank's share is a 16.7 M-row system, i.e. the 256^3 workload of the single-GPU line
            cb = fn(256, max(40, int(getattr(args, "cpu_iters", 120))))       # never fewer than 40 iterations (VERDICT r5 #1c)
            cb.pop("_history", None)
            cb["sample"] += f"; one rank's share of the {world}-rank system has the same 16.7 M rows (the CPU would need {world} x as long per iteration of the global system)"
            out["cpu_baseline"] = cb
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()
