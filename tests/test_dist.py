"""Row-partitioned CG (dist.py): partition / halo-plan logic, the in-process loopback ranks, and the
torch.distributed orchestration on gloo with world_size 2 (CPU, via a numpy test double of the engine),
plus the same loopback on the real HIP engine (-m gpu)."""
import importlib
import os
import sys

import numpy as np
import pytest

from conftest import KN, ROOT

sys.path.insert(0, os.path.join(ROOT, "tests"))


def dist_mod(pkg):
    return importlib.import_module(pkg.__name__ + ".dist")


def global_csr(orc, N, NZ):
    """The N x N x NZ Laplacian as global CSR via dist._laplace_rows (all rows)"""
    return None


# ------------------------------------------------------------------------------------------------
# partitioning and plans
# ------------------------------------------------------------------------------------------------
def test_partition_rows(pkg):
    d = dist_mod(pkg)
    assert d.partition_rows(100, 4).tolist() == [0, 25, 50, 75, 100]
    assert d.partition_rows(10, 3).tolist() == [0, 3, 6, 10]
    off = d.partition_rows(6 ** 3, 4, align=36)
    assert off[0] == 0 and off[-1] == 216 and all(o % 36 == 0 for o in off) and np.all(np.diff(off) > 0)
    assert d.partition_rows(5, 8).tolist()[-1] == 5            # more ranks than rows: empty blocks allowed


@pytest.mark.parametrize("P", [1, 2, 3, 5])
def test_halo_plans_are_consistent(pkg, P):
    """localize + complete_plan: renumbered local blocks reproduce the global SpMV; send/recv lists mirror"""
    d = dist_mod(pkg)
    N, NZ = 5, 7
    n, ptr, idx, val = d._laplace_rows(pkg, N, NZ, 0, N * N * NZ, np.float64)
    import scipy.sparse as sp
    Aglob = sp.csr_matrix((val, idx, ptr), shape=(n, n))
    offsets = d.partition_rows(n, P, align=N * N)
    x = np.random.default_rng(0).standard_normal(n)
    plans, blocks = [], []
    for p in range(P):
        _, pp, ii, vv = d._laplace_rows(pkg, N, NZ, offsets[p], offsets[p + 1], np.float64)
        li, plan = d.localize_block(pp, ii, offsets, p)
        plans.append(plan)
        blocks.append((pp, li, vv))
    needs = [pl.ghost_gids for pl in plans]
    for pl in plans:
        d.complete_plan(pl, offsets, needs)
    y = np.empty(n)
    for p, (pl, (pp, li, vv)) in enumerate(zip(plans, blocks)):
        r0, r1 = offsets[p], offsets[p + 1]
        x_ext = np.concatenate([x[r0:r1], np.zeros(pl.n_ghost)])
        for (peer, off, cnt) in pl.recv:                      # what the exchange would deliver
            src = plans[peer]
            soff = next(o for (q, o, c) in src.send if q == p)
            assert next(c for (q, o, c) in src.send if q == p) == cnt
            packed = x[offsets[peer]:offsets[peer + 1]][src.send_idx[soff:soff + cnt]]
            x_ext[pl.n_loc + off: pl.n_loc + off + cnt] = packed
        assert np.array_equal(x_ext[pl.n_loc:], x[pl.ghost_gids])
        Aloc = sp.csr_matrix((vv, li, pp), shape=(pl.n_loc, pl.n_loc + pl.n_ghost))
        y[r0:r1] = Aloc @ x_ext
    assert np.array_equal(y, Aglob @ x)
    if P > 1:
        assert plans[0].n_ghost == N * N and plans[1].n_ghost in (N * N, 2 * N * N)     # z-slab halos are planes


# ------------------------------------------------------------------------------------------------
# loopback ranks with the numpy engine (CPU)
# ------------------------------------------------------------------------------------------------
def make_engines(pkg, orc, N, NZ, P, make_engine, x0=None, b_scale=1.0):
    d = dist_mod(pkg)
    n = N * N * NZ
    offsets = d.partition_rows(n, P, align=N * N)
    plans, parts = [], []
    for p in range(P):
        _, pp, ii, vv = d._laplace_rows(pkg, N, NZ, offsets[p], offsets[p + 1], np.float64)
        li, plan = d.localize_block(pp, ii, offsets, p)
        plans.append(plan)
        parts.append((pp, li, vv))
    for pl in plans:
        d.complete_plan(pl, offsets, [q.ghost_gids for q in plans])
    b = pkg.fixtures.hashed_rhs(n) * b_scale
    engines = [make_engine(pp, li, vv, pl, b[offsets[p]:offsets[p + 1]], None if x0 is None else x0[offsets[p]:offsets[p + 1]])
               for p, (pl, (pp, li, vv)) in enumerate(zip(plans, parts))]
    return engines, offsets, b


def oracle_history(orc, pkg, N, NZ, offsets, b, shape, x0=None, maxiter=None):
    d = dist_mod(pkg)
    n, ptr, idx, val = d._laplace_rows(pkg, N, NZ, 0, N * N * NZ, np.float64)
    A = orc.CSC(n, ptr, idx, val, 0)            # symmetric: CSR arrays are a valid CSC
    orc.set_partition(offsets)
    try:
        return orc.cg(A, b, x0, mode="tree", shape=shape, maxiter=maxiter)
    finally:
        orc.set_partition(None)


@pytest.mark.parametrize("P", [1, 2, 3, 4])
@pytest.mark.parametrize("with_x0", [False, True])
def test_loopback_numpy_engine_matches_partitioned_oracle(pkg, orc, P, with_x0):
    from dist_double import NumpyEngine
    d = dist_mod(pkg)
    N, NZ, shape = 6, 8, (1, 1, 2, 2)
    x0 = np.random.default_rng(3).standard_normal(N * N * NZ) if with_x0 else None
    mk = lambda pp, li, vv, pl, bl, xl: NumpyEngine(orc, pp, li, vv, pl, bl, xl, abstol=0.0, reltol=1.5e-8, maxiter=10 ** 6, shape=shape)
    engines, offsets, b = make_engines(pkg, orc, N, NZ, P, mk, x0)
    lb = d.LoopbackCG(engines, maxiter=10 ** 6)
    hist = lb.solve()
    xo, ho = oracle_history(orc, pkg, N, NZ, offsets, b, shape, x0)
    assert hist.size == ho["iters"] and np.array_equal(hist, ho["resnorm"])
    assert np.array_equal(lb.solution(), xo)
    assert lb.residual <= lb.tol


# ------------------------------------------------------------------------------------------------
# torch.distributed on gloo, world_size 2 (CPU)
# ------------------------------------------------------------------------------------------------
def _gloo_worker(rank, world, port, N, nz, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import __graft_entry__ as graft
    from dist_double import NumpyEngine
    pkg = graft.load_package()
    orc = graft.load_oracle()
    d = importlib.import_module(pkg.__name__ + ".dist")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = d.TorchComm()
    ptr, li, val, plan, b_loc, n, offsets = d.build_rank_problem(pkg, comm, N, nz_per_rank=nz)
    eng = NumpyEngine(orc, ptr, li, val, plan, b_loc, abstol=0.0, reltol=1.5e-8, maxiter=10 ** 6)
    it = d.DistCGIterable(eng, comm, maxiter=10 ** 6)
    hist, iteration = [], 0
    while True:                                   # mix single steps and batches
        h = it.iterate_many(iteration, 1 if iteration < 3 else 7)
        if h.size == 0:
            break
        hist.append(h)
        iteration += h.size
    np.save(os.path.join(out_dir, f"hist{rank}.npy"), np.concatenate(hist))
    np.save(os.path.join(out_dir, f"x{rank}.npy"), eng.solution())
    np.save(os.path.join(out_dir, f"off{rank}.npy"), offsets)
    dist.destroy_process_group()


def test_gloo_world2_matches_partitioned_oracle(pkg, orc, tmp_path):
    import torch.multiprocessing as mp
    N, nz, world = 6, 4, 2
    port = 29600 + os.getpid() % 300
    mp.spawn(_gloo_worker, args=(world, port, N, nz, str(tmp_path)), nprocs=world, join=True)
    h0, h1 = np.load(tmp_path / "hist0.npy"), np.load(tmp_path / "hist1.npy")
    assert np.array_equal(h0, h1)                 # rank-ordered sums: every rank holds identical scalars
    offsets = np.load(tmp_path / "off0.npy")
    b = pkg.fixtures.hashed_rhs(N * N * nz * world)
    xo, ho = oracle_history(orc, pkg, N, nz * world, offsets, b, (1, 1, 2, 2))
    assert h0.size == ho["iters"] and np.array_equal(h0, ho["resnorm"])
    x = np.concatenate([np.load(tmp_path / "x0.npy"), np.load(tmp_path / "x1.npy")])
    assert np.array_equal(x, xo)


def _gloo_gmres_worker(rank, world, port, method, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scipy.sparse as sp
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    from dist_double import NumpyGmresRank
    pkg = graft.load_package()
    orc = graft.load_oracle()
    d = importlib.import_module(pkg.__name__ + ".dist")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = d.TorchComm()
    n, colptr, rowval, nzval, b = pkg.fixtures.advection_dominated(7, 300.0)        # nonsymmetric: CSC(A) != CSR(A)
    S = sp.csc_matrix((nzval, rowval - 1, colptr - 1), shape=(n, n)).tocsr()
    offsets = np.array([0, 150, n]) if world == 2 else d.partition_rows(n, world)    # cut inside a grid plane
    blk = S[offsets[rank]:offsets[rank + 1]]
    ptr, idx, val = blk.indptr.astype(np.int64), blk.indices.astype(np.int64), blk.data
    local_idx, plan = d.localize_block(ptr, idx, offsets, rank)
    d.complete_plan(plan, offsets, comm.all_gather_objects(plan.ghost_gids))
    x_ext = torch.zeros(max(plan.n_loc + plan.n_ghost, 1), dtype=torch.float64)
    send_buf = torch.zeros(max(plan.n_send, 1), dtype=torch.float64)
    links = d.PartitionLinks(comm, plan, send_buf, x_ext)
    g = NumpyGmresRank(orc, links, ptr, local_idx, val, plan, b[offsets[rank]:offsets[rank + 1]], restart=6,
                       reltol=1.5e-8, maxiter=60, method=method)
    hist, iteration = [], 0
    while (nxt := g.iterate(iteration)) is not None:
        hist.append(nxt[0])
        iteration = nxt[1]
    np.save(os.path.join(out_dir, f"hist{rank}.npy"), np.array(hist))
    np.save(os.path.join(out_dir, f"x{rank}.npy"), g.x)
    np.save(os.path.join(out_dir, f"mv{rank}.npy"), np.array([g.mv_products]))
    dist.destroy_process_group()


@pytest.mark.parametrize("method", ["mgs", "cgs"])
def test_gloo_world2_partitioned_gmres_links_match_oracle(pkg, orc, tmp_path, method):
    """The halo / rank-ordered-sum callbacks of the partitioned GMRES (dist.PartitionLinks over gloo) with a
    numpy stand-in for the device arithmetic: bit-identical to the oracle's gmres with the same partition."""
    import torch.multiprocessing as mp
    world = 2
    port = 29300 + os.getpid() % 250 + (7 if method == "cgs" else 0)
    mp.spawn(_gloo_gmres_worker, args=(world, port, method, str(tmp_path)), nprocs=world, join=True)
    A, _ = orc.advdiff(7, 300.0)
    b = pkg.fixtures.advection_dominated(7, 300.0)[4]     # the workers' rhs (the product-side fixture)
    orc.set_partition(np.array([0, 150, A.n]))
    try:
        xo, ho = orc.gmres(A, b, restart=6, maxiter=60, reltol=1.5e-8, orth_meth=method, mode="tree", shape=(2, 2))
    finally:
        orc.set_partition(None)
    h0, h1 = np.load(tmp_path / "hist0.npy"), np.load(tmp_path / "hist1.npy")
    assert np.array_equal(h0, h1) and np.array_equal(h0, ho["resnorm"])
    assert int(np.load(tmp_path / "mv0.npy")[0]) == ho["mvps"]
    assert np.array_equal(np.concatenate([np.load(tmp_path / "x0.npy"), np.load(tmp_path / "x1.npy")]), xo)


def test_rank_ordered_sum_and_thread_comm(pkg):
    import threading
    d = dist_mod(pkg)
    parts = np.array([[1e16, 1.0], [1.0, 1e16], [-1e16, -1e16]])
    assert np.array_equal(d.rank_ordered_sum(parts), np.array([(1e16 + 1.0) - 1e16, (1.0 + 1e16) - 1e16]))
    assert d.rank_ordered_sum(parts.astype(np.float32)).dtype == np.float32
    comms, got = d.ThreadComm.world(3), [None] * 3

    def work(r):
        got[r] = comms[r].all_gather_host(np.array([float(r), 10.0 * r]))

    ts = [threading.Thread(target=work, args=(r,)) for r in range(3)]
    [t.start() for t in ts]
    [t.join(30) for t in ts]
    assert all(np.array_equal(g, np.array([[0.0, 0.0], [1.0, 10.0], [2.0, 20.0]])) for g in got)


# ------------------------------------------------------------------------------------------------
# the HIP engine: loopback ranks on one GPU
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("P", [1, 2, 4])
@pytest.mark.parametrize("with_x0", [False, True])
def test_loopback_hip_engine_matches_partitioned_oracle(pkg, orc, ctx, P, with_x0):
    import torch
    d = dist_mod(pkg)
    N, NZ = 12, 16
    shape = ctx.cg_shape(np.float64)
    x0 = np.random.default_rng(3).standard_normal(N * N * NZ) if with_x0 else None
    stream = torch.cuda.Stream()
    mk = lambda pp, li, vv, pl, bl, xl: d.HipEngine(pkg, pp, li, vv, pl, bl, xl, abstol=0.0, reltol=1.5e-8, maxiter=10 ** 6, stream=stream)
    engines, offsets, b = make_engines(pkg, orc, N, NZ, P, mk, x0)
    lb = d.LoopbackCG(engines, maxiter=10 ** 6)
    hist = lb.solve()
    xo, ho = oracle_history(orc, pkg, N, NZ, offsets, b, shape, x0)
    assert hist.size == ho["iters"] and np.array_equal(hist, ho["resnorm"])
    assert np.array_equal(lb.solution(), xo)


def test_interior_row_blocks_of_a_slab(pkg):
    d = dist_mod(pkg)
    N, NZ, P = 16, 12, 3                          # one plane = 256 rows = one row-block
    offsets = d.partition_rows(N * N * NZ, P, align=N * N)
    for p, want in ((0, (0, 3)), (1, (1, 3)), (2, (1, 4))):
        _, pp, ii, vv = d._laplace_rows(pkg, N, NZ, offsets[p], offsets[p + 1], np.float64)
        li, plan = d.localize_block(pp, ii, offsets, p)
        assert d.interior_row_blocks(pp, li, plan.n_loc) == want
    # a partition whose halo rows are scattered over all blocks has no interior range
    ptr = np.arange(0, 2 * 1024 + 1, 2)
    li = np.stack([np.arange(1024), np.where(np.arange(1024) % 100 == 0, 1024, np.arange(1024))], axis=1).ravel()
    assert d.interior_row_blocks(ptr, li, 1024) is None
    assert d.interior_row_blocks(ptr, np.zeros(2048, np.int64), 1024) == (0, 4)      # no halo at all


@pytest.mark.gpu
@pytest.mark.parametrize("P", [1, 2, 3])
@pytest.mark.parametrize("overlap", ["1", "0"])
def test_loopback_with_halo_overlap_matches_partitioned_oracle(pkg, orc, ctx, P, overlap, monkeypatch):
    """slabs with an interior range: step B split into interior rows (before the halo arrives) + boundary rows"""
    import torch
    monkeypatch.setenv("MIK_DIST_OVERLAP", overlap)
    d = dist_mod(pkg)
    N, NZ = 16, 12
    shape = ctx.cg_shape(np.float64)
    stream = torch.cuda.Stream()
    mk = lambda pp, li, vv, pl, bl, xl: d.HipEngine(pkg, pp, li, vv, pl, bl, xl, abstol=0.0, reltol=1.5e-8, maxiter=10 ** 6, stream=stream)
    engines, offsets, b = make_engines(pkg, orc, N, NZ, P, mk)
    assert all(e.overlap == (overlap == "1") for e in engines)
    lb = d.LoopbackCG(engines, maxiter=10 ** 6)
    hist = lb.solve()
    xo, ho = oracle_history(orc, pkg, N, NZ, offsets, b, shape)
    assert hist.size == ho["iters"] and np.array_equal(hist, ho["resnorm"])
    assert np.array_equal(lb.solution(), xo)


@pytest.mark.gpu
def test_dist_world1_equals_single_gpu_path(pkg, orc, ctx):
    """P = 1 through DistCGIterable + SelfComm is bit-identical to the fused single-GPU iterable"""
    d = dist_mod(pkg)
    N = 16
    comm = d.SelfComm()
    ptr, li, val, plan, b_loc, n, offsets = d.build_rank_problem(pkg, comm, N, nz_per_rank=N)
    eng = d.HipEngine(pkg, ptr, li, val, plan, b_loc, abstol=0.0, reltol=1.5e-8, maxiter=10 ** 6)
    it = d.DistCGIterable(eng, comm, maxiter=10 ** 6)
    hist = np.array(list(it))
    nA, colptr, rowval, nzval = pkg.fixtures.laplace_matrix(N, 3)
    x, ch = pkg.cg(pkg.HipCSR(nA, nA, colptr, rowval, nzval), pkg.HipVector.from_numpy(b_loc), reltol=1.5e-8, log=True)
    assert np.array_equal(hist, ch["resnorm"]) and np.array_equal(eng.solution(), x.to_numpy())


# ------------------------------------------------------------------------------------------------
# two PROCESSES sharing one GPU: the real multi-process orchestration (DistCGIterable + TorchComm +
# HipEngine), with gloo staging the device buffers through the host because RCCL refuses two ranks
# on one device ("Duplicate GPU detected")
# ------------------------------------------------------------------------------------------------
def _gpu_worker(rank, world, port, N, nz, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.load_package()
    d = importlib.import_module(pkg.__name__ + ".dist")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = d.TorchComm()
    ptr, li, val, plan, b_loc, n, offsets = d.build_rank_problem(pkg, comm, N, nz_per_rank=nz)
    eng = d.HipEngine(pkg, ptr, li, val, plan, b_loc, abstol=0.0, reltol=1.5e-8, maxiter=10 ** 6, device=0)
    it = d.DistCGIterable(eng, comm, maxiter=10 ** 6)
    hist, iteration = [], 0
    while True:
        h = it.iterate_many(iteration, 1 if iteration < 2 else 9)
        if h.size == 0:
            break
        hist.append(h)
        iteration += h.size
    np.save(os.path.join(out_dir, f"hist{rank}.npy"), np.concatenate(hist))
    np.save(os.path.join(out_dir, f"x{rank}.npy"), eng.solution())
    np.save(os.path.join(out_dir, f"off{rank}.npy"), offsets)
    eng.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_multiprocess_ranks_on_one_gpu_match_partitioned_oracle(pkg, orc, ctx, tmp_path, world):
    import torch.multiprocessing as mp
    N, nz = 16, 4                                 # 4 planes of 256 rows per rank: interior row-blocks exist (halo overlap path)
    port = 29700 + os.getpid() % 200 + world
    mp.spawn(_gpu_worker, args=(world, port, N, nz, str(tmp_path)), nprocs=world, join=True)
    hs = [np.load(tmp_path / f"hist{r}.npy") for r in range(world)]
    assert all(np.array_equal(hs[0], h) for h in hs)
    offsets = np.load(tmp_path / "off0.npy")
    b = pkg.fixtures.hashed_rhs(N * N * nz * world)
    xo, ho = oracle_history(orc, pkg, N, nz * world, offsets, b, ctx.cg_shape(np.float64))
    assert hs[0].size == ho["iters"] and np.array_equal(hs[0], ho["resnorm"])
    x = np.concatenate([np.load(tmp_path / f"x{r}.npy") for r in range(world)])
    assert np.array_equal(x, xo)


def _mailbox_worker(rank, world, port, N, nz, out_dir, scale, batch, knob6):
    """one rank = one process, all on GPU 0: mik_cgd_iterate_many over the mailbox transport (gloo only carries the IPC handles)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), MIK_MAILBOX_TIMEOUT_MS="20000")
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.load_package()
    d = importlib.import_module(pkg.__name__ + ".dist")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    boot = d.TorchComm()
    pkg.lib().mik_set_tuning(6, knob6)
    ptr, li, val, plan, b_loc, n, offsets = d.build_rank_problem(pkg, boot, N, nz_per_rank=nz)
    eng = d.HipEngine(pkg, ptr, li, val, plan, b_loc * scale, abstol=0.0, reltol=1.5e-8, maxiter=10 ** 6, device=0)
    nc = d.NativeComm(pkg, eng.ctx, boot, transport="mailbox")
    assert not nc.uses_rccl() and nc.mailbox()[0]
    it = d.NativeDistCGIterable(pkg, eng, nc, maxiter=10 ** 6)
    hist, iteration = [], 0
    while True:
        h = it.iterate_many(iteration, 1 if iteration < 2 else batch)
        if h.size == 0:
            break
        hist.append(h)
        iteration += h.size
    np.save(os.path.join(out_dir, f"hist{rank}.npy"), np.concatenate(hist))
    np.save(os.path.join(out_dir, f"x{rank}.npy"), eng.solution())
    np.save(os.path.join(out_dir, f"off{rank}.npy"), offsets)
    boot.barrier()
    eng.close()
    nc.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,scale,batch,knob6", [(2, 1.0, 9, 0), (3, 1.0, 9, 0), (2, 1e-140, 1, 0), (3, 1e140, 7, 0), (2, 1.0, 5, 8), (2, 1.0, 5, 1)])
def test_mailbox_transport_ranks_in_processes_on_one_gpu(pkg, orc, ctx, tmp_path, world, scale, batch, knob6):
    """VERDICT r3 #4 / SURVEY.md section 5 backend B: the library-driven step (mik_cgd_iterate_many) with NO collective launch --
    the two scalars of a step as {value, sequence number} stores into peer-mapped mailboxes (fine-grained device memory opened with
    hipIpcOpenMemHandle), summed in rank order on every rank; the halo pushed into the neighbours' IPC-mapped ghost regions by a
    kernel on the side stream, flags instead of events.  2 and 3 ranks as separate PROCESSES on the one GPU of the box (HIP IPC has
    no one-rank-per-device rule; RCCL is not loaded at all); bit-exact against the partition-aware oracle, also on a right-hand
    side that sends every step through the scaled norm across the ranks (three more gathers per step, lane 2 of the mailbox).
    MIK_KNOB_TRANSPORT: 8 = the step's scalars through the one-wave gather launches instead of inside the finalisers; 1 = the pack ->
    push ordering by an event instead of a flag."""
    import torch.multiprocessing as mp
    N, nz = 16, 4
    port = 29300 + os.getpid() % 300 + 7 * world + batch + 11 * knob6
    mp.spawn(_mailbox_worker, args=(world, port, N, nz, str(tmp_path), scale, batch, knob6), nprocs=world, join=True)
    hs = [np.load(tmp_path / f"hist{r}.npy") for r in range(world)]
    assert all(np.array_equal(hs[0], h) for h in hs)
    offsets = np.load(tmp_path / "off0.npy")
    b = pkg.fixtures.hashed_rhs(N * N * nz * world) * scale
    xo, ho = oracle_history(orc, pkg, N, nz * world, offsets, b, ctx.cg_shape(np.float64))
    assert ho["iters"] > 10 and ho["isconverged"]
    assert hs[0].size == ho["iters"] and np.array_equal(hs[0], ho["resnorm"])
    x = np.concatenate([np.load(tmp_path / f"x{r}.npy") for r in range(world)])
    assert np.array_equal(x, xo)


# ------------------------------------------------------------------------------------------------
# the exchanges INSIDE libmik.so (include/mik.h "Transport 1 / 2"): no host code between the phases
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("P", [1, 2, 3, 4])
@pytest.mark.parametrize("with_x0", [False, True])
def test_inprocess_group_matches_partitioned_oracle(pkg, orc, ctx, P, with_x0):
    """mik_cgd_group_*: one host thread, P ranks with their own ctx / stream on one GPU, halos and scalar gathers as
    event-ordered peer copies enqueued by the library: bit-exact against the partition-aware oracle, batches of mixed
    length, with and without an interior range (N = 16: a plane is one row-block)"""
    d = dist_mod(pkg)
    for N, NZ in ((12, 16), (16, 12)):
        shape = ctx.cg_shape(np.float64)
        x0 = np.random.default_rng(3).standard_normal(N * N * NZ) if with_x0 else None
        mk = lambda pp, li, vv, pl, bl, xl: d.HipEngine(pkg, pp, li, vv, pl, bl, xl, abstol=0.0, reltol=1.5e-8, maxiter=10 ** 6)
        engines, offsets, b = make_engines(pkg, orc, N, NZ, P, mk, x0)
        assert len({e.stream.cuda_stream for e in engines}) == P        # every rank on its own stream
        grp = d.GroupCG(pkg, engines, maxiter=10 ** 6)
        hist, iteration = [], 0
        while True:
            h = grp.iterate_many(iteration, 1 if iteration < 2 else 13)
            if h.size == 0:
                break
            hist.append(h)
            iteration += h.size
        hist = np.concatenate(hist)
        xo, ho = oracle_history(orc, pkg, N, NZ, offsets, b, shape, x0)
        assert hist.size == ho["iters"] and np.array_equal(hist, ho["resnorm"])
        assert np.array_equal(grp.solution(), xo)
        grp.close()
        for e in engines:
            e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("P", [1, 2, 3])
@pytest.mark.parametrize("scale,with_x0", [(1e-140, False), (1e140, False), (1e-140, True)])
def test_inprocess_group_badly_scaled_rhs_takes_the_scaled_norm_across_ranks(pkg, orc, ctx, P, scale, with_x0):
    """VERDICT r2 / ADVICE r2: |r|^2 summed over the ranks leaves the safe range in every step (and in cg_iterator!): the ranks
    freeze the batch on the same total, exchange max |r_i|, rescale by the common power of two and go on -- the history of the
    partition-aware oracle (whose safe norm does exactly that), bit for bit, instead of MIK_ERR_RANGE; switching from one GPU to
    N does not change what converges"""
    d = dist_mod(pkg)
    N, NZ = 12, 12
    shape = ctx.cg_shape(np.float64)
    x0 = np.random.default_rng(5).standard_normal(N * N * NZ) * scale if with_x0 else None
    mk = lambda pp, li, vv, pl, bl, xl: d.HipEngine(pkg, pp, li, vv, pl, bl, xl, abstol=0.0, reltol=1.5e-8, maxiter=10 ** 6)
    engines, offsets, b = make_engines(pkg, orc, N, NZ, P, mk, x0, b_scale=scale)
    grp = d.GroupCG(pkg, engines, maxiter=10 ** 6)
    hist, iteration = [], 0
    while True:
        h = grp.iterate_many(iteration, 1 if iteration < 2 else 9)
        if h.size == 0:
            break
        hist.append(h)
        iteration += h.size
    hist = np.concatenate(hist)
    xo, ho = oracle_history(orc, pkg, N, NZ, offsets, b, shape, x0)
    assert ho["iters"] > 10 and ho["isconverged"]
    assert hist.size == ho["iters"] and np.array_equal(hist, ho["resnorm"])
    assert np.array_equal(grp.solution(), xo)
    grp.close()
    for e in engines:
        e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("P", [2, 3, 4])
@pytest.mark.parametrize("knobs", [0, KN.X_IN_STEP, KN.HALO_AFTER_SWEEP])
def test_inprocess_group_early_halo_step(pkg, orc, ctx, P, knobs):
    """Tall slabs (48 planes of 8 x 8): the boundary planes of EVERY rank, middle ranks with two runs included, are at most a
    quarter of its rows, so the group runs the step of cgd_enqueue_head -- boundary planes updated and packed first
    (phase 9), halo copies, bulk of the sweep (phase 8), interior row-blocks, then the boundary row-blocks in one launch
    around the interior range (mik_spmv_launch_outside) -- with x .+= alpha .* u riding on the next sweep and the flush at
    the end of a batch.  Bit-exact against the partition-aware oracle; MIK_KNOB_CG_STEP bit 0 (x updated in the step) and bit 2 (halo
    after the whole sweep) give the same bits."""
    d = dist_mod(pkg)
    N, NZ = 8, 48
    L = pkg.lib()
    L.mik_set_tuning(KN.CG_STEP, knobs)
    try:
        shape = ctx.cg_shape(np.float64)
        x0 = np.random.default_rng(11).standard_normal(N * N * NZ)
        mk = lambda pp, li, vv, pl, bl, xl: d.HipEngine(pkg, pp, li, vv, pl, bl, xl, abstol=0.0, reltol=1.5e-8, maxiter=10 ** 6)
        engines, offsets, b = make_engines(pkg, orc, N, NZ, P, mk, x0)
        grp = d.GroupCG(pkg, engines, maxiter=10 ** 6)
        early = grp.halo_early()
        if knobs == KN.HALO_AFTER_SWEEP:
            assert all(runs == 0 for runs, _, _ in early)
        else:
            want = [1 if p in (0, P - 1) else 2 for p in range(P)]
            assert [runs for runs, _, _ in early] == want and all(m for _, _, m in early)
            assert [rows for _, rows, _ in early] == [N * N * w for w in want]
        hist, iteration = [], 0
        while True:
            h = grp.iterate_many(iteration, 1 if iteration < 2 else 7)
            if h.size == 0:
                break
            hist.append(h)
            iteration += h.size
            if iteration == 9:                                       # between batches x is complete (the flush of phase 6)
                xm = grp.solution()
                assert np.isfinite(xm).all() and not np.array_equal(xm, x0)
        hist = np.concatenate(hist)
        xo, ho = oracle_history(orc, pkg, N, NZ, offsets, b, shape, x0)
        assert hist.size == ho["iters"] and np.array_equal(hist, ho["resnorm"])
        assert np.array_equal(grp.solution(), xo)
        grp.close()
        for e in engines:
            e.close()
    finally:
        L.mik_set_tuning(KN.CG_STEP, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("force_rccl", [False, True])
def test_native_comm_world1_equals_single_gpu_path(pkg, orc, ctx, force_rccl):
    """mik_cgd_iterate_many through a mik_comm: a world of one without the library, and a REAL RCCL communicator of one
    rank (ncclCommInitRank + ncclAllGather issued from inside libmik.so on the ctx stream) -- both bit-identical to the
    fused single-GPU iterable"""
    d = dist_mod(pkg)
    N = 16
    boot = d.SelfComm()
    ptr, li, val, plan, b_loc, n, offsets = d.build_rank_problem(pkg, boot, N, nz_per_rank=N)
    eng = d.HipEngine(pkg, ptr, li, val, plan, b_loc, abstol=0.0, reltol=1.5e-8, maxiter=10 ** 6)
    nc = d.NativeComm(pkg, eng.ctx, boot, force_rccl=force_rccl)
    assert nc.uses_rccl() == force_rccl
    it = d.NativeDistCGIterable(pkg, eng, nc, maxiter=10 ** 6)
    hist, iteration = [], 0
    while True:
        h = it.iterate_many(iteration, 1 if iteration < 3 else 20)
        if h.size == 0:
            break
        hist.append(h)
        iteration += h.size
    hist = np.concatenate(hist)
    nA, colptr, rowval, nzval = pkg.fixtures.laplace_matrix(N, 3)
    x, ch = pkg.cg(pkg.HipCSR(nA, nA, colptr, rowval, nzval), pkg.HipVector.from_numpy(b_loc), reltol=1.5e-8, log=True)
    assert np.array_equal(hist, ch["resnorm"]) and np.array_equal(eng.solution(), x.to_numpy())
    # the reduce entry a row-partitioned GMRES host wraps: rank-ordered sums (identity in a world of one)
    v = np.array([1.5, -2.0, 3.25])
    pkg._lib.check(pkg.lib().mik_comm_allgather_sum(nc.handle, 0, 3, v.ctypes.data_as(__import__("ctypes").c_void_p)), "mik_comm_allgather_sum")
    assert v.tolist() == [1.5, -2.0, 3.25]
    eng.close()
    nc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("knobs", [(0, 0), (1, 0), (0, KN.SEPARATE_ALPHA), (1, KN.SEPARATE_ALPHA)])
@pytest.mark.parametrize("scale,with_x0", [(1e-140, False), (1e140, False), (1e-140, True)])
@pytest.mark.parametrize("batch", [1, 9])
def test_rccl_path_badly_scaled_rhs_freezes_on_the_last_step_of_a_batch(pkg, orc, ctx, knobs, scale, with_x0, batch):
    """mik_cgd_iterate_many -- the RCCL call path, a world of one needs no transport -- on a system whose |r|^2
    leaves the safe range in EVERY step.  With batch = 1 (DistCG.iterate()) every frozen step is the last of its batch: the head
    enqueued ahead applies the step's x .+= alpha .* u but no tail follows to clear the pending flag, and the next call's fresh head
    used to add the same alpha u again -- a silently wrong x under a correct-looking history.  x and the history bit for bit
    against the oracle, with and without look-ahead (MIK_KNOB_NO_LOOKAHEAD), both alpha forms (MIK_KNOB_CG_STEP bit 3), freezes mid-batch and on the last step
    (batch = 9 with a 1-step prologue)."""
    d = dist_mod(pkg)
    N, NZ = 12, 12
    L = pkg.lib()
    L.mik_set_tuning(KN.NO_LOOKAHEAD, knobs[0])
    L.mik_set_tuning(KN.CG_STEP, knobs[1])
    try:
        shape = ctx.cg_shape(np.float64)
        x0 = np.random.default_rng(5).standard_normal(N * N * NZ) * scale if with_x0 else None
        mk = lambda pp, li, vv, pl, bl, xl: d.HipEngine(pkg, pp, li, vv, pl, bl, xl, abstol=0.0, reltol=1.5e-8, maxiter=10 ** 6)
        (eng,), offsets, b = make_engines(pkg, orc, N, NZ, 1, mk, x0, b_scale=scale)
        nc = d.NativeComm(pkg, eng.ctx, d.SelfComm())
        it = d.NativeDistCGIterable(pkg, eng, nc, maxiter=10 ** 6)
        hist, iteration = [], 0
        while True:
            h = it.iterate_many(iteration, 1 if iteration < 2 else batch)
            if h.size == 0:
                break
            hist.append(h)
            iteration += h.size
        hist = np.concatenate(hist)
        xo, ho = oracle_history(orc, pkg, N, NZ, offsets, b, shape, x0)
        assert ho["iters"] > 10 and ho["isconverged"]
        assert hist.size == ho["iters"] and np.array_equal(hist, ho["resnorm"])
        assert np.array_equal(eng.solution(), xo)
        eng.close()
        nc.close()
    finally:
        L.mik_set_tuning(KN.NO_LOOKAHEAD, 0)
        L.mik_set_tuning(KN.CG_STEP, 0)


def test_native_transport_argument_checks(pkg):
    """no GPU needed: the new entry points reject NULL handles instead of crashing"""
    import ctypes as C
    L = pkg.lib()
    assert L.mik_cgd_set_halo_plan(None, 0, None, None, None, 0, None, None, None) == 1
    assert L.mik_cgd_set_comm(None, None) == 1
    assert L.mik_cgd_init(None, None, None) == 1
    n = C.c_int64()
    assert L.mik_cgd_iterate_many(None, 0, 1, None, C.byref(n)) == 1
    assert L.mik_cgd_group_init(None, 2, None, None) == 1
    assert L.mik_comm_create(None, None, 0, 1, None) == 1
    assert L.mik_comm_destroy(None) == 0 and L.mik_comm_allgather_sum(None, 0, 1, None) == 1
    # transport 3 (round 4)
    assert L.mik_comm_mailbox_export(None, None) == 1 and L.mik_comm_mailbox_connect(None, None) == 1 and L.mik_comm_mailbox_info(None, None, None) == 1
    assert L.mik_cgd_ghost_export(None, None) == 1 and L.mik_cgd_connect_ghosts(None, None, None, None) == 1
    # the device-driven links (round 5)
    h = C.c_void_p()
    assert L.mik_plink_create(None, 0, 0, 0, None, None, None, 0, None, None, None, C.byref(h)) == 1 and L.mik_plink_export(None, None) == 1
    assert L.mik_plink_connect(None, None, None, None) == 1 and L.mik_plink_info(None, None, None, None) == 1 and L.mik_plink_destroy(None) == 0
    assert L.mik_cgd_profile(None, 0, None, None) == 1
    g = C.c_int()
    assert L.mik_spmv_long_group(C.byref(g)) == 0 and g.value == 4 and L.mik_minres_proj_shape(None, None, None) == 1


def test_torch_slab_generator_and_localisation_equal_the_numpy_ones(pkg):
    """dist._laplace_rows_torch / localize_block_torch / interior_row_blocks on tensors (here on the CPU device; the bench runs
    them on the GPU) against the numpy reference functions: same CSR slab, same local column ids, same halo plan"""
    import torch
    dist = dist_mod(pkg)
    N, NZ, P = 10, 12, 4
    offsets = np.arange(P + 1, dtype=np.int64) * (N * N * (NZ // P))
    for r in range(P):
        _, ptr, idx, val = dist._laplace_rows(pkg, N, NZ, offsets[r], offsets[r + 1], np.float64)
        _, tptr, tidx, tval = dist._laplace_rows_torch(N, NZ, offsets[r], offsets[r + 1], np.float64, torch.device("cpu"))
        assert np.array_equal(ptr, tptr.numpy()) and np.array_equal(idx, tidx.numpy()) and np.array_equal(val, tval.numpy())
        li, plan = dist.localize_block(ptr, idx, offsets, r)
        tli, tplan = dist.localize_block_torch(tptr, tidx, offsets, r)
        assert np.array_equal(li, tli.numpy()) and np.array_equal(plan.ghost_gids, tplan.ghost_gids) and plan.recv == tplan.recv
        assert dist.interior_row_blocks(ptr, li, plan.n_loc) == dist.interior_row_blocks(tptr, tli, plan.n_loc)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_rccl_send_recv_call_path_on_one_device(pkg, ctx, dtype):
    """ncclSend / ncclRecv as libmik.so issues them (mik_comm_halo: group start, receives, sends, group end on the ctx stream),
    exercised on hardware with the only peer a single-GPU box offers -- the rank itself: two segments of a packed send buffer
    land at their offsets in the ghost region, element type and counts as in the halo plan of the row-partitioned solvers"""
    import ctypes as C
    d = dist_mod(pkg)
    boot = d.SelfComm()
    hctx = pkg.HipContext(0)
    nc = d.NativeComm(pkg, hctx, boot, force_rccl=True)
    assert nc.uses_rccl()
    rng = np.random.default_rng(17)
    send = rng.standard_normal(5000).astype(dtype)
    dsend = pkg.HipVector.from_numpy(send, hctx)
    ghost = pkg.HipVector(7000, dtype, hctx).fill_(0)
    peers = (C.c_int * 2)(0, 0)
    roff, rcnt = (C.c_int64 * 2)(100, 4000), (C.c_int64 * 2)(1500, 2500)
    soff, scnt = (C.c_int64 * 2)(0, 2000), (C.c_int64 * 2)(1500, 2500)
    code = pkg.lib().mik_comm_halo(nc.handle, 0 if dtype == np.float64 else 1, C.c_void_p(dsend.ptr), C.c_void_p(ghost.ptr), 2, peers, roff, rcnt, 2, peers,
                                   soff, scnt)
    pkg._lib.check(code, "mik_comm_halo", hctx.handle)
    got = ghost.to_numpy()
    want = np.zeros(7000, dtype)
    want[100:1600] = send[0:1500]
    want[4000:6500] = send[2000:4500]
    assert np.array_equal(got, want)
    nc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("transport,knob6", [("rccl", 0), ("rccl", 1), ("rccl+mailbox", 4), ("mailbox", 4), ("mailbox", 5)])
def test_full_rccl_step_on_one_device_through_a_periodic_self_halo(pkg, orc, ctx, overlap, transport, knob6, monkeypatch):
    """The complete library-driven step of the row-partitioned CG over REAL RCCL on one GPU: a grid that is periodic in z, cut
    into ONE slab, is its own neighbour -- the halo of the bottom / top planes is ncclSend / ncclRecv to the rank itself on the
    side stream, overlapped with the interior row-blocks, followed by the two ncclAllGather; every call mik_cgd_iterate_many makes
    at P = 8 is made here.  Bit-identical to the single-GPU iterable on the same periodic operator."""
    import scipy.sparse as sp
    monkeypatch.setenv("MIK_DIST_OVERLAP", "1" if overlap else "0")
    d = dist_mod(pkg)
    N, NZ = 16, 12
    n, plane = N * N * NZ, N * N
    _, ptr, idx, val = d._laplace_rows(pkg, N, NZ, 0, n, np.float64)
    S = sp.csr_matrix((val, idx, ptr), shape=(n, n)).tolil()
    for j in range(plane):                                   # periodic wrap in z
        S[j, j + plane * (NZ - 1)] = -1.0
        S[j + plane * (NZ - 1), j] = -1.0
    S = S.tocsr()
    S.sort_indices()
    # local block of the only rank: the wrap entries become halo columns (global order of a row's entries is kept)
    gi = S.indices.astype(np.int64)
    rows = np.repeat(np.arange(n), np.diff(S.indptr))
    wrap = np.abs(gi - rows) == plane * (NZ - 1)
    ghost_gids = np.unique(gi[wrap])
    li = gi.copy()
    li[wrap] = n + np.searchsorted(ghost_gids, gi[wrap])
    plan = d.HaloPlan(0, 1, n, ghost_gids)
    plan.recv = [(0, 0, int(ghost_gids.size))]
    plan.send = [(0, 0, int(ghost_gids.size))]
    plan.send_idx = ghost_gids.astype(np.int32)
    b = orc.hashed_rhs(n)
    eng = d.HipEngine(pkg, S.indptr.astype(np.int64), li, S.data.copy(), plan, b, abstol=0.0, reltol=1e-9, maxiter=10 ** 6)
    assert eng.overlap == overlap
    # transports: RCCL with flags (default) or events (MIK_KNOB_TRANSPORT bit 0) ordering the side stream; the two scalars through the
    # mailbox although the world is one rank (bit 2); no RCCL at all -- the halo pushed into the rank's own ghost region
    pkg.lib().mik_set_tuning(6, knob6)
    nc = d.NativeComm(pkg, eng.ctx, d.SelfComm(), force_rccl=transport != "mailbox", transport=transport)
    assert nc.uses_rccl() == (transport != "mailbox")
    it = d.NativeDistCGIterable(pkg, eng, nc, maxiter=10 ** 6)
    hist, iteration = [], 0
    while True:
        h = it.iterate_many(iteration, 1 if iteration < 4 else 17)
        if h.size == 0:
            break
        hist.append(h)
        iteration += h.size
    hist = np.concatenate(hist)
    C = S.tocsc()
    C.sort_indices()
    x, ch = pkg.cg(pkg.HipCSR(n, n, C.indptr.astype(np.int64), C.indices.astype(np.int64), C.data, index_base=0), pkg.HipVector.from_numpy(b),
                   reltol=1e-9, log=True)
    pkg.lib().mik_set_tuning(6, 0)
    assert ch.isconverged and np.array_equal(hist, ch["resnorm"]) and np.array_equal(eng.solution(), x.to_numpy())
    eng.close()
    nc.close()


def test_landing_targets_follow_the_receivers_plans(pkg):
    """no GPU needed: where a sender's segments land in the receivers' landing buffers (NativeComm._landing_targets) -- the offset of the MATCHING
    receive segment in the receiver's ghost region, matched in order when a rank receives several segments from the same peer (a z-periodic slab
    that is its own lower and upper neighbour), and a loud failure when the plans of two ranks disagree"""
    d = dist_mod(pkg)
    nc = d.NativeComm.__new__(d.NativeComm)
    # three z-slabs of a 4 x 4 x 12 grid: rank 1 receives 16 entries from rank 0 (ghost offset 0) and 16 from rank 2 (ghost offset 16)
    recv = {0: [(1, 0, 16)], 1: [(0, 0, 16), (2, 16, 16)], 2: [(1, 0, 16)]}
    send = {0: [(1, 0, 16)], 1: [(0, 0, 16), (2, 16, 16)], 2: [(1, 0, 16)]}
    info = [(b"", 16 if q != 1 else 32, recv[q]) for q in range(3)]
    want = {0: [0], 1: [0, 0], 2: [16]}            # rank 0 -> rank 1's ghost offset 0; rank 1 -> offset 0 of ranks 0 and 2; rank 2 -> rank 1's offset 16
    for q in range(3):
        nc.rank = q
        plan = d.HaloPlan(q, 3, 64, np.zeros(0, np.int64))
        plan.send = send[q]
        assert nc._landing_targets(plan, info).tolist() == want[q]
    # one slab, periodic in z: two segments to itself, landing at ghost offsets 0 and 16 in order
    nc.rank = 0
    plan = d.HaloPlan(0, 1, 64, np.zeros(0, np.int64))
    plan.send = [(0, 0, 16), (0, 16, 16)]
    assert nc._landing_targets(plan, [(b"", 32, [(0, 0, 16), (0, 16, 16)])]).tolist() == [0, 16]
    plan.send = [(0, 0, 16), (0, 16, 8)]
    with pytest.raises(RuntimeError, match="halo plans disagree"):
        nc._landing_targets(plan, [(b"", 32, [(0, 0, 16), (0, 16, 16)])])
