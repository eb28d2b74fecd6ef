"""Row-partitioned GMRES (mik_gmres_create_partitioned + dist.DistGMRESIterable) against the oracle's
gmres with the same partition-ordered sums: bit-exact residual history, solution and counters.

On one GPU the P ranks are P host threads (dist.ThreadComm) or P processes with gloo-staged exchanges;
the RCCL path uses the same callbacks with device tensors (dist.TorchComm)."""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def csr_block(S, r0, r1):
    blk = S[r0:r1]
    return blk.indptr.astype(np.int64), blk.indices.astype(np.int64), np.ascontiguousarray(blk.data)


def run_threads(pkg, dist, S, b, P, *, x0=None, pl=None, pr=None, offsets=None, **kw):
    n = S.shape[0]
    offsets = dist.partition_rows(n, P) if offsets is None else np.asarray(offsets, np.int64)
    comms = dist.ThreadComm.world(P)
    out, errs = [None] * P, [None] * P

    def worker(r):
        try:
            r0, r1 = int(offsets[r]), int(offsets[r + 1])
            ptr, idx, val = csr_block(S, r0, r1)
            local_idx, plan = dist.localize_block(ptr, idx, offsets, r)
            dist.complete_plan(plan, offsets, comms[r].all_gather_objects(plan.ghost_gids))
            it = dist.DistGMRESIterable(pkg, comms[r], ptr, local_idx, val, plan, b[r0:r1], None if x0 is None else x0[r0:r1],
                                        pl_diag=None if pl is None else pl[r0:r1], pr_diag=None if pr is None else pr[r0:r1],
                                        n_global=n, **kw)
            hist = it.solve()
            out[r] = dict(hist=hist, x=it.solution(), mvps=it.mv_products, conv=it.converged())
            it.close()
        except BaseException as e:          # release the other ranks from their barrier
            errs[r] = e
            comms[r].shared["barrier"].abort()

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(P)]
    [t.start() for t in ts]
    [t.join(timeout=120) for t in ts]
    first = next((e for e in errs if e is not None and not isinstance(e, threading.BrokenBarrierError)), None) or next((e for e in errs if e), None)
    if first is not None:
        raise first
    assert all(o is not None for o in out), "a rank thread did not finish"
    return offsets, out


@pytest.mark.parametrize("P", [1, 2, 3])
@pytest.mark.parametrize("orth", ["mgs", "cgs", "dgks"])
def test_partitioned_gmres_bit_exact_vs_partitioned_oracle(pkg, orc, ctx, dist, P, orth):
    A, b = orc.advdiff(12, 1000.0)
    S = A.to_scipy().tocsr()
    M = {"mgs": pkg.ModifiedGramSchmidt(), "cgs": pkg.ClassicalGramSchmidt(), "dgks": pkg.DGKS()}[orth]
    offsets, out = run_threads(pkg, dist, S, b, P, restart=10, orth_meth=M)
    orc.set_partition(offsets)
    try:
        xo, ho = orc.gmres(A, b, restart=10, orth_meth=orth, mode="tree", shape=ctx.reduce_shape(np.float64))
    finally:
        orc.set_partition(None)
    for o in out:
        assert np.array_equal(o["hist"], ho["resnorm"]) and o["mvps"] == ho["mvps"] and o["conv"] == ho["isconverged"]
    assert np.array_equal(np.concatenate([o["x"] for o in out]), xo)
    assert np.linalg.norm(S @ xo - b) / np.linalg.norm(b) <= 2e-8


@pytest.mark.parametrize("orth", ["mgs", "cgs", "dgks"])
def test_partitioned_gmres_on_a_badly_scaled_system(pkg, orc, ctx, dist, orth):
    """operator and rhs scaled by 1e-160: beta and every Gram-Schmidt norm leave the safe range; the ranks obtain max |x_i|
    through the sum callback (own slot of a zero vector), rescale by the common power of two and agree with the partition-aware
    oracle bit for bit -- MIK_ERR_RANGE no longer separates the partitioned solve from the single-GPU one"""
    A, b = orc.advdiff(9, 200.0)
    s = 1e-160
    As = orc.CSC(A.n, A.colptr, A.rowval, A.nzval * s, A.index_base)
    bs = b * s
    S = As.to_scipy().tocsr()
    M = {"mgs": pkg.ModifiedGramSchmidt(), "cgs": pkg.ClassicalGramSchmidt(), "dgks": pkg.DGKS()}[orth]
    offsets, out = run_threads(pkg, dist, S, bs, 3, restart=12, orth_meth=M)
    orc.set_partition(offsets)
    try:
        xo, ho = orc.gmres(As, bs, restart=12, orth_meth=orth, mode="tree", shape=ctx.reduce_shape(np.float64))
    finally:
        orc.set_partition(None)
    assert ho["isconverged"] and ho["iters"] > 5
    for o in out:
        assert np.array_equal(o["hist"], ho["resnorm"]) and o["mvps"] == ho["mvps"] and o["conv"] == ho["isconverged"]
    assert np.array_equal(np.concatenate([o["x"] for o in out]), xo)


def test_partitioned_gmres_preconditioned_nonzero_start_uneven_blocks(pkg, orc, ctx, dist):
    A, b = orc.advdiff(10, 500.0)
    S = A.to_scipy().tocsr()
    n = S.shape[0]
    d = S.diagonal()
    pl, pr = np.sqrt(np.abs(d)), np.sign(d) * np.sqrt(np.abs(d))          # Pl * Pr = diag(A)
    x0 = np.cos(np.arange(n) * 0.37)
    offsets = np.array([0, 137, 138, 700, n])          # ragged blocks, one of a single row
    offsets, out = run_threads(pkg, dist, S, b, 4, x0=x0, pl=pl, pr=pr, offsets=offsets, restart=7, maxiter=40)
    orc.set_partition(offsets)
    try:
        xo, ho = orc.gmres(A, b, x0, restart=7, maxiter=40, mode="tree", shape=ctx.reduce_shape(np.float64), pl_diag=pl, pr_diag=pr)
    finally:
        orc.set_partition(None)
    for o in out:
        assert np.array_equal(o["hist"], ho["resnorm"]) and o["mvps"] == ho["mvps"]
    assert np.array_equal(np.concatenate([o["x"] for o in out]), xo)


def test_partitioned_gmres_fp32(pkg, orc, ctx, dist):
    A64, b64 = orc.advdiff(10, 100.0)
    A, b = orc.CSC(A64.n, A64.colptr, A64.rowval, A64.nzval.astype(np.float32), A64.index_base), b64.astype(np.float32)
    S = A.to_scipy().tocsr()
    offsets, out = run_threads(pkg, dist, S, b, 2, restart=15, orth_meth=pkg.ClassicalGramSchmidt())
    orc.set_partition(offsets)
    try:
        xo, ho = orc.gmres(A, b, restart=15, orth_meth="cgs", mode="tree", shape=ctx.reduce_shape(np.float32))
    finally:
        orc.set_partition(None)
    assert np.array_equal(out[0]["hist"], ho["resnorm"]) and np.array_equal(out[1]["hist"], ho["resnorm"])
    assert np.array_equal(np.concatenate([o["x"] for o in out]), xo)


def test_one_rank_partition_equals_the_plain_iterable(pkg, orc, ctx, dist):
    A, b = orc.advdiff(12, 1000.0)
    S = A.to_scipy().tocsr()
    _, out = run_threads(pkg, dist, S, b, 1, restart=10)
    dA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval, index_base=A.index_base)
    x, ch = pkg.gmres(dA, pkg.HipVector.from_numpy(b), restart=10, log=True)
    assert np.array_equal(out[0]["hist"], ch["resnorm"]) and np.array_equal(out[0]["x"], x.to_numpy())


def test_callback_failure_surfaces_as_an_error(pkg, orc, ctx, dist):
    A, b = orc.advdiff(6, 10.0)
    S = A.to_scipy().tocsr()

    class Boom(dist.SelfComm):
        calls = 0

        def all_gather_host(self, values):
            Boom.calls += 1
            if Boom.calls > 3:
                raise RuntimeError("link down")
            return super().all_gather_host(values)

    n = S.shape[0]
    offsets = np.array([0, n])
    ptr, idx, val = csr_block(S, 0, n)
    local_idx, plan = dist.localize_block(ptr, idx, offsets, 0)
    dist.complete_plan(plan, offsets, [plan.ghost_gids])
    it = dist.DistGMRESIterable(pkg, Boom(), ptr, local_idx, val, plan, b, n_global=n, restart=5)
    with pytest.raises(RuntimeError, match="link down"):
        list(it)
    it.close()


def _proc_worker(rank, world, port, out_dir, backend="gloo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if backend == "nccl":
        os.environ["MIK_DIST_FORCE_COLLECTIVES"] = "1"      # a world of one still issues the RCCL calls
    import torch
    import torch.distributed as td
    import __graft_entry__ as graft
    from importlib import import_module
    pkg = graft.load_package()
    dist = import_module(pkg.__name__ + ".dist")
    torch.cuda.set_device(0)
    if backend == "nccl":
        td.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        td.init_process_group("gloo", rank=rank, world_size=world)
    comm = dist.TorchComm()
    n, colptr, rowval, nzval, b = pkg.fixtures.advection_dominated(10, 1000.0)
    import scipy.sparse as sp
    S = sp.csc_matrix((nzval, rowval - 1, colptr - 1), shape=(n, n)).tocsr()
    offsets = dist.partition_rows(n, world)
    r0, r1 = int(offsets[rank]), int(offsets[rank + 1])
    ptr, idx, val = csr_block(S, r0, r1)
    local_idx, plan = dist.localize_block(ptr, idx, offsets, rank)
    dist.complete_plan(plan, offsets, comm.all_gather_objects(plan.ghost_gids))
    it = dist.DistGMRESIterable(pkg, comm, ptr, local_idx, val, plan, b[r0:r1], n_global=n, restart=8, orth_meth=pkg.DGKS())
    hist = it.solve()
    np.save(os.path.join(out_dir, f"hist{rank}.npy"), hist)
    np.save(os.path.join(out_dir, f"x{rank}.npy"), it.solution())
    np.save(os.path.join(out_dir, f"off{rank}.npy"), offsets)
    it.close()
    td.destroy_process_group()


@pytest.mark.parametrize("world,backend", [(2, "gloo"), (1, "nccl")])
def test_process_ranks_match_partitioned_oracle(pkg, orc, ctx, tmp_path, world, backend):
    """2 processes on one GPU with gloo-staged exchanges; and the RCCL call path (device all-gather of the
    partial sums, stream-ordered send/recv plumbing) in a world of one -- RCCL refuses two ranks per device."""
    import torch.multiprocessing as mp
    port = 29900 + os.getpid() % 90 + world
    mp.spawn(_proc_worker, args=(world, port, str(tmp_path), backend), nprocs=world, join=True)
    A, _ = orc.advdiff(10, 1000.0)
    b = pkg.fixtures.advection_dominated(10, 1000.0)[4]   # the workers' rhs (the product-side fixture)
    offsets = np.load(tmp_path / "off0.npy")
    orc.set_partition(offsets)
    try:
        xo, ho = orc.gmres(A, b, restart=8, orth_meth="dgks", mode="tree", shape=ctx.reduce_shape(np.float64))
    finally:
        orc.set_partition(None)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"hist{r}.npy"), ho["resnorm"])
    assert np.array_equal(np.concatenate([np.load(tmp_path / f"x{r}.npy") for r in range(world)]), xo)


# ------------------------------------------------------------------------------------------------
# device-driven coupling (mik_partition.link, include/mik.h "Transport 3"): no host callback, no host round trip inside an Arnoldi column
# ------------------------------------------------------------------------------------------------
def _link_worker(rank, world, port, out_dir, orth, scale, dtype_name, restart, batch, gs=0):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), MIK_MAILBOX_TIMEOUT_MS="20000")
    import torch
    import torch.distributed as td
    import scipy.sparse as sp
    import __graft_entry__ as graft
    from importlib import import_module
    pkg = graft.load_package()
    dist = import_module(pkg.__name__ + ".dist")
    torch.cuda.set_device(0)
    td.init_process_group("gloo", rank=rank, world_size=world)          # carries the IPC handles, nothing else
    comm = dist.TorchComm()
    pkg.lib().mik_set_tuning(5, gs)                                       # MIK_KNOB_GS: 0 = single launch, 2 = launch-lean chain, 1 = general chain
    dtype = np.dtype(dtype_name)
    n, colptr, rowval, nzval, b = pkg.fixtures.advection_dominated(11, 700.0)
    S = sp.csc_matrix(((nzval * scale).astype(dtype), rowval - 1, colptr - 1), shape=(n, n)).tocsr()
    b = (b * scale).astype(dtype)
    offsets = dist.partition_rows(n, world)
    r0, r1 = int(offsets[rank]), int(offsets[rank + 1])
    ptr, idx, val = csr_block(S, r0, r1)
    local_idx, plan = dist.localize_block(ptr, idx, offsets, rank)
    dist.complete_plan(plan, offsets, comm.all_gather_objects(plan.ghost_gids))
    M = {"mgs": pkg.ModifiedGramSchmidt(), "cgs": pkg.ClassicalGramSchmidt(), "dgks": pkg.DGKS()}[orth]
    calls = {"halo": 0, "reduce": 0}
    it = dist.DistGMRESIterable(pkg, comm, ptr, local_idx, val, plan, b[r0:r1], n_global=n, restart=restart, orth_meth=M, native="mailbox")
    orig_halo, orig_reduce = it.links.halo, it.links.reduce
    it.links.halo = lambda: (calls.__setitem__("halo", calls["halo"] + 1), orig_halo())[1]
    it.links.reduce = lambda v: (calls.__setitem__("reduce", calls["reduce"] + 1), orig_reduce(v))[1]
    if batch:
        hist, k = [], 0
        while True:
            h = it.iterate_many(k, 1 if k < 2 else batch)
            if h.size == 0:
                break
            hist.extend(h.tolist())
            k += h.size
        hist = np.asarray(hist)
    else:
        hist = it.solve()
    assert calls == {"halo": 0, "reduce": 0}, calls                     # the library never came back to the host for an exchange
    np.save(os.path.join(out_dir, f"hist{rank}.npy"), hist)
    np.save(os.path.join(out_dir, f"x{rank}.npy"), it.solution())
    np.save(os.path.join(out_dir, f"off{rank}.npy"), offsets)
    np.save(os.path.join(out_dir, f"mv{rank}.npy"), np.array([it.mv_products, int(it.converged())]))
    comm.barrier()
    it.close()
    td.destroy_process_group()


@pytest.mark.parametrize("world,orth,scale,dtype_name,restart,batch,gs", [
    (2, "mgs", 1.0, "float64", 10, 0, 0), (3, "mgs", 1.0, "float64", 10, 7, 0), (2, "cgs", 1.0, "float64", 10, 0, 0), (3, "dgks", 1.0, "float64", 8, 0, 0),
    (2, "mgs", 1.0, "float32", 10, 5, 0), (2, "cgs", 1.0, "float32", 12, 0, 0), (2, "dgks", 1.0, "float32", 10, 0, 0),
    (2, "mgs", 1e-160, "float64", 10, 0, 0), (3, "cgs", 1e-160, "float64", 10, 4, 0), (2, "dgks", 1e-160, "float64", 10, 0, 0), (2, "mgs", 1e-22, "float32", 10, 0, 0),
    (1, "mgs", 1.0, "float64", 10, 0, 0),
    (2, "mgs", 1.0, "float64", 10, 0, 1), (3, "mgs", 1e-160, "float64", 10, 3, 1), (2, "mgs", 1.0, "float32", 10, 0, 1),
    (2, "mgs", 1.0, "float64", 10, 0, 2), (3, "mgs", 1e-160, "float64", 10, 3, 2), (2, "mgs", 1.0, "float32", 10, 6, 2), (3, "mgs", 1.0, "float64", 10, 0, 2),
    (2, "cgs", 1.0, "float64", 10, 0, 2), (3, "cgs", 1e-160, "float64", 10, 4, 1), (2, "cgs", 1.0, "float32", 12, 5, 2),
    (2, "dgks", 1.0, "float64", 10, 0, 1), (3, "dgks", 1e-160, "float64", 10, 4, 1), (2, "dgks", 1.0, "float32", 10, 3, 3), (2, "dgks", 1.0, "float64", 30, 0, 0),
    (2, "mgs", 1.0, "float64", 70, 0, 0), (2, "cgs", 1.0, "float64", 70, 0, 0), (2, "dgks", 1.0, "float64", 70, 0, 0)])     # restart > 62: more passes than vector slots -- the chains
def test_device_driven_partitioned_gmres_ranks_in_processes_on_one_gpu(pkg, orc, ctx, tmp_path, world, orth, scale, dtype_name, restart, batch, gs):
    """mik_gmres_create_partitioned with mik_partition.link -- halo pushed into the neighbours' landing buffers, every
    projection and norm summed over the ranks INSIDE the kernel that finalises it (mailbox slots, rank order), coefficients read from device
    memory by the next sweep: no host callback at all (counted), one host wait per inner iteration.  2 and 3 ranks as separate processes
    on the box's one GPU over HIP IPC; history, solution, mv_products and isconverged bit-exact against the partition-aware oracle --
    ModifiedGramSchmidt, ClassicalGramSchmidt, DGKS; fp64 and fp32; systems scaled by 1e-160 (fp32: 1e-22), which send every norm through the
    scaled pass across the ranks; per-step calls and mik_gmres_iterate_many batches.  Modified Gram-Schmidt (gs = MIK_KNOB_GS): 0 = the SINGLE-LAUNCH
    kernel with the exchange inside (k_mgs_fused<..., MailSumPass>: the total of every pass posted to the peers by workgroup 0, collected from
    the mailbox by every workgroup, one vector slot per pass; up to 2048 segments per rank, restart <= 62, one Arnoldi column enqueued ahead of
    the host), 2 = the launch-lean chain (every pass finalises AND exchanges the previous reduction itself: k + 2 launches, up to 256 segments
    per rank), 1 = the general chain with its finalise-and-exchange launches (what larger slabs run).  ClassicalGramSchmidt: 0 = k_cgs_fused with the
    exchange inside (the reducer workgroup of every column swaps the rank's total for the sum over the ranks), 1 / 2 = batched dot + one vector
    exchange (k_mail_sum_vec) + axpy sweep.  DGKS: 0 = k_cgs_fused<DGKS> with the exchange inside (as many rounds per launch as the 64 vector slots
    hold: 64 / (restart + 1), at most 3), 3 = one round per launch, then the link's host loop, 1 / 2 = the chain with its loop on the host."""
    import torch.multiprocessing as mp
    port = 29100 + (os.getpid() * 3 + world * 17 + len(orth) * 5 + restart + batch + (40 if scale != 1.0 else 0) + (80 if dtype_name == "float32" else 0) + 160 * gs) % 700
    mp.spawn(_link_worker, args=(world, port, str(tmp_path), orth, scale, dtype_name, restart, batch, gs), nprocs=world, join=True)
    dtype = np.dtype(dtype_name)
    A64, b64 = orc.advdiff(11, 700.0)
    A = orc.CSC(A64.n, A64.colptr, A64.rowval, (A64.nzval * scale).astype(dtype), A64.index_base)
    b = (pkg.fixtures.advection_dominated(11, 700.0)[4] * scale).astype(dtype)
    offsets = np.load(tmp_path / "off0.npy")
    orc.set_partition(offsets)
    try:
        xo, ho = orc.gmres(A, b, restart=restart, orth_meth=orth, mode="tree", shape=ctx.reduce_shape(dtype))
    finally:
        orc.set_partition(None)
    assert ho["iters"] > min(restart, 40)                                  # (restart = 10: at least one restart cycle -- update_solution!, init!)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"hist{r}.npy"), ho["resnorm"]), r
        assert tuple(np.load(tmp_path / f"mv{r}.npy")) == (ho["mvps"], int(ho["isconverged"]))
    assert np.array_equal(np.concatenate([np.load(tmp_path / f"x{r}.npy") for r in range(world)]), xo)


def _reorth_system(dtype):
    import scipy.sparse as sp
    n = 3000
    S = (sp.identity(n) + 1e-4 * sp.random(n, n, density=0.002, random_state=3)).tocsc()
    S.sort_indices()
    return n, S.astype(dtype), np.random.default_rng(21).standard_normal(n).astype(dtype)


def _link_reorth_worker(rank, world, port, out_dir, dtype_name, gs):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), MIK_MAILBOX_TIMEOUT_MS="20000")
    import torch
    import torch.distributed as td
    import __graft_entry__ as graft
    from importlib import import_module
    pkg = graft.load_package()
    dist = import_module(pkg.__name__ + ".dist")
    torch.cuda.set_device(0)
    td.init_process_group("gloo", rank=rank, world_size=world)
    comm = dist.TorchComm()
    pkg.lib().mik_set_tuning(5, gs)                                       # MIK_KNOB_GS: 0 = three rounds inside the launch, 3 = one (then handed back), 1 = the chain
    n, S, b = _reorth_system(np.dtype(dtype_name))
    S = S.tocsr()
    offsets = dist.partition_rows(n, world)
    r0, r1 = int(offsets[rank]), int(offsets[rank + 1])
    ptr, idx, val = csr_block(S, r0, r1)
    local_idx, plan = dist.localize_block(ptr, idx, offsets, rank)
    dist.complete_plan(plan, offsets, comm.all_gather_objects(plan.ghost_gids))
    it = dist.DistGMRESIterable(pkg, comm, ptr, local_idx, val, plan, b[r0:r1], n_global=n, restart=12, maxiter=20, reltol=0.0, orth_meth=pkg.DGKS(), native="mailbox")
    hist = it.solve()
    np.save(os.path.join(out_dir, f"hist{rank}.npy"), hist)
    np.save(os.path.join(out_dir, f"x{rank}.npy"), it.solution())
    np.save(os.path.join(out_dir, f"off{rank}.npy"), offsets)
    comm.barrier()
    it.close()
    td.destroy_process_group()


@pytest.mark.parametrize("world,dtype_name,gs", [(2, "float64", 0), (3, "float64", 3), (2, "float64", 1), (2, "float32", 0), (2, "float32", 3), (3, "float32", 1)])
def test_device_driven_partitioned_dgks_reorthogonalises_across_the_ranks(pkg, orc, ctx, tmp_path, world, dtype_name, gs):
    """A = I + a tiny perturbation (the system of test_gmres_dgks_reorthogonalisation_inside_the_single_launch_kernel) cut into 2 / 3 row blocks, ranks
    as processes on one GPU: the DGKS condition (src/orthogonalize.jl:26) holds, so the loop runs -- inside k_cgs_fused<DGKS, MailSumPass> with
    every round's column totals and norm summed over the ranks in the launch (gs = 0), handed back after one round to the link's host loop
    (gs = 3: every rank takes the hand-back together, the totals being identical), and as the chain (gs = 1).  Bit-exact against the
    partition-aware oracle, which differs from its own CGS run (the loop really ran)."""
    import torch.multiprocessing as mp
    port = 29100 + (os.getpid() * 3 + world * 19 + 7 * gs + (50 if dtype_name == "float32" else 0) + 333) % 700
    mp.spawn(_link_reorth_worker, args=(world, port, str(tmp_path), dtype_name, gs), nprocs=world, join=True)
    dtype = np.dtype(dtype_name)
    n, S, b = _reorth_system(dtype)
    A = orc.CSC.from_scipy(S).astype(dtype)
    orc.set_partition(np.load(tmp_path / "off0.npy"))
    try:
        xo, ho = orc.gmres(A, b, restart=12, orth_meth="dgks", mode="tree", shape=ctx.reduce_shape(dtype), maxiter=20, reltol=0.0)
        xc, hc = orc.gmres(A, b, restart=12, orth_meth="cgs", mode="tree", shape=ctx.reduce_shape(dtype), maxiter=20, reltol=0.0)
    finally:
        orc.set_partition(None)
    assert not np.array_equal(ho["resnorm"], hc["resnorm"]) or not np.array_equal(xo, xc)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"hist{r}.npy"), ho["resnorm"]), r
    assert np.array_equal(np.concatenate([np.load(tmp_path / f"x{r}.npy") for r in range(world)]), xo)


def _link_precond_worker(rank, world, port, out_dir, orth):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), MIK_MAILBOX_TIMEOUT_MS="20000")
    import torch
    import torch.distributed as td
    import scipy.sparse as sp
    import __graft_entry__ as graft
    from importlib import import_module
    pkg = graft.load_package()
    dist = import_module(pkg.__name__ + ".dist")
    torch.cuda.set_device(0)
    td.init_process_group("gloo", rank=rank, world_size=world)
    comm = dist.TorchComm()
    n, colptr, rowval, nzval, b = pkg.fixtures.advection_dominated(10, 400.0)
    S = sp.csc_matrix((nzval, rowval - 1, colptr - 1), shape=(n, n)).tocsr()
    pl = np.abs(S.diagonal()) ** 0.5
    pr = 1.0 + 0.5 * np.cos(np.arange(n))
    x0 = np.random.default_rng(4).standard_normal(n)
    offsets = np.array([0, 517, n]) if world == 2 else dist.partition_rows(n, world)      # an uneven cut
    r0, r1 = int(offsets[rank]), int(offsets[rank + 1])
    ptr, idx, val = csr_block(S, r0, r1)
    local_idx, plan = dist.localize_block(ptr, idx, offsets, rank)
    dist.complete_plan(plan, offsets, comm.all_gather_objects(plan.ghost_gids))
    M = {"mgs": pkg.ModifiedGramSchmidt(), "cgs": pkg.ClassicalGramSchmidt(), "dgks": pkg.DGKS()}[orth]
    it = dist.DistGMRESIterable(pkg, comm, ptr, local_idx, val, plan, b[r0:r1], x0[r0:r1], n_global=n, restart=7, maxiter=40, orth_meth=M,
                                pl_diag=pl[r0:r1], pr_diag=pr[r0:r1], native="mailbox")
    hist = it.solve()
    np.save(os.path.join(out_dir, f"hist{rank}.npy"), hist)
    np.save(os.path.join(out_dir, f"x{rank}.npy"), it.solution())
    np.save(os.path.join(out_dir, f"off{rank}.npy"), offsets)
    np.save(os.path.join(out_dir, f"mv{rank}.npy"), np.array([it.mv_products, int(it.converged())]))
    comm.barrier()
    it.close()
    td.destroy_process_group()


@pytest.mark.parametrize("world,orth", [(2, "mgs"), (3, "cgs"), (2, "dgks")])
def test_device_driven_partitioned_gmres_with_pl_pr_and_a_starting_guess(pkg, orc, ctx, tmp_path, world, orth):
    """SURVEY 8f rank 2 over the device-driven link: all three expand! methods' preconditioned form (ldiv!(Pr, .), mul!, ldiv!(Pl, .) on the local
    rows, src/gmres.jl:297-304), update_solution! with Pr (:278-283), and x0 != 0 -- init! then needs a halo exchange and a reduction before
    the first Arnoldi column (src/gmres.jl:241-253); maxiter ends the solve inside a cycle (the mv_products quirk of :101).  Uneven row cut."""
    import torch.multiprocessing as mp
    port = 29050 + (os.getpid() * 5 + world * 11 + len(orth)) % 40
    mp.spawn(_link_precond_worker, args=(world, port, str(tmp_path), orth), nprocs=world, join=True)
    A, _ = orc.advdiff(10, 400.0)
    n = A.n
    b = pkg.fixtures.advection_dominated(10, 400.0)[4]
    S = A.to_scipy().tocsr()
    pl = np.abs(S.diagonal()) ** 0.5
    pr = 1.0 + 0.5 * np.cos(np.arange(n))
    x0 = np.random.default_rng(4).standard_normal(n)
    offsets = np.load(tmp_path / "off0.npy")
    orc.set_partition(offsets)
    try:
        xo, ho = orc.gmres(A, b, x0, restart=7, maxiter=40, orth_meth=orth, mode="tree", shape=ctx.reduce_shape(np.float64), pl_diag=pl, pr_diag=pr)
    finally:
        orc.set_partition(None)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"hist{r}.npy"), ho["resnorm"]), r
        assert tuple(np.load(tmp_path / f"mv{r}.npy")) == (ho["mvps"], int(ho["isconverged"]))
    assert np.array_equal(np.concatenate([np.load(tmp_path / f"x{r}.npy") for r in range(world)]), xo)
