"""One process per DEVICE: the row-partitioned iterables across real GPUs (BASELINE.json configs[3]; VERDICT r4 #1d).

Switches on only where at least two devices are visible -- the single-GPU boxes of `gpurun` skip it, the first multi-GPU box runs
every transport of libmik.so (RCCL halo + RCCL scalars, RCCL halo + mailbox scalars, mailbox only: peer stores over xGMI into
fine-grained slots and into the receivers' halo landing buffers) and the row-partitioned GMRES at world = min(device_count, 8),
bit-exact against the partition-aware oracle: well-scaled and badly scaled right-hand sides (every step through the scaled norm
across the ranks), small slabs and slabs whose halos (2 x 2 MB per rank) do not fit next to anything in an XCD's L2 -- a stale
cached halo line would change the history."""
import importlib
import os
import sys

import numpy as np
import pytest

from conftest import ROOT


def _device_count():
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:      # noqa: BLE001
        return 0


# MIK_TEST_WORLD=P (development): run this file's workers as P processes on the devices there are (rank r on device r mod device_count) -- on a
# one-GPU box that exercises every line of the workers with the transports that allow ranks to share a device (mailbox, links, gloo callbacks);
# the RCCL cases are skipped there (RCCL refuses two ranks on one device).
FORCED = int(os.environ.get("MIK_TEST_WORLD", "0"))
NDEV = _device_count()
WORLD = FORCED if FORCED >= 2 else min(NDEV, 8)
SHARED = WORLD > NDEV                      # ranks share devices
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(WORLD < 2 or NDEV < 1, reason="needs at least two visible devices (one process per device)")]


def _dev(rank):
    return rank % max(NDEV, 1)


def _needs_distinct_devices(what):
    if SHARED:
        pytest.skip(f"{what}: RCCL refuses two ranks on one device (MIK_TEST_WORLD on a box with fewer devices than ranks)")


def _init(rank, world, port, backend="gloo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MIK_MAILBOX_TIMEOUT_MS="30000", HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as td
    import __graft_entry__ as graft
    pkg = graft.load_package()
    d = importlib.import_module(pkg.__name__ + ".dist")
    torch.cuda.set_device(_dev(rank))
    if backend == "nccl":                  # the Python-side exchanges themselves run over RCCL (device tensors)
        td.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", _dev(rank)))
    else:                                  # bootstrap only: ncclUniqueId, IPC handles, barriers
        td.init_process_group("gloo", rank=rank, world_size=world)
    return pkg, d, td


def _cg_worker(rank, world, port, N, nz, out_dir, transport, scale, batch, maxiter, knob6):
    pkg, d, td = _init(rank, world, port)
    boot = d.TorchComm()
    pkg.lib().mik_set_tuning(6, knob6)
    ptr, li, val, plan, b_loc, n, offsets = d.build_rank_problem(pkg, boot, N, nz_per_rank=nz, device=None)          # (host generation: PyTorch device generation stalls with several processes per GPU)
    eng = d.HipEngine(pkg, ptr, li, val, plan, b_loc * scale, abstol=0.0, reltol=1.5e-8, maxiter=maxiter, device=_dev(rank))
    nc = d.NativeComm(pkg, eng.ctx, boot, transport=transport)
    assert nc.uses_rccl() == (transport != "mailbox")
    it = d.NativeDistCGIterable(pkg, eng, nc, maxiter=maxiter)
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
    td.destroy_process_group()


def _oracle(orc, pkg, N, NZ, offsets, b, shape, maxiter=None):
    d = importlib.import_module(pkg.__name__ + ".dist")
    n, ptr, idx, val = d._laplace_rows(pkg, N, NZ, 0, N * N * NZ, np.float64)
    A = orc.CSC(n, ptr, idx, val, 0)
    orc.set_partition(offsets)
    try:
        return orc.cg(A, b, mode="tree", shape=shape, maxiter=maxiter)
    finally:
        orc.set_partition(None)


def _port(salt):
    return 28100 + (os.getpid() * 7 + salt * 13) % 1500


@pytest.mark.parametrize("transport", ["rccl", "rccl+mailbox", "mailbox"])
@pytest.mark.parametrize("scale,batch,knob6", [(1.0, 9, 0), (1e-140, 1, 0), (1e140, 7, 0), (1.0, 5, 1), (1.0, 5, 8)])
def test_cg_every_transport_across_devices_matches_partitioned_oracle(pkg, orc, ctx, tmp_path, transport, scale, batch, knob6):
    """mik_cgd_iterate_many, one rank per device: history and solution bit-exact against the oracle's cg! with the same partition;
    right-hand sides scaled by 1e-140 / 1e+140 freeze every step on the same total on every rank and finish it with the scaled
    norm across the ranks (tests/test_dist.py does this with all ranks on one GPU).  MIK_KNOB_TRANSPORT: 1 = the side stream ordered by
    events instead of mailbox flags, 8 = the step's scalars through the one-wave gather launches."""
    import torch.multiprocessing as mp
    if transport != "mailbox":
        _needs_distinct_devices(transport)
    N, nz = 16, 4
    mp.spawn(_cg_worker, args=(WORLD, _port(len(transport) + batch + knob6), N, nz, str(tmp_path), transport, scale, batch, 10 ** 6, knob6), nprocs=WORLD, join=True)
    hs = [np.load(tmp_path / f"hist{r}.npy") for r in range(WORLD)]
    assert all(np.array_equal(hs[0], h) for h in hs)
    offsets = np.load(tmp_path / "off0.npy")
    b = pkg.fixtures.hashed_rhs(N * N * nz * WORLD) * scale
    xo, ho = _oracle(orc, pkg, N, nz * WORLD, offsets, b, ctx.cg_shape(np.float64))
    assert ho["iters"] > 10 and ho["isconverged"]
    assert hs[0].size == ho["iters"] and np.array_equal(hs[0], ho["resnorm"])
    assert np.array_equal(np.concatenate([np.load(tmp_path / f"x{r}.npy") for r in range(WORLD)]), xo)


@pytest.mark.parametrize("transport", ["rccl", "rccl+mailbox", "mailbox"])
def test_cg_halos_larger_than_l2_across_devices(pkg, orc, ctx, tmp_path, transport):
    """two 512 x 512 planes of doubles per rank and step (4 MB: an XCD's whole L2) written by the NEIGHBOUR device, read by
    this rank's next SpMV launch -- 24 steps, batches of 5, bit-exact against the oracle (a halo line served stale from a cache
    would move the history at once)."""
    import torch.multiprocessing as mp
    if transport != "mailbox":
        _needs_distinct_devices(transport)
    N, nz, steps = 512, 2, 24
    mp.spawn(_cg_worker, args=(WORLD, _port(91 + len(transport)), N, nz, str(tmp_path), transport, 1.0, 5, steps, 0), nprocs=WORLD, join=True)
    hs = [np.load(tmp_path / f"hist{r}.npy") for r in range(WORLD)]
    assert all(np.array_equal(hs[0], h) for h in hs)
    offsets = np.load(tmp_path / "off0.npy")
    b = pkg.fixtures.hashed_rhs(N * N * nz * WORLD)
    xo, ho = _oracle(orc, pkg, N, nz * WORLD, offsets, b, ctx.cg_shape(np.float64), maxiter=steps)
    assert hs[0].size == steps == ho["iters"] and np.array_equal(hs[0], ho["resnorm"])
    assert np.array_equal(np.concatenate([np.load(tmp_path / f"x{r}.npy") for r in range(WORLD)]), xo)


def _gmres_worker(rank, world, port, out_dir, orth, scale, backend):
    pkg, d, td = _init(rank, world, port, "gloo" if backend == "link" else backend)
    import scipy.sparse as sp
    comm = d.TorchComm()
    n, colptr, rowval, nzval, b = pkg.fixtures.advection_dominated(12, 1000.0)
    S = sp.csc_matrix((nzval * scale, rowval - 1, colptr - 1), shape=(n, n)).tocsr()
    offsets = d.partition_rows(n, world)
    r0, r1 = int(offsets[rank]), int(offsets[rank + 1])
    blk = S[r0:r1]
    ptr, idx, val = blk.indptr.astype(np.int64), blk.indices.astype(np.int64), np.ascontiguousarray(blk.data)
    local_idx, plan = d.localize_block(ptr, idx, offsets, rank)
    d.complete_plan(plan, offsets, comm.all_gather_objects(plan.ghost_gids))
    M = {"mgs": pkg.ModifiedGramSchmidt(), "cgs": pkg.ClassicalGramSchmidt(), "dgks": pkg.DGKS()}[orth]
    it = d.DistGMRESIterable(pkg, comm, ptr, local_idx, val, plan, (b * scale)[r0:r1], n_global=n, restart=10, orth_meth=M, device=_dev(rank),
                             native="mailbox" if backend == "link" else None)
    hist = it.solve()
    np.save(os.path.join(out_dir, f"hist{rank}.npy"), hist)
    np.save(os.path.join(out_dir, f"x{rank}.npy"), it.solution())
    np.save(os.path.join(out_dir, f"off{rank}.npy"), offsets)
    comm.barrier()
    it.close()
    td.destroy_process_group()


@pytest.mark.parametrize("backend", ["link", "nccl", "gloo"])
@pytest.mark.parametrize("orth,scale", [("mgs", 1.0), ("cgs", 1.0), ("dgks", 1.0), ("mgs", 1e-160), ("cgs", 1e-160)])
def test_partitioned_gmres_across_devices_matches_partitioned_oracle(pkg, orc, ctx, tmp_path, orth, scale, backend):
    """mik_gmres_create_partitioned, one rank per device: "link" = the device-driven coupling (mik_partition.link: halo pushed over xGMI
    into the neighbours' landing buffers, every projection / norm summed over the ranks inside the finalising kernel through the
    peer-mapped mailboxes); "nccl" / "gloo" = halo and rank-ordered sums through the host's callbacks (RCCL device collectives, or gloo
    with host staging).  History and solution bit-exact against the oracle's gmres with the same partition; a system scaled by 1e-160
    sends every norm through the scaled pass across the ranks."""
    import torch.multiprocessing as mp
    if backend == "nccl":
        _needs_distinct_devices("nccl callbacks")
    mp.spawn(_gmres_worker, args=(WORLD, _port(len(orth) + {"nccl": 3, "gloo": 0, "link": 7}[backend] + (5 if scale != 1.0 else 0)), str(tmp_path), orth, scale, backend),
             nprocs=WORLD, join=True)
    A, _ = orc.advdiff(12, 1000.0)
    A = orc.CSC(A.n, A.colptr, A.rowval, A.nzval * scale, A.index_base)
    b = pkg.fixtures.advection_dominated(12, 1000.0)[4] * scale
    offsets = np.load(tmp_path / "off0.npy")
    orc.set_partition(offsets)
    try:
        xo, ho = orc.gmres(A, b, restart=10, orth_meth=orth, mode="tree", shape=ctx.reduce_shape(np.float64))
    finally:
        orc.set_partition(None)
    for r in range(WORLD):
        assert np.array_equal(np.load(tmp_path / f"hist{r}.npy"), ho["resnorm"])
    assert np.array_equal(np.concatenate([np.load(tmp_path / f"x{r}.npy") for r in range(WORLD)]), xo)


def test_bench_line_of_the_partitioned_run_is_contract_complete(tmp_path):
    """`python bench.py --gpus N` on this box: the line's roofline is the CSR kernel timed inside the partitioned loop, bytes per step over
    ms per step stay below the HBM peak, `value` is the global system's iteration rate, and every transport that came up is bit-identical
    to the partition-aware oracle on the small system."""
    import json
    import subprocess
    env = dict(os.environ, MIK_BENCH_MIN_SECONDS="0.05", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if SHARED:                             # all ranks on device 0, the one transport that allows it
        env.update(MIK_FORCE_DEVICE="0", MIK_NATIVE_TRANSPORTS="mailbox")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(WORLD), "--steps", "40", "--warmup", "5", "--cpu-iters", "3"],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == WORLD and line["value_is_contract"] and "k_spmv_rowgather" in line["roofline"]["kernel"]
    assert abs(line["value"] - 1e3 / line["ms_per_step"]) < 1e-6 * line["value"]
    assert line["value_bytes_per_step_per_gpu"] / (line["ms_per_step"] * 1e-3) / 1e9 <= 8000.0
    assert line["roofline"]["avg_launch_ms"] <= line["ms_per_step"] and 0.0 < line["roofline"]["frac"] <= 1.0
    assert line["parity_vs_oracle"]["bit_identical"], line["parity_vs_oracle"]
    assert line["contract_csr_loop"]["first_residuals_equal_the_default_layout_bit_for_bit"]
