"""Per-inner-iteration cost of the row-partitioned GMRES iterable, all ranks on the box's ONE GPU:
   world 1: the fused single-GPU iterable | the partitioned handle with host callbacks | the partitioned handle with a device-driven link
   world 2: two processes, device-driven link (mailbox slots + landing buffers over HIP IPC) | host callbacks staged through gloo
configs[2] (advection_dominated N = 50, restart 30) and a 128^3 Laplacian, ModifiedGramSchmidt and ClassicalGramSchmidt; the loop of gmres!
inside the library (mik_gmres_iterate_many).  Two ranks that share a GPU also share its CUs: each runs half the rows, so "within 15 % of
world 1" is about the exchanges' cost, not a speed-up.        python scripts/gmres_part_bench.py"""
import json
import os
import sys
import time
from importlib import import_module

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def problem(pkg, case):
    if case == "advdiff50":
        n, colptr, rowval, nzval, b = pkg.fixtures.advection_dominated(50, 1000.0)
    else:
        n, colptr, rowval, nzval = pkg.fixtures.laplace_matrix(128, 3)
        b = pkg.fixtures.hashed_rhs(n)
    return n, colptr, rowval, nzval, b


def worker(rank, world, port, case, mode, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import scipy.sparse as sp
    import torch
    import torch.distributed as td
    import __graft_entry__ as graft
    pkg = graft.load_package()
    dist = import_module(pkg.__name__ + ".dist")
    torch.cuda.set_device(0)
    if world > 1:
        td.init_process_group("gloo", rank=rank, world_size=world)
        comm = dist.TorchComm()
    else:
        comm = dist.SelfComm()
    n, colptr, rowval, nzval, b = problem(pkg, case)
    restart, iters = 30, 90
    res = {}
    for mname, M in (("mgs", pkg.ModifiedGramSchmidt()), ("cgs", pkg.ClassicalGramSchmidt()), ("dgks", pkg.DGKS())):
        if mode == "fused":
            dA = pkg.HipCSR(n, n, colptr, rowval, nzval, index_base=1)
            db = pkg.HipVector.from_numpy(b)
            it = pkg.gmres_iterable_(pkg.zerox(dA, db), dA, db, restart=restart, reltol=0.0, maxiter=10 ** 9, initially_zero=True, orth_meth=M)
            many = lambda k, m: it.iterate_many(k, m)
        else:
            S = sp.csc_matrix((nzval, rowval - 1, colptr - 1), shape=(n, n)).tocsr()
            offsets = dist.partition_rows(n, world)
            r0, r1 = int(offsets[rank]), int(offsets[rank + 1])
            blk = S[r0:r1]
            ptr, idx, val = blk.indptr.astype(np.int64), blk.indices.astype(np.int64), np.ascontiguousarray(blk.data)
            li, plan = dist.localize_block(ptr, idx, offsets, rank)
            dist.complete_plan(plan, offsets, comm.all_gather_objects(plan.ghost_gids))
            it = dist.DistGMRESIterable(pkg, comm, ptr, li, val, plan, b[r0:r1], restart=restart, reltol=0.0, maxiter=10 ** 9, orth_meth=M, n_global=n,
                                        native="mailbox" if mode == "link" else None)
            many = lambda k, m: it.iterate_many(k, m)
        h = many(0, restart)                         # one warm cycle
        best = None
        k = restart
        for _rep in range(3):
            comm.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            h = many(k, iters)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            k += iters
            best = dt if best is None else min(best, dt)
        res[mname] = {"us_per_inner_iteration": best / iters * 1e6, "last_residual": float(h[-1]).hex()}
        if hasattr(it, "close"):
            it.close()
    if rank == 0:
        json.dump(res, open(out_path, "w"))
    if world > 1:
        comm.barrier()
        td.destroy_process_group()


def main():
    import tempfile
    import torch.multiprocessing as mp
    out = {}
    tmp = tempfile.mkdtemp()
    port = 29650
    for case in ("advdiff50", "laplace128"):
        out[case] = {}
        for world, mode in ((1, "fused"), (1, "callbacks"), (1, "link"), (2, "link"), (2, "callbacks")):
            path = os.path.join(tmp, f"{case}_{world}_{mode}.json")
            port += 1
            try:
                mp.spawn(worker, args=(world, port, case, mode, path), nprocs=world, join=True)
                out[case][f"world{world}_{mode}"] = json.load(open(path))
            except Exception as e:     # noqa: BLE001
                out[case][f"world{world}_{mode}"] = {"error": str(e)[:300]}
        o = out[case]
        for m in ("mgs", "cgs", "dgks"):
            try:
                o[f"{m}_world2_link_over_world1_link"] = o["world2_link"][m]["us_per_inner_iteration"] / o["world1_link"][m]["us_per_inner_iteration"]
                o[f"{m}_same_bits_link_vs_callbacks_world2"] = o["world2_link"][m]["last_residual"] == o["world2_callbacks"][m]["last_residual"]
                o[f"{m}_same_bits_world1_all"] = len({o[k][m]["last_residual"] for k in ("world1_fused", "world1_callbacks", "world1_link")}) == 1
            except Exception:          # noqa: BLE001
                pass
    print(json.dumps(out))


if __name__ == "__main__":
    main()
