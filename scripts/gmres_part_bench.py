"""Cost of the row-partitioned GMRES iterable on ONE rank (SelfComm) next to the fused single-GPU iterable:
config 3 (advection_dominated N=50, restart 30) and a 128^3 Laplacian.  The partitioned handle finalises every
projection to a host scalar (that is where the ranks' sums meet), so MGS pays k host round trips per step; CGS /
DGKS pay one.   python scripts/gmres_part_bench.py"""
import json
import os
import sys
import time
from importlib import import_module

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
dist = import_module(pkg.__name__ + ".dist")
import torch  # noqa: E402


def run(name, n, colptr, rowval, nzval, b, restart, iters):
    S = sp.csc_matrix((nzval, rowval - 1, colptr - 1), shape=(n, n)).tocsr()
    offsets = np.array([0, n])
    ptr, idx, val = S.indptr.astype(np.int64), S.indices.astype(np.int64), S.data
    li, plan = dist.localize_block(ptr, idx, offsets, 0)
    dist.complete_plan(plan, offsets, [plan.ghost_gids])
    dA = pkg.HipCSR(n, n, colptr, rowval, nzval, index_base=1)
    for mname, M in (("mgs", pkg.ModifiedGramSchmidt()), ("cgs", pkg.ClassicalGramSchmidt()), ("dgks", pkg.DGKS())):
        out = {"case": name, "orth": mname, "restart": restart}
        for label in ("fused", "partitioned"):
            if label == "fused":
                it = pkg.gmres_iterable_(pkg.zerox(dA, pkg.HipVector.from_numpy(b)), dA, pkg.HipVector.from_numpy(b), restart=restart,
                                         reltol=0.0, maxiter=10 ** 9, initially_zero=True, orth_meth=M)
            else:
                it = dist.DistGMRESIterable(pkg, dist.SelfComm(), ptr, li, val, plan, b, restart=restart, reltol=0.0, maxiter=10 ** 9,
                                            orth_meth=M, n_global=n)
            i = 0
            for _ in range(restart):          # one warm cycle
                it.iterate(i); i += 1
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                it.iterate(i); i += 1
            torch.cuda.synchronize()
            out[label + "_us_per_iter"] = (time.perf_counter() - t0) / iters * 1e6
            out[label + "_residual"] = it.residual_current
        out["same_bits"] = out["fused_residual"] == out["partitioned_residual"]
        print(json.dumps(out))


n, colptr, rowval, nzval, b = pkg.fixtures.advection_dominated(50, 1000.0)
run("advdiff50", n, colptr, rowval, nzval, b, 30, 90)
n, colptr, rowval, nzval = pkg.fixtures.laplace_matrix(128, 3)
run("laplace128", n, colptr, rowval, nzval, pkg.fixtures.hashed_rhs(n), 30, 60)
