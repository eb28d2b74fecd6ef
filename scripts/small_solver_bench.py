"""BiCGStab(l) and MINRES at the sizes the reference authors benchmark them at.  BiCGStab(l) on their own benchmark of it -- advection_dominated(), n = 125,000, l = 2 and 4, max_mv_products
= 1000 (benchmark/benchmark-linear-systems.jl:68-77) -- per outer iteration: the whole-iteration C call with device-resident
scalars (mik_bicgstab_step; `fused`) against the same kernels driven statement by statement through the L1 entry points.
GPU box.
    python scripts/small_solver_bench.py [--N 50]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=50)
args = ap.parse_args()
pkg = graft.load_package()
import torch  # noqa: E402

n, colptr, rowval, nzval, b = pkg.fixtures.advection_dominated(args.N)
A = pkg.HipCSR(n, n, colptr, rowval, nzval, index_base=1)
db = pkg.HipVector.from_numpy(b)
sh = pkg.HipVector.from_numpy(pkg.fixtures.hashed_rhs(n) + 0.5)
for l in (2, 4):
    hist = {}
    for fused in (True, False):
        best = None
        for rep in range(3):
            x = pkg.zerox(A, db)
            it = pkg.bicgstabl_iterator_(x, A, db, l, max_mv_products=1000, initial_zero=True, r_shadow=sh, fused=fused)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = list(it)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        hist[fused] = np.array(res)
        print(json.dumps({"solver": f"bicgstabl(l={l})", "n": n, "path": "mik_bicgstab_step" if fused else "L1 entry points", "outer_iterations": len(res),
                          "mv_products": it.mv_products, "converged": bool(it.converged()), "seconds": best, "us_per_outer_iteration": best / max(len(res), 1) * 1e6,
                          "final_residual": float(res[-1]) if res else None}))
    # the whole-iteration call forms sigma and rho in the SpMV launches (one partial per 256-row block: mik_bicgstab_dot_shape), the L1 path as
    # vector dots: same statements, different summation trees -- each is bit-exact against the oracle with its own shape (tests/test_bicgstabl.py)
    m = min(len(hist[True]), len(hist[False]), 3)
    print(json.dumps({"solver": f"bicgstabl(l={l})", "first_residuals_max_rel_dev_between_the_paths": float(np.max(np.abs(hist[True][:m] - hist[False][:m]) / hist[False][:m]))}))

# MINRES as the reference benchmarks it (benchmark/benchmark-linear-systems.jl:80-86): SymTridiagonal(2.1, -1), n = 100,000, b = A * ones, maxiter = 100
n = 100_000
import scipy.sparse as sp  # noqa: E402
T3 = sp.diags([np.full(n - 1, -1.0), np.full(n, 2.1), np.full(n - 1, -1.0)], [-1, 0, 1], format="csc")
A3 = pkg.HipCSR(n, n, T3.indptr + 1, T3.indices + 1, T3.data, index_base=1)
b3 = pkg.HipVector.from_numpy(T3 @ np.ones(n))
hist = {}
for fused in (True, False):
    best = None
    for rep in range(3):
        x = pkg.zerox(A3, b3)
        it = pkg.minres_iterable_(x, A3, b3, initially_zero=True, maxiter=100, fused=fused)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = list(it)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    hist[fused] = np.array(res)
    print(json.dumps({"solver": "minres", "n": n, "path": "mik_minres_step" if fused else "L1 entry points", "iterations": len(res), "seconds": best,
                      "us_per_iteration": best / max(len(res), 1) * 1e6, "final_residual": float(res[-1])}))
m = min(len(hist[True]), len(hist[False]))
print(json.dumps({"solver": "minres", "history_max_rel_dev_between_the_paths": float(np.max(np.abs(hist[True][:m] - hist[False][:m]) / hist[False][:m]))}))
