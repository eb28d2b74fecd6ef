"""LSQR, LSMR and QMR (src/lsqr.jl, src/lsmr.jl, src/qmr.jl) per iteration on the 256^3 Laplacian, fp64, one MI355X: two SpMV per iteration (A and
adjoint(A) -- the second operator is the SAME CSC arrays uploaded as CSR, HipCSR.with_adjoint) plus their vector statements: LSQR / LSMR with the
fused sweeps (mik_xpby_nrm2, mik_lsqr_update / mik_lsmr_update; QMR: mik_axpy2_dot, mik_scal2, mik_qmr_update) and statement by statement (several host-visible
norms per iteration like the reference's loop).  `frac` = bytes the launches of an iteration move (both operators' stored bytes + the words per
row of its sweeps) / time / 8 TB/s.
    python scripts/adjoint_solver_bench.py [--grid 256] [--iters 30]"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=256)
ap.add_argument("--iters", type=int, default=30)
args = ap.parse_args()
pkg = graft.load_package()
import torch  # noqa: E402

N = args.grid
n, colptr, rowval, nzval = pkg.fixtures.laplace_matrix(N, 3)
A = pkg.extras.with_adjoint(n, n, colptr, rowval, nzval, index_base=1)
del colptr, rowval, nzval
b = pkg.HipVector.from_numpy(pkg.fixtures.hashed_rhs(n))
out = {"grid": N, "n": n, "operator_layout": A.layout(), "adjoint_layout": pkg.extras.adjoint(A).layout()}
spmv = A.spmv_stored_bytes() + pkg.extras.adjoint(A).spmv_stored_bytes()


def run(name, fn, words):
    fn(3)                                                                    # warm: allocations, first launches
    gc.collect()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h = fn(args.iters + 3)
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    t0 = time.perf_counter()
    fn(3)
    torch.cuda.synchronize()
    dt = (t_all - (time.perf_counter() - t0)) / args.iters                   # per iteration, the set-up (initial products, allocations) cancelled
    moved = spmv + words * 8 * n
    out[name] = {"us_per_iteration": dt * 1e6, "vector_words_per_row": words, "bytes_moved": moved, "frac_of_8000": moved / dt / 8e12,
                 "iterations": int(h.iters)}


# words per row of the sweeps of one iteration (reads + writes): fused / unfused as the reference writes them
run("lsqr", lambda k: pkg.extras.lsqr(A, b, maxiter=k, atol=0.0, btol=0.0, conlim=0.0, log=True)[1], 3 + 2 + 3 + 2 + 5)
run("lsqr_statement_by_statement", lambda k: pkg.extras.lsqr(A, b, maxiter=k, atol=0.0, btol=0.0, conlim=0.0, log=True, fused=False)[1], 3 + 1 + 2 + 3 + 1 + 2 + 3 + 3 + 2 + 2 + 1)
run("lsmr", lambda k: pkg.extras.lsmr(A, b, maxiter=k, atol=0.0, btol=0.0, conlim=0.0, log=True)[1], 3 + 2 + 3 + 2 + 7)
run("lsmr_statement_by_statement", lambda k: pkg.extras.lsmr(A, b, maxiter=k, atol=0.0, btol=0.0, conlim=0.0, log=True, fused=False)[1], 3 + 1 + 2 + 3 + 1 + 2 + 3 + 3 + 3 + 1)
run("qmr", lambda k: pkg.extras.qmr(A, b, maxiter=k, reltol=0.0, log=True)[1], 2 + 4 + 5 + 4 + 6)
run("qmr_statement_by_statement", lambda k: pkg.extras.qmr(A, b, maxiter=k, reltol=0.0, log=True, fused=False)[1], 2 + 3 + 3 + 2 + 3 + 3 + 2 + 2 + 2 + 2 + 3 + 3 + 2 + 3 + 2 + 2)
print(json.dumps(out))
