"""Modified Gram-Schmidt in the resident-w form (csrc/mik_mgs_res.h) against the multi-launch chain (MIK_KNOB_GS = 6): same residual history and
solution bit for bit, and microseconds per inner iteration, at five sizes (advection_dominated(N), fp64 / fp32) and on the 256^3 CSR operator.
    gpurun -- python scripts/micro/mgs_resident_check.py"""
import sys, time, json
import numpy as np
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import __graft_entry__ as g
pkg = g.load_package()
import torch
ctx = pkg.default_context()
out = {}
def run(A, b, knob, restart, inner, dtype):
    ctx.set_tuning(5, knob)
    it = pkg.gmres_iterable_(pkg.zerox(A, b), A, b, restart=restart, orth_meth=pkg.ModifiedGramSchmidt(), initially_zero=True, reltol=0.0, maxiter=inner)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h = it.iterate_many(0, inner)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    x = it.x.to_numpy()
    ctx.set_tuning(5, 0)
    return h, x, dt
for dtype in (np.float64, np.float32):
    for N in (140, 161, 200):
        n, cp, rv, nz, b = pkg.fixtures.advection_dominated(N, 300.0)
        A = pkg.HipCSR(n, n, cp, rv, nz.astype(dtype), index_base=1)
        db = pkg.HipVector.from_numpy(b.astype(dtype))
        h0, x0, t0 = run(A, db, 6, 7, 17, dtype)
        h1, x1, t1 = run(A, db, 0, 7, 17, dtype)
        h1, x1, t1 = run(A, db, 0, 7, 17, dtype)
        h0, x0, t0 = run(A, db, 6, 7, 17, dtype)
        print(dtype.__name__, N, n, 'same hist', np.array_equal(h0, h1), 'same x', np.array_equal(x0, x1), 'chain us', t0/17*1e6, 'resident us', t1/17*1e6, flush=True)
        del A, db
# 256^3
n, cp, rv, nz = pkg.fixtures.laplace_matrix(256, 3)
A = pkg.HipCSR(n, n, cp, rv, nz, index_base=1); A.set_layout("csr")
b = pkg.HipVector.from_numpy(pkg.fixtures.hashed_rhs(n))
for rep in range(2):
    h0, x0, t0 = run(A, b, 6, 30, 60, np.float64)
    h1, x1, t1 = run(A, b, 0, 30, 60, np.float64)
    print('256^3 same', np.array_equal(h0, h1), np.array_equal(x0, x1), 'chain us/inner', t0/60*1e6, 'resident us/inner', t1/60*1e6, flush=True)
