"""GMRES(30) at a bandwidth-bound size: 256^3 Laplacian, fp64 (development tool, GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
N = int(os.environ.get("N", 256))
n, colptr, rowval, nzval = pkg.fixtures.laplace_matrix(N, 3)
A = pkg.HipCSR(n, n, colptr, rowval, nzval); del colptr, rowval, nzval
b = pkg.HipVector.from_numpy(pkg.fixtures.hashed_rhs(n))
for name, M in (("mgs", pkg.ModifiedGramSchmidt()), ("cgs", pkg.ClassicalGramSchmidt())):
    pkg.gmres(A, b, restart=30, orth_meth=M, maxiter=31)
    pkg.default_context().synchronize(); t0 = time.perf_counter()
    x, ch = pkg.gmres(A, b, restart=30, orth_meth=M, maxiter=90, log=True)
    pkg.default_context().synchronize(); dt = time.perf_counter() - t0
    # bytes: per inner step k: SpMV + (4k + 1) n s (MGS fused passes + dot0 2n + scal 2n ...) ~ see DESIGN
    ks = np.tile(np.arange(1, 31), 3)[:ch.iters]
    mgs_bytes = sum((2 + 4 * (k - 1) + 3 + 2) * n * 8 for k in ks) if name == "mgs" else sum(((k + 1) + (k + 2) + 1 + 2) * n * 8 for k in ks)
    total = mgs_bytes + ch.iters * A.spmv_algorithmic_bytes()
    print(f"{name}: {ch.iters} inner iterations in {dt*1e3:.1f} ms = {dt/ch.iters*1e3:.2f} ms/iteration; algorithmic {total/dt/1e9:.0f} GB/s")
