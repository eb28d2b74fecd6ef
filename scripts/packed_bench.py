"""Plain CSR vs dictionary-coded operator on the 256^3 Laplacian: SpMV time and CG iterations/s (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
N = int(os.environ.get("N", 256))
n, colptr, rowval, nzval = pkg.fixtures.laplace_matrix(N, 3)
A = pkg.HipCSR(n, n, colptr, rowval, nzval)
x = pkg.HipVector.from_numpy(pkg.fixtures.hashed_rhs(n)); y = pkg.HipVector(n)
def cg_rate(A, K=300):
    it = pkg.cg_iterator_(pkg.zerox(A, x), A, x, initially_zero=True, maxiter=10 ** 9)
    it.iterate_many(0, 20); pkg.default_context().synchronize()
    t0 = time.perf_counter(); r = it.iterate_many(20, K); pkg.default_context().synchronize()
    return r.size / (time.perf_counter() - t0), r
A.time_spmv(x, y, reps=3, fused_dot=True); t_plain = A.time_spmv(x, y, reps=20, fused_dot=True); r_plain, h_plain = cg_rate(A)
t0 = time.time(); ok = A.pack(); print("pack:", ok, f"{time.time()-t0:.1f} s")
A.time_spmv(x, y, reps=3, fused_dot=True); t_pack = A.time_spmv(x, y, reps=20, fused_dot=True); r_pack, h_pack = cg_rate(A)
print(f"SpMV plain {t_plain*1e3:.1f} us  packed {t_pack*1e3:.1f} us   CG plain {r_plain:.0f} it/s  packed {r_pack:.0f} it/s  histories identical: {np.array_equal(h_plain, h_pack)}")
