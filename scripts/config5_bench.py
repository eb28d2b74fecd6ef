"""BASELINE.json configs[4] stand-in: gmres!(restart=50), fp32, synthetic irregular CSR (n = 1e6, 34 M nnz).
Development tool (GPU box): SpMV time / GB/s and GMRES time per inner iteration."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
n = int(os.environ.get("N", 1_000_000))
t0 = time.time(); n, rowptr, colidx, val = pkg.fixtures.irregular_matrix(n, np.float32, long_rows=os.environ.get("LONG", "1") == "1"); print(f"generated in {time.time()-t0:.1f} s: n {n} nnz {val.size} max row {np.diff(rowptr).max()}")
A = pkg.HipCSR(n, n, rowptr, colidx, val, index_base=0, is_csc=False)
b = pkg.HipVector.from_numpy(pkg.fixtures.hashed_rhs(n, dtype=np.float32))
y = pkg.HipVector(n, np.float32)
A.time_spmv(b, y, reps=3)          # first launch of a kernel instantiation pays lazy code-object loading
ms = A.time_spmv(b, y, reps=20)
msf = A.time_spmv(b, y, reps=3, fused_dot=True); msf = A.time_spmv(b, y, reps=20, fused_dot=True)
print(f"SpMV fused-dot variant (long rows in their own launch) {msf*1e3:.1f} us")
print(f"SpMV {ms*1e3:.1f} us  {A.spmv_algorithmic_bytes()/ms/1e6:.0f} GB/s algorithmic ({A.spmv_algorithmic_bytes()/1e6:.0f} MB)")
for name, M in ((("mgs", pkg.ModifiedGramSchmidt()), ("cgs", pkg.ClassicalGramSchmidt())) if os.environ.get("GMRES", "1") == "1" else ()):
    pkg.gmres(A, b, restart=50, orth_meth=M, maxiter=60)
    pkg.default_context().synchronize(); t0 = time.perf_counter()
    x, ch = pkg.gmres(A, b, restart=50, orth_meth=M, log=True, maxiter=2000)
    pkg.default_context().synchronize(); dt = time.perf_counter() - t0
    print(f"gmres fp32 restart=50 {name}: iters {ch.iters} converged {ch.isconverged} {dt*1e3:.1f} ms  {dt/max(ch.iters,1)*1e6:.1f} us/inner-iteration  final rel {ch['resnorm'][-1]/ch['resnorm'][0]:.2e}")
