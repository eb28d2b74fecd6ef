"""GMRES timing on BASELINE.json configs[2]: gmres!(restart=30) on advection_dominated(N=50, beta=1000), fp64.
Development tool (GPU box): prints wall time per inner iteration for MGS / CGS / DGKS and the oracle's CPU time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
N = int(os.environ.get("N", 50)); restart = int(os.environ.get("RESTART", 30))
n, colptr, rowval, nzval, b = pkg.fixtures.advection_dominated(N, 1000.0)
A = pkg.HipCSR(n, n, colptr, rowval, nzval)
db = pkg.HipVector.from_numpy(b)
for name, M in (("mgs", pkg.ModifiedGramSchmidt()), ("cgs", pkg.ClassicalGramSchmidt()), ("dgks", pkg.DGKS())):
    pkg.gmres(A, db, restart=restart, orth_meth=M, maxiter=40)       # warm-up
    pkg.default_context().synchronize()
    t0 = time.perf_counter()
    x, ch = pkg.gmres(A, db, restart=restart, orth_meth=M, log=True)
    pkg.default_context().synchronize()
    dt = time.perf_counter() - t0
    # the same solve with the inner loop inside the library (mik_gmres_iterate_many): what a compiled host pays
    for _rep in range(2):                                    # the first full-length call of a process is not representative
        it = pkg.gmres_iterable_(pkg.zerox(A, db), A, db, restart=restart, orth_meth=M, initially_zero=True)
        pkg.default_context().synchronize()
        t1 = time.perf_counter()
        hist = it.iterate_many(0, 4096)
        pkg.default_context().synchronize()
        dt2 = time.perf_counter() - t1
    same = np.array_equal(hist, ch["resnorm"])
    print(f"{name}: iters {ch.iters} mvps {ch.mvps} converged {ch.isconverged}  {dt*1e3:8.1f} ms total  {dt/ch.iters*1e6:8.1f} us/inner-iteration "
          f"(Python loop)  {dt2/max(hist.size,1)*1e6:8.1f} us/inner-iteration (loop inside libmik, same history: {same})  final {ch['resnorm'][-1]:.3e}")
if os.environ.get("CPU", "1") == "1":
    orc = g.load_oracle()
    Ao = orc.CSC(n, colptr, rowval, nzval, 1)
    t0 = time.perf_counter(); xo, ho = orc.gmres(Ao, b, restart=restart); dt = time.perf_counter() - t0
    print(f"oracle SEQ (1 thread): iters {ho['iters']}  {dt*1e3:8.1f} ms total  {dt/ho['iters']*1e6:8.1f} us/inner-iteration")
