import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
A = orc.laplace(12, 3).astype(np.float32)
b = orc.hashed_rhs(A.n).astype(np.float32)
dA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval)
db = pkg.HipVector.from_numpy(b)
W, L = pkg.default_context().reduce_shape(np.float32)
for fused in (True, False):
    x, ch = pkg.cg(dA, db, log=True, fused=fused, maxiter=3)
    shape = (1, 1, W, L) if fused else (W, L, W, L)
    xo, ho = orc.cg(A, b, mode="tree", shape=shape, maxiter=3)
    print("fused", fused, ch["resnorm"], ho["resnorm"], ch["resnorm"] == ho["resnorm"])
it = pkg.cg_iterator_(pkg.zerox(dA, db), dA, db, initially_zero=True)
print("res0 gpu", repr(it.residual), "oracle", repr(float(orc.nrm2(b, "tree", W, L))), "tol", it.tol)
# one step by hand with L1 ops
u = db.copy(); c = dA @ u
print("dot(u,c)", repr(pkg.dot(u, c)), repr(orc.dot(b, orc.spmv(A, b), "tree", W, L)), repr(orc.dot(b, orc.spmv(A, b), "tree", 1, 1)))
