import sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
import __graft_entry__ as g
pkg = g.load_package()
ctx = pkg.default_context()
def run(A, b, knob, restart, inner):
    ctx.set_tuning(5, knob)
    it = pkg.gmres_iterable_(pkg.zerox(A, b), A, b, restart=restart, orth_meth=pkg.ModifiedGramSchmidt(), initially_zero=True, reltol=0.0, maxiter=inner)
    h = it.iterate_many(0, inner); x = it.x.to_numpy(); ctx.set_tuning(5, 0)
    return h, x
n, cp, rv, nz = pkg.fixtures.laplace_matrix(256, 3)
b64 = pkg.fixtures.hashed_rhs(n)
for dtype, machine in ((np.float64, 128 | (8 << 16)), (np.float32, 128 | (8 << 16)), (np.float32, 64 | (8 << 16)), (np.float32, 0)):
    ctx.set_tuning(12, machine)
    A = pkg.HipCSR(n, n, cp, rv, nz.astype(dtype), index_base=1)
    b = pkg.HipVector.from_numpy(b64.astype(dtype))
    h1, x1 = run(A, b, 0, 6, 9); h0, x0 = run(A, b, 6, 6, 9)
    ctx.set_tuning(12, 0)
    print(dtype.__name__, hex(machine), 'same', np.array_equal(h1, h0), np.array_equal(x1, x0), h1[:4], h0[:4], flush=True)
    del A, b
