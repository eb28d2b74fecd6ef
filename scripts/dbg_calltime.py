"""Wall time spent inside each libmik entry point during MINRES iterations at 256^3 (development aid)."""
import collections, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package()
import torch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n, colptr, rowval, nzval = pkg.fixtures.laplace_matrix(N, 3)
A = pkg.HipCSR(n, n, colptr, rowval, nzval, index_base=1)
b = pkg.HipVector.from_numpy(pkg.fixtures.hashed_rhs(n))
L = pkg.lib()
acc = collections.defaultdict(lambda: [0, 0.0])
class Wrap:
    def __init__(self, L): object.__setattr__(self, "L", L)
    def __getattr__(self, name):
        f = getattr(self.L, name)
        def g(*a):
            t = time.perf_counter(); r = f(*a); dt = time.perf_counter() - t
            acc[name][0] += 1; acc[name][1] += dt
            return r
        return g
api = sys.modules[pkg.__name__ + ".api"]
orig = api.lib
api.lib = lambda: Wrap(orig())
if len(sys.argv) > 2:
    xb = pkg.zerox(A, b)
    itb = pkg.bicgstabl_iterator_(xb, A, b, 2, reltol=0.0, max_mv_products=10 ** 9, initial_zero=True)
    j = 0
    for _ in range(int(sys.argv[2])): _, j = itb.iterate(j)
    if len(sys.argv) > 3: del itb, xb
x = pkg.zerox(A, b)
it = pkg.minres_iterable_(x, A, b, reltol=0.0, initially_zero=True, maxiter=10 ** 9)
i = 1
for _ in range(5): _, i = it.iterate(i)
acc.clear()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(40): _, i = it.iterate(i)
torch.cuda.synchronize(); tot = time.perf_counter() - t0
print("total per iter us", tot / 40 * 1e6)
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:24s} calls/iter {c/40:5.1f}  us/iter {t/40*1e6:9.1f}")
