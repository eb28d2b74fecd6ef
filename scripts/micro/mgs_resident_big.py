import sys, time, os
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
pkg = g.load_package()
import torch
ctx = pkg.default_context()
def run(A, b, knob, restart, inner):
    ctx.set_tuning(5, knob)
    it = pkg.gmres_iterable_(pkg.zerox(A, b), A, b, restart=restart, orth_meth=pkg.ModifiedGramSchmidt(), initially_zero=True, reltol=0.0, maxiter=inner)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h = it.iterate_many(0, inner)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ctx.set_tuning(5, 0)
    return h, dt
for N in (272, 288):
    n, cp, rv, nz = pkg.fixtures.laplace_matrix(N, 3)
    A = pkg.HipCSR(n, n, cp, rv, nz, index_base=1)
    del cp, rv, nz
    b = pkg.HipVector.from_numpy(pkg.fixtures.hashed_rhs(n))
    nseg = -(-n // 1024); S = -(-nseg // 256); rounds = -(-S // 2)
    for rep in range(2):
        h0, t0 = run(A, b, 6, 30, 60); h1, t1 = run(A, b, 0, 30, 60)
    print(N, n, 'segments per workgroup', S, 'resident share', min(1.0, 26 / rounds), 'same', np.array_equal(h0, h1), 'chain us/inner', t0 / 60 * 1e6, 'resident us/inner', t1 / 60 * 1e6, flush=True)
    del A, b
