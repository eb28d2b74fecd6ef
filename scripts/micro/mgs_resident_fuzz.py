import sys, os, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
pkg = g.load_package()
import scipy.sparse as sp
ctx = pkg.default_context()
rng = np.random.default_rng(2026)
def run(A, b, knob, restart, inner):
    ctx.set_tuning(5, knob)
    it = pkg.gmres_iterable_(pkg.zerox(A, b), A, b, restart=restart, orth_meth=pkg.ModifiedGramSchmidt(), initially_zero=True, reltol=0.0, maxiter=inner)
    h = it.iterate_many(0, inner); x = it.x.to_numpy(); ctx.set_tuning(5, 0)
    return h, x
bad = 0
for case in range(28):
    dtype = np.float64 if case % 2 == 0 else np.float32
    seg = 1024 if dtype == np.float64 else 2048
    nseg = int(rng.integers(2049, 9000))
    n = nseg * seg - int(rng.integers(0, seg))          # random tail
    if case % 5 == 0: n |= 1                              # odd n
    # tridiagonal-ish nonsymmetric operator (cheap to build): CSC arrays directly
    d = np.full(n, 4.0); lo = np.full(n - 1, -1.0 - 0.3); up = np.full(n - 1, -1.0 + 0.3)
    M = sp.diags([lo, d, up], [-1, 0, 1], format="csc", dtype=dtype)
    A = pkg.HipCSR(n, n, M.indptr.astype(np.int64), M.indices.astype(np.int64), M.data, index_base=0)
    b = pkg.HipVector.from_numpy(rng.standard_normal(n).astype(dtype))
    restart = int(rng.integers(1, 12)); inner = restart + int(rng.integers(1, 6))
    h1, x1 = run(A, b, 0, restart, inner); h0, x0 = run(A, b, 6, restart, inner)
    ok = np.array_equal(h1, h0) and np.array_equal(x1, x0)
    bad += not ok
    print(case, dtype.__name__, 'n', n, 'nseg', -(-n // seg), 'S', -(-(-(-n // seg)) // 256), 'restart', restart, 'inner', inner, 'OK' if ok else 'MISMATCH', flush=True)
    del A, b, M
print('mismatches', bad)
