"""Per-pass cost of the Gram-Schmidt chains at small n: time orthogonalize for several k (development aid)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
rng = np.random.default_rng(0)
V = pkg.HipMatrix.from_numpy(np.asfortranarray(np.linalg.qr(rng.standard_normal((n, 31)))[0]))
w0 = rng.standard_normal(n)
ctx = pkg.default_context()
for name, M in (("mgs", pkg.ModifiedGramSchmidt()), ("cgs", pkg.ClassicalGramSchmidt())):
    res = []
    for k in (0, 1, 5, 10, 20, 30):
        w = pkg.HipVector.from_numpy(w0)
        h = np.zeros(max(k, 1))
        for _ in range(20):
            pkg.orthogonalize_and_normalize_(V, k, w, h, M)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            pkg.orthogonalize_and_normalize_(V, k, w, h, M)
        ctx.synchronize()
        res.append((k, (time.perf_counter() - t0) / 200 * 1e6))
    print(name, " ".join(f"k={k}: {t:.1f}us" for k, t in res), " slope(10..30) = %.2f us/pass" % ((res[-1][1] - res[3][1]) / 20))
