"""GMRES(30) per inner iteration against the number of reduction segments (development aid): even ONE workgroup costs 26.5 us
(k + 1 dependent publish / poll round trips through memory at ~1.1 us each); 245 segments 42 us (~2.0 us per pass)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
for kv in os.environ.get("MIK_KNOBS", "").split(","):
    if kv:
        pkg.lib().mik_set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1]))
for N in [int(v) for v in os.environ.get("NS", "8,16,24,32,40,50").split(",")]:
    n, cp, rv, nz, b = pkg.fixtures.advection_dominated(N)
    A = pkg.HipCSR(n, n, cp, rv, nz, index_base=1)
    db = pkg.HipVector.from_numpy(b)
    it = pkg.gmres_iterable_(pkg.HipVector(n).fill_(0), A, db, restart=30, maxiter=10**6, reltol=0.0, initially_zero=True)
    it.iterate_many(0, 60)
    pkg.default_context().synchronize(); t0 = time.perf_counter()
    r = it.iterate_many(60, 300)
    pkg.default_context().synchronize(); dt = time.perf_counter() - t0
    print(f"N={N} n={n} segments={(n + 511) // 512}: {dt / max(r.size,1) * 1e6:.1f} us per inner iteration ({r.size} its)")
