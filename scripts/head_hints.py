"""CG step rate at 256^3 for cache-hint variants of the fused head sweep (development knob 26) and the two-launch head (knob 25)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package()
import torch
L = pkg.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n, colptr, rowval, nzval = pkg.fixtures.laplace_matrix(N, 3)
A = pkg.HipCSR(n, n, colptr, rowval, nzval, index_base=1)
b = pkg.HipVector.from_numpy(pkg.fixtures.hashed_rhs(n))
for label, knobs in (("two launches", {25: 1}), ("head, default hints (8)", {}), ("head, u' nt (12)", {26: 12}), ("head, c temporal (24)", {26: 24}),
                     ("head, both (28)", {26: 28}), ("head, x cached, u' nt (4)", {26: 4}), ("head, x cached, c temporal (16)", {26: 16})):
    for k, v in knobs.items():
        L.mik_set_tuning(k, v)
    it = pkg.cg_iterator_(pkg.zerox(A, b), A, b, reltol=0.0, initially_zero=True, maxiter=10 ** 9)
    k = 0
    for _ in range(20):
        it.iterate(k); k += 1
    it.profile(2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300):
        it.iterate(k); k += 1
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    pk = it.profile_kernels()
    it.profile(0)
    print(f"{label:36s} {300 / dt:7.0f} it/s (with events)  " + "  ".join(f"{kk} {v[0] / max(v[1], 1) * 1e3:6.1f} us" for kk, v in pk.items()), flush=True)
    for k2 in knobs:
        L.mik_set_tuning(k2, 0)
    del it
