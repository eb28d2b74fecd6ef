"""CG step rate at 256^3 under single-bit flips of the vector kernels' cache-hint mask (development knob 7; default 121 with the
fused x update at the time; 248 since): which streams should bypass the caches now that the SpMV is k_spmv_sdiab2.
    HINT_BASE=248 python scripts/hint_sweep.py [extra masks...]      SWEEP_DIR=<knob 27> ONLY=1 ... <masks> for a fixed list"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package()
import torch
L = pkg.lib()
N = 256
n, colptr, rowval, nzval = pkg.fixtures.laplace_matrix(N, 3)
A = pkg.HipCSR(n, n, colptr, rowval, nzval, index_base=1)
b = pkg.HipVector.from_numpy(pkg.fixtures.hashed_rhs(n))
base = int(os.environ.get("HINT_BASE", "121"))
masks = ([base] + [base ^ (1 << i) for i in range(8)] if not os.environ.get("ONLY") else []) + [int(a) for a in sys.argv[1:]]


DIR = int(os.environ.get("SWEEP_DIR", "0"))       # development knob 27


def rate(mask, steps=400):
    L.mik_set_tuning(7, mask)
    L.mik_set_tuning(27, DIR)
    it = pkg.cg_iterator_(pkg.zerox(A, b), A, b, reltol=0.0, initially_zero=True, maxiter=10 ** 9)
    k = 0
    for _ in range(20):
        it.iterate(k); k += 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        it.iterate(k); k += 1
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    it.profile(2)
    for _ in range(60):
        it.iterate(k); k += 1
    torch.cuda.synchronize()
    pk = it.profile_kernels(); it.profile(0)
    L.mik_set_tuning(7, 0)
    L.mik_set_tuning(27, 0)
    return steps / dt, {kk: v[0] / max(v[1], 1) * 1e3 for kk, v in pk.items()}


for rnd in range(1 if os.environ.get("ONLY") else 2):
    for m in masks:
        r, pk = rate(m)
        print(f"round {rnd} mask {m:3d} = {m:08b}  {r:7.0f} it/s   " + "  ".join(f"{kk} {v:6.1f}" for kk, v in pk.items()), flush=True)
