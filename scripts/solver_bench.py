"""Per-iteration time of the widened solvers on the 256^3 Laplacian, fp64, one MI355X:
PCG with a Jacobi Pl, BiCGStab(2), MINRES, Chebyshev -- next to plain CG.  `gbs` = bytes of the reference's
own operation sequence (SpMV algorithmic bytes + its vector sweeps, unfused) / time; `frac_moved_of_8000` = the bytes the
device path actually streams per iteration (the operator layout's stored bytes per SpMV + the words of its fused sweeps) / time
/ 8 TB/s -- the figure to hold against the 0.78 copy ceiling.
    python scripts/solver_bench.py [--grid 256] [--iters 60]"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=256)
ap.add_argument("--iters", type=int, default=60)
ap.add_argument("--only", default="")
args = ap.parse_args()
pkg = graft.load_package()
import torch  # noqa: E402

for kv in os.environ.get("MIK_KNOBS", "").split(","):            # development knobs for A/B runs, e.g. MIK_KNOBS=8=2 (MIK_KNOB_SOLVER_FORM = 2: no SpMV epilogues)
    if kv:
        pkg.lib().mik_set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1]))

N = args.grid
n, colptr, rowval, nzval = pkg.fixtures.laplace_matrix(N, 3)
A = pkg.HipCSR(n, n, colptr, rowval, nzval, index_base=1)
del colptr, rowval, nzval
b = pkg.HipVector.from_numpy(pkg.fixtures.hashed_rhs(n))
spmv_b = A.spmv_algorithmic_bytes()
vec = 8 * n


def timed(name, it, start, words, mv_per_iter, iters=args.iters, warm=5, moved_words=None):
    if args.only and name not in args.only.split(","):
        return
    i = start
    for _ in range(warm):
        _, i = it.iterate(i)
    gc.collect()                      # device buffers of the previous solver are freed here, not inside the timed loop
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    per = []
    for _ in range(iters):
        t1 = time.perf_counter()
        _, i = it.iterate(i)
        per.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    if os.environ.get("SOLVER_BENCH_DEBUG"):
        print(name, "per-iteration us:", " ".join("%.0f" % (p * 1e6) for p in per), file=sys.stderr)
    bytes_ = mv_per_iter * spmv_b + words * vec
    out = {"solver": name, "grid": N, "us_per_iter": dt * 1e6, "iters_per_sec": 1 / dt, "spmv_per_iter": mv_per_iter,
           "vector_words_per_row_unfused": words, "gbs_of_reference_sequence": bytes_ / dt / 1e9}
    if moved_words is not None:
        moved = mv_per_iter * A.spmv_stored_bytes() + moved_words * vec
        out.update({"vector_words_per_row_moved": moved_words, "bytes_moved_per_iter": moved, "frac_moved_of_8000": moved / dt / 8e12})
    print(json.dumps(out))


x = pkg.zerox(A, b)
timed("cg", pkg.cg_iterator_(x, A, b, reltol=0.0, initially_zero=True, maxiter=10 ** 9), 0, 12, 1, moved_words=3 + 5)             # r -= a c (+ |r|^2): 3; x += a u, u = r + b u: 5
d = pkg.HipVector.from_numpy(np.full(n, 6.0))
x = pkg.zerox(A, b)
timed("pcg_jacobi", pkg.cg_iterator_(x, A, b, pkg.JacobiPrec(d), reltol=0.0, initially_zero=True, maxiter=10 ** 9), 0, 15, 1, moved_words=5 + 5)   # r -= a c, c = r ./ d (+ 2 sums): 5; x, u sweep: 5
x = pkg.zerox(A, b)
# BiCGStab(2): per outer iteration 4 SpMV; sweeps: 2 dots x2 (2 words each) ... counted from src/bicgstabl.jl:88-132
l = 2
words = sum(2 + 3 * (j + 1) + 2 + 3 * (j + 1) + 3 for j in range(l)) + (l + 1) * (l + 1) * 2 + (l + 2) + (l + 2) + (l + 2) + 1
bit = pkg.bicgstabl_iterator_(x, A, b, 2, reltol=0.0, max_mv_products=10 ** 9, initial_zero=True)
kept = "8=2" not in os.environ.get("MIK_KNOBS", "")   # rho of the first column: segment sums left by the MR sweep of the step before
ep = bit.dot_shape() == x.ctx.spmv_dot_shape()      # sigma and rho (from the second column on) leave the SpMV launches: r_shadow read once each, no sweep
timed("bicgstabl2", bit, 0, words, 2 * l, iters=max(args.iters // 3, 10),
      moved_words=sum((1 if ep and j else (0 if kept and j == 0 else 2)) + 3 * (j + 1) + (1 if ep else 2) + 3 * (j + 1) + 3 for j in range(l)) + (l + 1) + (3 * l + 4) + (1 if kept else 0))   # + one-pass Gram + one-sweep MR update (+ r_shadow for the next rho)
x = pkg.zerox(A, b)
timed("minres", pkg.minres_iterable_(x, A, b, reltol=0.0, initially_zero=True, maxiter=10 ** 9), 1, 27, 1, moved_words=1 + 3 + 8)             # Lanczos step + projection in the SpMV epilogue (round 4): + v_prev; orthogonalise + norm: 3; tail: 8
x = pkg.zerox(A, b)
timed("chebyshev", pkg.chebyshev_iterable_(x, A, b, 4.5e-4, 12.0, reltol=0.0, initially_zero=True, maxiter=10 ** 9), 0, 2 + 3 + 3 + 3 + 1, 1,
      moved_words=3 + 6)                                               # direction: 3; x, r update + norm: 6
