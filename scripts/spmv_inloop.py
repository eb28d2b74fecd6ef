"""A/B timing of SpMV kernel variants INSIDE the CG loop and back to back (development tool, GPU box).

    python scripts/spmv_inloop.py "8=1" "8=1,14=1" ""        # knob sets for mik_set_tuning, one operator upload each

Per variant: HIP-event time of the SpMV launch inside mik_cg_iterate (mik_cg_profile), back-to-back launch time
(mik_time_spmv), CG steps/s; rounds are interleaved in one process so that clock drift hits all variants alike."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g

pkg = g.load_package()
L = pkg.lib()
N = int(os.environ.get("N", 256))
STEPS = int(os.environ.get("STEPS", 60))
ROUNDS = int(os.environ.get("ROUNDS", 3))
specs = sys.argv[1:] or [""]
variants = [{int(k): int(v) for k, v in (kv.split("=") for kv in spec.split(",") if kv)} for spec in specs]


def set_knobs(v):
    for k in range(32):
        L.mik_set_tuning(k, v.get(k, 0))


b = pkg.HipVector.from_numpy(pkg.fixtures.hashed_rhs(N ** 3))
ops = []
for v in variants:
    set_knobs(v)
    n, colptr, rowval, nzval = pkg.fixtures.laplace_matrix(N, 3)
    ops.append(pkg.HipCSR(n, n, colptr, rowval, nzval, index_base=1))
    del colptr, rowval, nzval
res = {i: dict(inloop=[], b2b=[], step=[]) for i in range(len(variants))}
final = {}
for rnd in range(ROUNDS):
    for i, v in enumerate(variants):
        set_knobs(v)
        A = ops[i]
        it = pkg.cg_iterator_(pkg.zerox(A, b), A, b, reltol=0.0, initially_zero=True, maxiter=10 ** 9)
        k = 0
        for _ in range(5):
            it.iterate(k); k += 1
        it.profile(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(STEPS):
            it.iterate(k); k += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ms, launches = it.profile(0)
        res[i]["inloop"].append(ms / launches)
        res[i]["step"].append(dt / STEPS * 1e3)
        u = pkg.HipVector.wrap(it.u.ptr, A.n_rows, np.float64, pkg.default_context(), owner=it.u)
        y = pkg.HipVector(A.n_rows)
        A.time_spmv(u, y, reps=3, fused_dot=True)
        res[i]["b2b"].append(A.time_spmv(u, y, reps=20, fused_dot=True))
        final[i] = it.residual
        del it, y
alg = ops[0].spmv_algorithmic_bytes()
print(f"N={N} algorithmic CSR bytes {alg}  ({ROUNDS} rounds x {STEPS} steps)")
for i, spec in enumerate(specs):
    A = ops[i]
    set_knobs(variants[i])
    il, bb, st = (np.median(res[i][k]) for k in ("inloop", "b2b", "step"))
    sb = A.spmv_stored_bytes()
    print(f"{spec or '(default)':>16s} {A.layout():>36s}  in-loop {il*1e3:7.1f} us  b2b {bb*1e3:7.1f} us  step {st*1e3:7.1f} us"
          f"  alg {alg/il/1e6:6.0f} GB/s  moved {sb/il/1e6:6.0f} GB/s ({sb/il/1e6/8000:.3f} of 8 TB/s)  res {final[i]:.17g}")
