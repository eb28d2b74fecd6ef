"""IDR(s) per-step time (src/idrs.jl:164-272; one SpMV per step) on one MI355X, s = 8, fp64:
  * advection_dominated(N = 50) (n = 125,000: the size the reference authors benchmark their non-symmetric solvers on,
    benchmark/benchmark-linear-systems.jl:57-77) -- launch-bound: one C call per step (mik_idrs_step) against the statement-by-statement
    composition from the L1 entry points (same bits);
  * the 256^3 Laplacian (HBM-bound): bytes the launches of a cycle move / time / 8 TB/s, default layout and plain CSR arrays.
    python scripts/idrs_bench.py [--cycles 6] [--only small|large]"""
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
ap.add_argument("--cycles", type=int, default=6)
ap.add_argument("--only", default="")
args = ap.parse_args()
pkg = graft.load_package()
import torch  # noqa: E402

S = 8


def shadow(n, dtype=np.float64):
    P = pkg.HipMatrix(n, S, dtype)
    v = pkg.fixtures.hashed_rhs(n) + 0.5
    for j in range(S):
        P.col(j).copy_from_host(np.roll(v, 7919 * (j + 1)).astype(dtype))
    return P


def timed(A, b, P, fused, cycles):
    it = pkg.extras.idrs_iterable_(None, pkg.zerox(A, b), A, b, S, None, 0.0, 0.0, 10 ** 9, P=P, fused=fused)
    state = (1, 1)
    for _ in range(S + 1):
        _, state = it.iterate(state)
    gc.collect()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(cycles * (S + 1)):
        r, state = it.iterate(state)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (cycles * (S + 1))
    return dt, float(r).hex()


out = {}
if args.only in ("", "small"):
    n, colptr, rowval, nzval, b = pkg.fixtures.advection_dominated(50, 1000.0)
    A = pkg.HipCSR(n, n, colptr, rowval, nzval, index_base=1)
    db = pkg.HipVector.from_numpy(b)
    P = shadow(n)
    f, rf = timed(A, db, P, True, 4 * args.cycles)
    u, ru = timed(A, db, P, False, 4 * args.cycles)
    out["advection_dominated_50"] = {"n": n, "s": S, "us_per_step_one_call": f * 1e6, "us_per_step_statement_by_statement": u * 1e6,
                                     "same_bits": rf == ru, "speedup": u / f}
    del A, db, P
if args.only in ("", "large"):
    N = 256
    n, colptr, rowval, nzval = pkg.fixtures.laplace_matrix(N, 3)
    A = pkg.HipCSR(n, n, colptr, rowval, nzval, index_base=1)
    del colptr, rowval, nzval
    db = pkg.HipVector.from_numpy(pkg.fixtures.hashed_rhs(n))
    P = shadow(n)
    words = (sum(2 * (S - k) + 2 + ((2 + 7 * k - 1) if k else 0) + (S - k) + 1 + 6 for k in range(S)) + 7) / (S + 1)
    rec = {"n": n, "s": S, "vector_words_per_row_per_step": words}
    for layout in ("auto", "csr"):
        A.set_layout(layout)
        f, rf = timed(A, db, P, True, args.cycles)
        moved = A.spmv_stored_bytes() + words * 8 * n
        rec["default_layout" if layout == "auto" else "csr_arrays"] = {"spmv_kernel": A.spmv_kernel(), "us_per_step": f * 1e6, "bytes_moved_per_step": moved,
                                                                        "frac_of_8000": moved / f / 8e12, "last_residual": rf}
    rec["same_bits_across_layouts"] = rec["default_layout"]["last_residual"] == rec["csr_arrays"]["last_residual"]
    out["laplace_256"] = rec
print(json.dumps(out))
