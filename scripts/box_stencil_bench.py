"""cg! on a constant-coefficient (3^dims)-point box stencil (27-point in 3-D): the operator class of the wide slice-constant layout
(k_spmv_sdiaw).  Prints one JSON line: CG iterations/s, the SpMV's HIP-event time inside the loop, bytes moved and the fraction of
8 TB/s on them, the same loop on the plain CSR arrays, upload time.  Development / measurement tool (GPU box).

    python scripts/box_stencil_bench.py --grid 256 [--dims 3] [--steps 100]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=128)
ap.add_argument("--dims", type=int, default=3)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--no-csr", action="store_true")
ap.add_argument("--shape", default="", help="nodes per dimension, first fastest (e.g. 128,130,126) instead of --grid^--dims")
args = ap.parse_args()
pkg = graft.load_package()
import torch  # noqa: E402

for kv in os.environ.get("MIK_KNOBS", "").split(","):            # development knobs for A/B runs
    if kv:
        pkg.lib().mik_set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1]))

t0 = time.perf_counter()
if args.shape:
    shape = tuple(int(v) for v in args.shape.split(","))
    n, rowptr, colidx, val = pkg.fixtures.fe_matrix(shape, 1, np.float64, renumber=False)
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rowptr))
    val = np.ascontiguousarray(np.where(colidx == rows, float(3 ** len(shape) - 1), -1.0))
    del rows
    args.dims = len(shape)
    args.grid = "x".join(str(v) for v in shape)
else:
    n, rowptr, colidx, val = pkg.fixtures.box_stencil_matrix(args.grid, args.dims, np.float64)
tgen = time.perf_counter() - t0
t0 = time.perf_counter()
A = pkg.HipCSR(n, n, rowptr, colidx, val, index_base=0, is_csc=False)
tup = time.perf_counter() - t0
nnz = A.nnz
del rowptr, colidx, val
b = pkg.HipVector.from_numpy(pkg.fixtures.hashed_rhs(n))


def loop(Aop):
    it = pkg.cg_iterator_(pkg.zerox(Aop, b), Aop, b, reltol=0.0, initially_zero=True, maxiter=10 ** 9)
    k = 0
    for _ in range(10):
        it.iterate(k); k += 1
    it.profile(1)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        it.iterate(k); k += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t1) / args.steps
    ms, cnt = it.profile(0)
    return dt, ms / max(cnt, 1)


gridname = str(args.grid) if args.shape else "%s^%d" % (args.grid, args.dims)
dt, spmv_ms = loop(A)
sb, ab = A.spmv_stored_bytes(), A.spmv_algorithmic_bytes()
out = {"workload": f"cg! on the {3 ** args.dims}-point box stencil, {gridname}, fp64", "n": n, "nnz": nnz, "layout": A.layout(), "kernel": A.spmv_kernel(),
       "iters_per_sec": 1 / dt, "us_per_step": dt * 1e6, "spmv_in_loop_us": spmv_ms * 1e3, "bytes_moved_per_launch": sb,
       "frac_moved_of_8000": sb / (spmv_ms * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes_per_launch": ab,
       "generate_seconds": tgen, "upload_seconds": tup}
if not args.no_csr:
    A.set_layout("csr")
    dt2, ms2 = loop(A)
    out["csr_layout"] = {"kernel": A.spmv_kernel(), "iters_per_sec": 1 / dt2, "spmv_in_loop_us": ms2 * 1e3, "frac_of_8000": ab / (ms2 * 1e-3) / 1e9 / 8000.0}
    A.set_layout("auto")
print(json.dumps(out))
