"""Operator upload time (mik_csr_create) at 256^3: device-side pipeline vs the host path (MIK_KNOB_UPLOAD = 1).

    python scripts/upload_bench.py            # N=256 by default
"""
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
n, colptr, rowval, nzval = pkg.fixtures.laplace_matrix(N, 3)
x = pkg.HipVector.from_numpy(np.random.default_rng(0).standard_normal(n))
ref = None
# development knobs (include/mik_dev.h): 4 = MIK_KNOB_UPLOAD (1: host path), 0 = MIK_KNOB_LAYOUTS (1: CSR only, 2: no slice-constant values)
for label, knobs in (("device pipeline", {}), ("host path", {4: 1}), ("device pipeline, per-row values", {0: 2}), ("host path, per-row values", {4: 1, 0: 2}),
                     ("device pipeline, CSR only", {0: 1}), ("host path, CSR only", {4: 1, 0: 1})):
    for k, v in knobs.items():
        L.mik_set_tuning(k, v)
    ts = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        A = pkg.HipCSR(n, n, colptr, rowval, nzval, index_base=1)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        if rep < 2:
            del A
    y = pkg.mul_(pkg.HipVector(n), A, x).to_numpy()
    if ref is None:
        ref = y
    print(f"{label:36s} {min(ts):6.3f} s (best of 3; first {ts[0]:.3f})  layout {A.layout():40s} kernel {A.spmv_kernel():18s} same bits: {np.array_equal(y, ref)}")
    del A
    for k in knobs:
        L.mik_set_tuning(k, 0)
