"""CSR SpMV when x does not fit the 256 MB Infinity Cache: z-slabs of the 512 x 512 x NZ 7-point Laplacian, fp64, on the
plain CSR arrays (k_spmv_rowgather), generated on the device.  NZ=64 is one rank's slab of configs[3] (x = 134 MB, the north-star
size), NZ=256 has x = 537 MB and 5.9 GB of CSR arrays.  Back-to-back HIP-event time per launch, SURVEY.md 8d bytes, fraction of 8 TB/s;
plain launch and the launch with the CG step's fused dot.   NZS="64 256"  GRID=512  LAYOUT=csr|auto"""
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
d = importlib.import_module(pkg.__name__ + ".dist")
N = int(os.environ.get("GRID", 512))
out = {}
for NZ in [int(v) for v in os.environ.get("NZS", "64 256").split()]:
    n, ptr, idx, val = d._laplace_rows_torch(N, NZ, 0, N * N * NZ, np.float64, 0)
    torch.cuda.synchronize()
    ctx = pkg.default_context()
    A = pkg.HipCSR.from_device(n, n, int(val.numel()), ptr.data_ptr(), idx.data_ptr(), val.data_ptr(), np.float64, index_base=0, is_csc=False, ctx=ctx)
    del ptr, idx, val
    torch.cuda.empty_cache()
    A.set_layout(os.environ.get("LAYOUT", "csr"))
    x = pkg.HipVector.from_numpy(pkg.fixtures.hashed_rhs(n))
    y = pkg.HipVector(n)
    rec = {"n": int(n), "nnz": int(A.nnz), "x_bytes": int(n * 8), "operator_layout": A.layout(), "kernel": A.spmv_kernel(),
           "algorithmic_bytes_per_launch": A.spmv_algorithmic_bytes()}
    for fused in (False, True):
        A.time_spmv(x, y, reps=3, fused_dot=fused)
        ms = A.time_spmv(x, y, reps=int(os.environ.get("REPS", 20)), fused_dot=fused)
        key = "fused_dot" if fused else "plain"
        rec[key] = {"ms": ms, "achieved": rec["algorithmic_bytes_per_launch"] / (ms * 1e-3) / 1e9, "frac": rec["algorithmic_bytes_per_launch"] / (ms * 1e-3) / 1e9 / 8000.0}
    out[f"{N}x{N}x{NZ}"] = rec
    del A, x, y
    torch.cuda.empty_cache()
print(json.dumps(out))
