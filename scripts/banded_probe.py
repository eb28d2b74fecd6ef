"""Development probe (GPU box): where does the time of the irregular SpMV (configs[4] stand-ins) go?
One matrix, the launch-time development knobs varied on it; optionally the same matrix without its long rows."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g

pkg = g.load_package()
L = pkg.lib()
for kind in os.environ.get("KINDS", "banded,random").split(","):
    for long_rows in ((True, False) if os.environ.get("SHORT", "1") == "1" else (True,)):
        t0 = time.time()
        n, rowptr, colidx, val = pkg.fixtures.irregular_matrix(1_000_000, np.float32, long_rows=long_rows, bandwidth=0 if kind == "random" else 2000)
        print(f"== {kind} long_rows={long_rows}: nnz {val.size} generated in {time.time() - t0:.1f} s", flush=True)
        b = pkg.HipVector.from_numpy(pkg.fixtures.hashed_rhs(n, dtype=val.dtype))
        y = pkg.HipVector(n, val.dtype)
        A = pkg.HipCSR(n, n, rowptr, colidx, val, index_base=0, is_csc=False)
        ab = A.spmv_algorithmic_bytes()
        for knobs in ({}, {29: 2}, {2: 1}, {14: 2}, {0: 1}):
            for k, v in knobs.items():
                L.mik_set_tuning(k, v)
            A.time_spmv(b, y, reps=3)
            ms = min(A.time_spmv(b, y, reps=20) for _ in range(3))
            A.time_spmv(b, y, reps=3, fused_dot=True)
            msf = min(A.time_spmv(b, y, reps=20, fused_dot=True) for _ in range(2))
            print(f"   knobs {knobs}: kernel {A.spmv_kernel()}  SpMV {ms * 1e3:7.1f} us  {ab / ms / 1e6 / 8000:.3f} of 8 TB/s; with dot {msf * 1e3:7.1f} us", flush=True)
            for k in knobs:
                L.mik_set_tuning(k, 0)
        del A
