"""Development probe (GPU box): where does the time of the irregular SpMV (configs[4] stand-ins) go?
One matrix per kind, operators built with several long-row segment lengths, launch-time development knobs varied on each.

    KINDS=banded,random SHORT=0 SEGS=0,2048 KNOBSETS=,0=1,29=2 python scripts/banded_probe.py
(run under `rocprofv3 --kernel-trace --stats` for the per-kernel split: the instantiations carry the variant in their template arguments)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g

pkg = g.load_package()
L = pkg.lib()
knobsets = [dict((int(a.split("=")[0]), int(a.split("=")[1])) for a in kv.split("+") if a) for kv in os.environ.get("KNOBSETS", ",29=2,2=1,14=2,0=1").split(",")]
segs = [int(v) for v in os.environ.get("SEGS", "0").split(",")]
for kind in os.environ.get("KINDS", "banded,random").split(","):
    for long_rows in ((True, False) if os.environ.get("SHORT", "1") == "1" else (True,)):
        t0 = time.time()
        n, rowptr, colidx, val = pkg.fixtures.irregular_matrix(1_000_000, np.float32, long_rows=long_rows, bandwidth=0 if kind == "random" else 2000)
        print(f"== {kind} long_rows={long_rows}: nnz {val.size} generated in {time.time() - t0:.1f} s", flush=True)
        b = pkg.HipVector.from_numpy(pkg.fixtures.hashed_rhs(n, dtype=val.dtype))
        y = pkg.HipVector(n, val.dtype)
        for seg in segs:
            L.mik_set_tuning(15, seg)
            A = pkg.HipCSR(n, n, rowptr, colidx, val, index_base=0, is_csc=False)
            L.mik_set_tuning(15, 0)
            print(f"  -- long-row segment {seg or 'default'}", flush=True)
            ab = A.spmv_algorithmic_bytes()
            for knobs in knobsets:
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
