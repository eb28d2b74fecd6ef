"""A/B timing of SpMV kernel variants on the 256^3 Laplacian (development tool, GPU box).
Variants are selected with mik_set_tuning; rounds are interleaved in one process."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
L = pkg.lib()
N = int(os.environ.get("N", 256))
n, colptr, rowval, nzval = pkg.fixtures.laplace_matrix(N, 3)
A = pkg.HipCSR(n, n, colptr, rowval, nzval)
x = pkg.HipVector.from_numpy(pkg.fixtures.hashed_rhs(n))
y = pkg.HipVector(n)
variants = []
for spec in sys.argv[1:]:
    variants.append({int(k): int(v) for k, v in (kv.split("=") for kv in spec.split(",") if kv)})
res = {i: [] for i in range(len(variants))}
for rnd in range(4):
    for i, v in enumerate(variants):
        for k in range(32):
            L.mik_set_tuning(k, v.get(k, 0))
        res[i].append(A.time_spmv(x, y, reps=20, fused_dot=True))
gb = A.spmv_algorithmic_bytes() / 1e9
for i, v in enumerate(variants):
    t = np.array(res[i][1:])
    print(f"{sys.argv[1+i]:>24s}  median {np.median(t)*1e3:7.1f} us  min {t.min()*1e3:7.1f} us  {gb/np.median(t)*1e3:7.0f} GB/s")
