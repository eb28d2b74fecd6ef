"""The HBM-bound GMRES leg of bench.py on its own (for rocprofv3: scripts/prof_r06.sh gmres_large): gmres!(restart = 30) on the 256^3
Laplacian, fp64, plain CSR arrays, 60 inner iterations.  ORTH=mgs|cgs|both  N=<grid>  LAYOUT=csr|auto  REPS=<calls>."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
N = int(os.environ.get("N", 256))
n, colptr, rowval, nzval = pkg.fixtures.laplace_matrix(N, 3)
A = pkg.HipCSR(n, n, colptr, rowval, nzval, index_base=1)
del colptr, rowval, nzval
A.set_layout(os.environ.get("LAYOUT", "csr"))
b = pkg.HipVector.from_numpy(pkg.fixtures.hashed_rhs(n))
want = os.environ.get("ORTH", "both")
out = bench.gmres_hbm_bound(A, b, n, reps=int(os.environ.get("REPS", 2)), methods=("mgs", "cgs") if want == "both" else (want,))
print(json.dumps(out))
