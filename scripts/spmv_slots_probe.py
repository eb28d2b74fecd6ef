"""How the layout-5 SpMV scales with the number of slots: the same 16.7 M rows as diagonal / 1-D 3-point / 2-D 5-point /
3-D 7-point operators, back-to-back launch time and the bytes each streams (GPU box)."""
import sys, os
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft

pkg = graft.load_package()
import torch
n = 256 ** 3


def lap(dims):
    I = [sp.identity(d, format="csr") for d in dims]
    T = [sp.diags([-np.ones(d - 1), 2.0 * np.ones(d), -np.ones(d - 1)], [-1, 0, 1], format="csr") for d in dims]
    out = None
    for k in range(len(dims)):
        term = None
        for j in range(len(dims)):
            m = T[j] if j == k else I[j]
            term = m if term is None else sp.kron(m, term, format="csr")      # first dim fastest
        out = term if out is None else out + term
    return out.tocsc()


cases = {"diagonal": sp.identity(n, format="csc") * 3.0, "1-D 3-point": lap([n]), "2-D 5-point": lap([4096, 4096]), "3-D 7-point": lap([256, 256, 256])}
for name, M in cases.items():
    M.sort_indices()
    A = pkg.HipCSR(n, n, M.indptr.astype(np.int64), M.indices.astype(np.int64), M.data.astype(np.float64), index_base=0)
    x = pkg.HipVector.from_numpy(np.random.default_rng(1).standard_normal(n))
    y = pkg.HipVector(n)
    for fd in (False, True):
        A.time_spmv(x, y, reps=3, fused_dot=fd)
        ms = A.time_spmv(x, y, reps=30, fused_dot=fd)
        sb = A.spmv_stored_bytes()
        print(f"{name:12s} {A.layout():40s} {A.spmv_kernel():14s} fused_dot={int(fd)}  {ms * 1e3:7.1f} us  {sb / 1e6:7.1f} MB  {sb / ms / 1e6:7.0f} GB/s", flush=True)
    del A, x, y
