"""BASELINE.json configs[4] stand-in: gmres!(restart=50), fp32, irregular CSR (n = 1e6, ~34 M nnz, rows from 6 to ~20 k entries).

Development / profiling tool (GPU box).  Two synthetic generators (fixtures.irregular_matrix) -- `random`: columns uniform
over the matrix (no locality: the worst case for the gather of x) and `banded`: the same rows with columns inside a band
(the locality an RCM-ordered finite-element matrix such as s3dkq4m2 has) -- plus every *.mtx under $MIK_MTX_DIR
(benchmark/matrixmarket.jl:5 reads such a file).  Prints row-length statistics, SpMV time / GB/s of the CSR algorithmic
bytes for both CSR kernels, and GMRES time per inner iteration.

    KINDS=random,banded GMRES=1 python scripts/config5_bench.py
"""
import glob
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g

pkg = g.load_package()
L = pkg.lib()
n_req = int(os.environ.get("N", 1_000_000))
kinds = [k for k in os.environ.get("KINDS", "random,banded").split(",") if k]
cases = []
for kind in kinds:
    t0 = time.time()
    if kind == "fe_shell":      # the shape of s3dkq4m2 (benchmark/matrixmarket.jl:5): 6 unknowns per node, 9-node neighbourhoods
        n, rowptr, colidx, val = pkg.fixtures.fe_matrix((123, 123), 6, np.float32)
    elif kind == "stencil27":   # 27-point variable-coefficient stencil on 128^3, fp64, lexicographic numbering: qualifies for the 8-bit column codes
        n, rowptr, colidx, val = pkg.fixtures.fe_matrix((128, 128, 128), 1, np.float64, renumber=False)
    elif kind == "box27":       # constant-coefficient 27-point box stencil on 128^3, fp64: the wide slice-constant layout
        n, rowptr, colidx, val = pkg.fixtures.box_stencil_matrix(128, 3, np.float64)
    elif kind == "fe_hex":      # 3 unknowns per node, 27-node neighbourhoods, larger than the Infinity Cache
        n, rowptr, colidx, val = pkg.fixtures.fe_matrix((64, 64, 64), 3, np.float32)
    else:
        # C5_CACHE=<dir>: keep the generated arrays on disk between the separate rocprofv3 passes of one profiling call (the hash-defined
        # generator takes ~25 s of numpy per matrix)
        cache = os.path.join(os.environ["C5_CACHE"], f"{kind}_{n_req}.npz") if os.environ.get("C5_CACHE") and os.environ.get("LONG", "1") == "1" and "BAND" not in os.environ else None
        if cache and os.path.exists(cache):
            z = np.load(cache)
            n, rowptr, colidx, val = int(z["n"]), z["rowptr"], z["colidx"], z["val"]
        else:
            n, rowptr, colidx, val = pkg.fixtures.irregular_matrix(n_req, np.float32, long_rows=os.environ.get("LONG", "1") == "1",
                                                                   bandwidth=0 if kind == "random" else int(os.environ.get("BAND", 2000)))
            if cache:
                os.makedirs(os.path.dirname(cache), exist_ok=True)
                np.savez(cache, n=n, rowptr=rowptr, colidx=colidx, val=val)
    cases.append((kind, n, rowptr, colidx, val, time.time() - t0))
for path in sorted(glob.glob(os.path.join(os.environ.get("MIK_MTX_DIR", "/nonexistent"), "*.mtx"))):
    t0 = time.time()
    n, colptr, rowval, nzval = pkg.fixtures.read_matrix_market(path)
    import scipy.sparse as sp
    S = sp.csc_matrix((nzval, rowval - 1, colptr - 1), shape=(n, n)).tocsr()
    S.sort_indices()
    cases.append((os.path.basename(path), n, S.indptr.astype(np.int64), S.indices.astype(np.int64), S.data.astype(np.float32), time.time() - t0))

for kind, n, rowptr, colidx, val, tgen in cases:
    lens = np.diff(rowptr)
    rb = np.add.reduceat(lens, np.arange(0, n, 256))
    print(f"== {kind}: n {n} nnz {val.size} generated/read in {tgen:.1f} s; row length min {lens.min()} median {int(np.median(lens))} "
          f"mean {lens.mean():.1f} max {lens.max()}; rows > 64: {(lens > 64).sum()}; nnz per 256-row block max/mean {rb.max() / rb.mean():.2f}")
    b = pkg.HipVector.from_numpy(pkg.fixtures.hashed_rhs(n, dtype=val.dtype))
    y = pkg.HipVector(n, val.dtype)
    extra = {int(a.split("=")[0]): int(a.split("=")[1]) for a in os.environ.get("MIK_KNOBS", "").split(",") if a}     # development knobs for A/B runs
    for knobs, name in ((extra, "default layout" + (f" + knobs {extra}" if extra else "")), ({0: 8, 1: 1}, "CSR only: k_spmv_rowblock (products in LDS, long rows merged)")):   # MIK_KNOB_LAYOUTS bit 8 = no jagged slices, MIK_KNOB_CSR_KERNEL = 1
        if knobs is not extra and os.environ.get("CSR", "1") != "1":
            continue
        for k, vv in knobs.items():
            L.mik_set_tuning(k, vv)
        t0 = time.time()
        A = pkg.HipCSR(n, n, rowptr, colidx, val, index_base=0, is_csc=False)
        tup = time.time() - t0
        A.time_spmv(b, y, reps=3)          # first launch of a kernel instantiation pays lazy code-object loading
        ms = A.time_spmv(b, y, reps=20)
        A.time_spmv(b, y, reps=3, fused_dot=True)
        msf = A.time_spmv(b, y, reps=20, fused_dot=True)
        ab = A.spmv_algorithmic_bytes()
        print(f"   {name}: layout {A.layout()} kernel {A.spmv_kernel()}  SpMV {ms * 1e3:7.1f} us = {ab / ms / 1e6:6.0f} GB/s of {ab / 1e6:.0f} MB algorithmic "
              f"({ab / ms / 1e6 / 8000:.3f} of 8 TB/s; stored {A.spmv_stored_bytes() / 1e6:.0f} MB); with dot(x, y) {msf * 1e3:7.1f} us; upload {tup:.1f} s")
        if knobs is extra and os.environ.get("GMRES", "1") == "1":
            for oname, M in (("mgs", pkg.ModifiedGramSchmidt()), ("cgs", pkg.ClassicalGramSchmidt())):
                pkg.gmres(A, b, restart=50, orth_meth=M, maxiter=60)
                pkg.default_context().synchronize()
                t0 = time.perf_counter()
                x, ch = pkg.gmres(A, b, restart=50, orth_meth=M, log=True, maxiter=2000)
                pkg.default_context().synchronize()
                dt = time.perf_counter() - t0
                print(f"   gmres fp32 restart=50 {oname}: iters {ch.iters} converged {ch.isconverged} {dt * 1e3:.1f} ms  "
                      f"{dt / max(ch.iters, 1) * 1e6:.1f} us/inner-iteration  final rel {ch['resnorm'][-1] / ch['resnorm'][0]:.2e}")
        del A
        for k in knobs:
            L.mik_set_tuning(k, 0)
