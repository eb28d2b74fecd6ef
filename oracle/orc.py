"""ctypes binding of the CPU oracle (oracle/mik_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by ``__graft_entry__.smoke()`` and by the
``cpu_baseline`` leg of ``bench.py`` -- never by the product package.  Each wrapper names the
reference function it restates (paths relative to the IterativeSolvers.jl v0.9.4 checkout).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libmik_oracle.so")

SEQ, PAIR, TREE, BLAS = 0, 1, 2, 3
MODES = {"seq": SEQ, "pair": PAIR, "tree": TREE, "blas": BLAS}
MGS, CGS, DGKS = 0, 1, 2
METHODS = {"mgs": MGS, "cgs": CGS, "dgks": DGKS}

_lib = None


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or not os.path.exists(os.path.join(_HERE, "_build", "libmik_oracle_omp.so")) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("mik_oracle.c", "orc_impl.inc", "Makefile", "mik_oracle_omp.c")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


_blas = None


def blas_info():
    """The host OpenBLAS bound for mode="blas" (None until bind_blas ran)."""
    return _blas


def bind_blas(threads: int = 1):
    """Bind mode="blas" to the OpenBLAS that SciPy ships (LP64 CBLAS symbols ``scipy_cblas_*``): LinearAlgebra.dot /
    norm / mul! of the reference reach the same library family through libblastrampoline.  ``threads`` pins the
    BLAS thread count (1 = deterministic across machines with the same kernel; OpenBLAS splits ?dot / ?nrm2 across
    threads for long vectors, which changes the summation order).  Returns a description of what was bound."""
    global _blas
    import glob
    import scipy
    cands = sorted(glob.glob(os.path.join(os.path.dirname(scipy.__file__) + ".libs", "libscipy_openblas*.so")))
    if not cands:
        raise RuntimeError("no bundled OpenBLAS found next to SciPy")
    B = C.CDLL(cands[0])
    B.scipy_openblas_set_num_threads(int(threads))
    B.scipy_openblas_get_config.restype = C.c_char_p
    B.scipy_openblas_get_corename.restype = C.c_char_p
    L = lib()
    L.orc_set_blas.argtypes = [C.c_void_p] * 6
    L.orc_set_blas.restype = C.c_int
    fns = [C.cast(getattr(B, "scipy_cblas_" + nm), C.c_void_p) for nm in ("ddot", "dnrm2", "dgemv", "sdot", "snrm2", "sgemv")]
    assert L.orc_set_blas(*fns) == 0
    _blas = dict(library=os.path.basename(cands[0]), config=B.scipy_openblas_get_config().decode(),
                 core=B.scipy_openblas_get_corename().decode(), threads=int(threads), _handle=B)
    return {k: v for k, v in _blas.items() if not k.startswith("_")}


def _mode(mode):
    m = MODES[mode]
    if m == BLAS and _blas is None:
        bind_blas(1)
    return m


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct)) if a is not None else None


_i64p, _f64p, _f32p, _i32p = (C.POINTER(t) for t in (C.c_int64, C.c_double, C.c_float, C.c_int))


def _declare(L):
    for suf, fp, ft in (("f64", _f64p, C.c_double), ("f32", _f32p, C.c_float)):
        f = getattr(L, f"orc_csc_spmv_{suf}")
        f.argtypes = [C.c_int64, C.c_int64, _i64p, _i64p, fp, C.c_int, fp, fp]
        f.restype = None
        f = getattr(L, f"orc_dot_{suf}")
        f.argtypes = [fp, fp, C.c_int64, C.c_int, C.c_int, C.c_int]
        f.restype = ft
        f = getattr(L, f"orc_nrm2_{suf}")
        f.argtypes = [fp, C.c_int64, C.c_int, C.c_int, C.c_int]
        f.restype = ft
        f = getattr(L, f"orc_cg_{suf}")
        f.argtypes = [C.c_int64, _i64p, _i64p, fp, C.c_int, fp, fp, C.c_double, C.c_double,
                      C.c_int64, C.c_int, fp, C.c_int, _i32p, _f64p, _i64p, _i64p, _i32p, _f64p,
                      _f64p]
        f.restype = None
        f = getattr(L, f"orc_gmres_{suf}")
        f.argtypes = [C.c_int64, _i64p, _i64p, fp, C.c_int, fp, fp, C.c_double, C.c_double,
                      C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, _i32p, _f64p, _i64p, _i64p,
                      _i32p, _f64p, _f64p, fp, fp]
        f.restype = None
        f = getattr(L, f"orc_orthogonalize_{suf}")
        f.argtypes = [fp, C.c_int64, C.c_int64, C.c_int, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int]
        f.restype = ft
        f = getattr(L, f"orc_gemv_n_{suf}")
        f.argtypes = [fp, C.c_int64, C.c_int64, C.c_int, fp, ft, fp]
        f.restype = None
        f = getattr(L, f"orc_givens_{suf}")
        f.argtypes = [ft, ft, fp, fp, fp]
        f.restype = None
        f = getattr(L, f"orc_bicgstabl_{suf}")
        f.argtypes = [C.c_int64, _i64p, _i64p, fp, C.c_int, fp, fp, fp, fp, C.c_int, C.c_double, C.c_double, C.c_int64,
                      C.c_int, C.c_int, _i32p, _f64p, _i64p, _i64p, _i32p, _f64p, _f64p]
        f.restype = C.c_int
        f = getattr(L, f"orc_lu_solve_{suf}")
        f.argtypes = [fp, C.c_int64, C.c_int, fp]
        f.restype = C.c_int
        f = getattr(L, f"orc_chebyshev_{suf}")
        f.argtypes = [C.c_int64, _i64p, _i64p, fp, C.c_int, fp, fp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64,
                      C.c_int, fp, C.c_int, _i32p, _f64p, _i64p, _i64p, _i32p, _f64p, _f64p]
        f.restype = None
        f = getattr(L, f"orc_minres_{suf}")
        f.argtypes = [C.c_int64, _i64p, _i64p, fp, C.c_int, fp, fp, C.c_int, C.c_double, C.c_double, C.c_int64, C.c_int,
                      C.c_int, _i32p, _f64p, _i64p, _i64p, _i32p, _f64p, _f64p]
        f.restype = None
        f = getattr(L, f"orc_hessenberg_ldiv_{suf}")
        f.argtypes = [fp, C.c_int64, C.c_int, fp]
        f.restype = None
        f = getattr(L, f"orc_lsqr_{suf}")
        f.argtypes = [C.c_int64, C.c_int64, _i64p, _i64p, fp, _i64p, _i64p, fp, C.c_int, fp, fp, C.c_double, C.c_double, C.c_double, C.c_double,
                      C.c_int64, C.c_int, _i32p, _f64p, _f64p, _f64p, _f64p, _i64p, _i64p, _i64p, _i32p]
        f.restype = None
        f = getattr(L, f"orc_lsmr_{suf}")
        f.argtypes = [C.c_int64, C.c_int64, _i64p, _i64p, fp, _i64p, _i64p, fp, C.c_int, fp, fp, C.c_double, C.c_double, C.c_double, C.c_double,
                      C.c_int64, C.c_int, _i32p, _f64p, _f64p, _f64p, _i64p, _i64p, _i64p, _i32p]
        f.restype = None
        f = getattr(L, f"orc_qmr_{suf}")
        f.argtypes = [C.c_int64, _i64p, _i64p, fp, _i64p, _i64p, fp, C.c_int, fp, fp, C.c_double, C.c_double, C.c_int64, C.c_int, C.c_int, _i32p,
                      _f64p, _i64p, _i32p, _f64p, _f64p]
        f.restype = None
        f = getattr(L, f"orc_powm_{suf}")
        f.argtypes = [C.c_int64, _i64p, _i64p, fp, C.c_int, fp, C.c_double, C.c_int64, C.c_int, _i32p, _f64p, _i64p, _i32p, _f64p]
        f.restype = None
        f = getattr(L, f"orc_idrs_{suf}")
        f.argtypes = [C.c_int64, _i64p, _i64p, fp, C.c_int, fp, fp, fp, fp, C.c_int, C.c_double, C.c_double, C.c_int64, C.c_int,
                      C.c_int, _i32p, _f64p, _i64p, _i64p, _i32p, _f64p, _f64p]
        f.restype = None
    L.orc_laplace_nnz.argtypes = [C.c_int64, C.c_int]
    L.orc_laplace_nnz.restype = C.c_int64
    L.orc_laplace_csc.argtypes = [C.c_int64, C.c_int, C.c_int, _i64p, _i64p, _f64p]
    L.orc_laplace_csc.restype = None
    L.orc_advdiff_csc.argtypes = [C.c_int64, C.c_double, C.c_int, _i64p, _i64p, _f64p, _f64p]
    L.orc_advdiff_csc.restype = None
    L.orc_set_long_row.argtypes = [C.c_int64]
    L.orc_set_long_row.restype = None
    L.orc_set_long_segment.argtypes = [C.c_int64]
    L.orc_set_long_segment.restype = None
    L.orc_set_long_group.argtypes = [C.c_int64]
    L.orc_set_long_group.restype = None
    L.orc_set_partition.argtypes = [C.c_int, _i64p]
    L.orc_set_partition.restype = C.c_int
    L.orc_hashed_rhs.argtypes = [C.c_int64, _f64p]
    L.orc_hashed_rhs.restype = None


def set_partition(offsets=None):
    """TREE-mode reductions over full-length vectors follow a row partition (rank order); None resets."""
    if offsets is None:
        lib().orc_set_partition(0, None)
        return
    off = np.ascontiguousarray(offsets, np.int64)
    assert lib().orc_set_partition(off.size - 1, _p(off, C.c_int64)) == 0


def set_long_row(threshold=0, segment=0, group=4):
    """Rows with more than `threshold` entries use the device's wave-shaped row sum in spmv (0 = off): lane l of 64 sums the
    groups l, l + 64, ... of `group` consecutive entries; rows with more than `segment` entries are summed segment by segment
    (0 = never cut) -- mik_spmv_long_row() / mik_spmv_long_segment() / mik_spmv_long_group()."""
    lib().orc_set_long_row(int(threshold or 0))
    lib().orc_set_long_segment(int(segment or 0))
    lib().orc_set_long_group(int(group or 1))


def _suf(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "f64", C.c_double
    if dtype == np.float32:
        return "f32", C.c_float
    raise TypeError(f"oracle supports float64/float32, got {dtype}")


@dataclass
class CSC:
    """A Julia ``SparseMatrixCSC{T,Int64}``: colptr / rowval (``index_base``-based) / nzval."""
    n: int
    colptr: np.ndarray
    rowval: np.ndarray
    nzval: np.ndarray
    index_base: int = 1

    @property
    def nnz(self) -> int:
        return int(self.nzval.shape[0])

    def astype(self, dtype) -> "CSC":
        return CSC(self.n, self.colptr, self.rowval, self.nzval.astype(dtype), self.index_base)

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csc_matrix((self.nzval, self.rowval - self.index_base,
                              self.colptr - self.index_base), shape=(self.n, self.n))

    @staticmethod
    def from_dense(a, index_base: int = 1) -> "CSC":
        """All entries stored (a dense Matrix pushed through the sparse interface)."""
        a = np.asarray(a)
        n = a.shape[0]
        colptr = np.arange(0, n * n + 1, n, dtype=np.int64) + index_base
        rowval = np.tile(np.arange(n, dtype=np.int64), n) + index_base
        return CSC(n, colptr, rowval, np.ascontiguousarray(a.T).reshape(-1).copy(), index_base)

    @staticmethod
    def from_scipy(m, index_base: int = 1) -> "CSC":
        m = m.tocsc()
        m.sort_indices()
        return CSC(m.shape[0], m.indptr.astype(np.int64) + index_base,
                   m.indices.astype(np.int64) + index_base, m.data.copy(), index_base)


def laplace(N: int, dims: int = 3, index_base: int = 1) -> CSC:
    """``laplace_matrix(Float64, N, dims)`` -- test/laplace_matrix.jl:1-12."""
    L = lib()
    n = N ** dims
    nnz = L.orc_laplace_nnz(N, dims)
    colptr = np.empty(n + 1, np.int64)
    rowval = np.empty(nnz, np.int64)
    nzval = np.empty(nnz, np.float64)
    L.orc_laplace_csc(N, dims, index_base, _p(colptr, C.c_int64), _p(rowval, C.c_int64),
                      _p(nzval, C.c_double))
    return CSC(n, colptr, rowval, nzval, index_base)


def advdiff(N: int = 50, beta: float = 1000.0, index_base: int = 1):
    """``advection_dominated(;N, β)`` -- benchmark/advection_diffusion.jl:3-30 -> (A, b)."""
    L = lib()
    n = N ** 3
    nnz = L.orc_laplace_nnz(N, 3)
    colptr = np.empty(n + 1, np.int64)
    rowval = np.empty(nnz, np.int64)
    nzval = np.empty(nnz, np.float64)
    b = np.empty(n, np.float64)
    L.orc_advdiff_csc(N, beta, index_base, _p(colptr, C.c_int64), _p(rowval, C.c_int64),
                      _p(nzval, C.c_double), _p(b, C.c_double))
    return CSC(n, colptr, rowval, nzval, index_base), b


def hashed_rhs(n: int) -> np.ndarray:
    """b[i] = ((i*2654435761) mod 2^32)/2^32 - 0.5."""
    b = np.empty(n, np.float64)
    lib().orc_hashed_rhs(n, _p(b, C.c_double))
    return b


def spmv(A: CSC, x: np.ndarray) -> np.ndarray:
    """``mul!(y, A::SparseMatrixCSC, x)`` (column scatter) -- call sites src/cg.jl:54."""
    suf, ct = _suf(A.nzval.dtype)
    x = np.ascontiguousarray(x, A.nzval.dtype)
    y = np.empty(A.n, A.nzval.dtype)
    getattr(lib(), f"orc_csc_spmv_{suf}")(A.n, A.n, _p(A.colptr, C.c_int64),
                                          _p(A.rowval, C.c_int64), _p(A.nzval, ct),
                                          A.index_base, _p(x, ct), _p(y, ct))
    return y


def dot(x, y, mode="seq", W=1, L=1):
    suf, ct = _suf(x.dtype)
    x = np.ascontiguousarray(x)
    y = np.ascontiguousarray(y, x.dtype)
    return getattr(lib(), f"orc_dot_{suf}")(_p(x, ct), _p(y, ct), x.size, _mode(mode), W, L)


def nrm2(x, mode="seq", W=1, L=1):
    suf, ct = _suf(x.dtype)
    x = np.ascontiguousarray(x)
    return getattr(lib(), f"orc_nrm2_{suf}")(_p(x, ct), x.size, _mode(mode), W, L)


def _eps_sqrt(dtype):
    return float(np.sqrt(np.finfo(dtype).eps))


def cg(A: CSC, b, x0=None, *, abstol=0.0, reltol=None, maxiter=None, jacobi_diag=None,
       mode="seq", shape=(1, 1, 1, 1)):
    """``cg!(x, A, b; log=true)`` / ``cg(A, b)`` when ``x0 is None`` -- src/cg.jl:209-242,162.

    Returns ``(x, history)`` with history keys iters, mvps, isconverged, resnorm, res0, tol.
    """
    dtype = A.nzval.dtype
    suf, ct = _suf(dtype)
    b = np.ascontiguousarray(b, dtype)
    n = A.n
    initially_zero = x0 is None
    x = np.zeros(n, dtype) if x0 is None else np.array(x0, dtype, copy=True)
    reltol = _eps_sqrt(dtype) if reltol is None else reltol
    maxiter = n if maxiter is None else int(maxiter)
    res = np.zeros(max(maxiter, 1), np.float64)
    iters, mvps = C.c_int64(0), C.c_int64(0)
    conv = C.c_int(0)
    res0, tol = C.c_double(0), C.c_double(0)
    shp = np.asarray(shape, np.int32)
    jd = None if jacobi_diag is None else np.ascontiguousarray(jacobi_diag, dtype)
    getattr(lib(), f"orc_cg_{suf}")(n, _p(A.colptr, C.c_int64), _p(A.rowval, C.c_int64),
                                    _p(A.nzval, ct), A.index_base, _p(b, ct), _p(x, ct),
                                    float(abstol), float(reltol), maxiter, int(initially_zero),
                                    _p(jd, ct), _mode(mode), _p(shp, C.c_int),
                                    _p(res, C.c_double), C.byref(iters), C.byref(mvps),
                                    C.byref(conv), C.byref(res0), C.byref(tol))
    hist = dict(iters=iters.value, mvps=mvps.value, isconverged=bool(conv.value),
                resnorm=res[:iters.value].copy(), res0=res0.value, tol=tol.value,
                abstol=abstol, reltol=reltol)
    return x, hist


def gmres(A: CSC, b, x0=None, *, abstol=0.0, reltol=None, restart=None, maxiter=None,
          orth_meth="mgs", mode="seq", shape=(1, 1), pl_diag=None, pr_diag=None):
    """``gmres!(x, A, b; log=true)`` / ``gmres(A, b)`` when ``x0 is None`` -- src/gmres.jl:184-222,143."""
    dtype = A.nzval.dtype
    suf, ct = _suf(dtype)
    b = np.ascontiguousarray(b, dtype)
    n = A.n
    initially_zero = x0 is None
    x = np.zeros(n, dtype) if x0 is None else np.array(x0, dtype, copy=True)
    reltol = _eps_sqrt(dtype) if reltol is None else reltol
    restart = min(20, n) if restart is None else int(restart)
    maxiter = n if maxiter is None else int(maxiter)
    res = np.zeros(max(maxiter, 1), np.float64)
    iters, mvps = C.c_int64(0), C.c_int64(0)
    conv = C.c_int(0)
    beta0, tol = C.c_double(0), C.c_double(0)
    shp = np.asarray(shape, np.int32)
    getattr(lib(), f"orc_gmres_{suf}")(n, _p(A.colptr, C.c_int64), _p(A.rowval, C.c_int64),
                                       _p(A.nzval, ct), A.index_base, _p(b, ct), _p(x, ct),
                                       float(abstol), float(reltol), restart, maxiter,
                                       int(initially_zero), METHODS[orth_meth], _mode(mode),
                                       _p(shp, C.c_int), _p(res, C.c_double), C.byref(iters),
                                       C.byref(mvps), C.byref(conv), C.byref(beta0), C.byref(tol),
                                       _p(None if pl_diag is None else np.ascontiguousarray(pl_diag, dtype), ct),
                                       _p(None if pr_diag is None else np.ascontiguousarray(pr_diag, dtype), ct))
    hist = dict(iters=iters.value, mvps=mvps.value, isconverged=bool(conv.value),
                resnorm=res[:iters.value].copy(), res0=beta0.value, tol=tol.value,
                abstol=abstol, reltol=reltol, restart=restart)
    return x, hist


def orthogonalize(V, w, method="mgs", mode="seq", W=1, L=1):
    """``orthogonalize_and_normalize!(V, w, h, method)`` -- src/orthogonalize.jl:13-79.

    ``V`` is (n, k) Fortran-ordered; returns (w_new, h, nrm)."""
    V = np.asfortranarray(V)
    suf, ct = _suf(V.dtype)
    n, k = V.shape
    w = np.array(w, V.dtype, copy=True)
    h = np.zeros(k, V.dtype)
    nrm = getattr(lib(), f"orc_orthogonalize_{suf}")(_p(V, ct), n, n, k, _p(w, ct), _p(h, ct),
                                                     METHODS[method], _mode(mode), W, L)
    return w, h, nrm


def gemv_n(V, c, y, alpha=1.0):
    """``mul!(y, V, c, alpha, 1)`` -- src/gmres.jl:275, src/orthogonalize.jl:16."""
    V = np.asfortranarray(V)
    suf, ct = _suf(V.dtype)
    n, k = V.shape
    c = np.ascontiguousarray(c, V.dtype)
    y = np.array(y, V.dtype, copy=True)
    getattr(lib(), f"orc_gemv_n_{suf}")(_p(V, ct), n, n, k, _p(c, ct), alpha, _p(y, ct))
    return y


def givens(f, g, dtype=np.float64):
    """``LinearAlgebra.givensAlgorithm(f, g)`` -> (c, s, r) -- used at src/hessenberg.jl:24."""
    suf, ct = _suf(dtype)
    out = np.zeros(3, dtype)
    getattr(lib(), f"orc_givens_{suf}")(f, g, _p(out[0:1], ct), _p(out[1:2], ct), _p(out[2:3], ct))
    return tuple(out.tolist())


def hessenberg_ldiv(H, rhs):
    """``ldiv!(FastHessenberg(H), rhs)`` -- src/hessenberg.jl:15-46.  Returns (R, [y; residual])."""
    H = np.array(H, order="F", copy=True)
    suf, ct = _suf(H.dtype)
    rhs = np.array(rhs, H.dtype, copy=True)
    getattr(lib(), f"orc_hessenberg_ldiv_{suf}")(_p(H, ct), H.shape[0], H.shape[1], _p(rhs, ct))
    return H, rhs


def lu_solve(A, b):
    """``ldiv!(x, lu!(A), b)`` on a small dense matrix -- src/bicgstabl.jl:124-125.  Returns x."""
    A = np.array(A, order="F", copy=True)
    suf, ct = _suf(A.dtype)
    b = np.array(b, A.dtype, copy=True)
    rc = getattr(lib(), f"orc_lu_solve_{suf}")(_p(A, ct), A.shape[0], A.shape[0], _p(b, ct))
    if rc:
        raise np.linalg.LinAlgError("singular matrix")
    return b


def bicgstabl(A: CSC, b, l=2, x0=None, *, r_shadow, abstol=0.0, reltol=None, max_mv_products=None, mode="seq",
              shape=(1, 1), pl_diag=None, dot_shape=None):
    """``bicgstabl!(x, A, b, l; Pl, log=true)`` / ``bicgstabl(A, b, l)`` when ``x0 is None`` -- src/bicgstabl.jl:181-219,142.
    ``r_shadow`` replaces the reference's ``rand(T, n)`` (src/bicgstabl.jl:38); ``pl_diag`` = the diagonal of a Jacobi ``Pl``
    (``ldiv!`` at src/bicgstabl.jl:55,98,108), None = ``Identity()``.  ``dot_shape``: (W, L) of ``sigma`` (:100) and of ``rho`` from
    the second column on (:89) (mik_bicgstab_dot_shape), default = ``shape``."""
    dtype = A.nzval.dtype
    suf, ct = _suf(dtype)
    b = np.ascontiguousarray(b, dtype)
    rsh = np.ascontiguousarray(r_shadow, dtype)
    n = A.n
    initial_zero = x0 is None
    x = np.zeros(n, dtype) if x0 is None else np.array(x0, dtype, copy=True)
    reltol = _eps_sqrt(dtype) if reltol is None else reltol
    max_mv = n if max_mv_products is None else int(max_mv_products)
    res = np.zeros(max(max_mv, 1), np.float64)
    iters, mvps = C.c_int64(0), C.c_int64(0)
    conv = C.c_int(0)
    res0, tol = C.c_double(0), C.c_double(0)
    shp = np.asarray(tuple(shape) + tuple(dot_shape if dot_shape is not None else shape), np.int32)
    rc = getattr(lib(), f"orc_bicgstabl_{suf}")(n, _p(A.colptr, C.c_int64), _p(A.rowval, C.c_int64), _p(A.nzval, ct),
                                                A.index_base, _p(b, ct), _p(x, ct), _p(rsh, ct),
                                                _p(None if pl_diag is None else np.ascontiguousarray(pl_diag, dtype), ct), int(l), float(abstol),
                                                float(reltol), max_mv, int(initial_zero), _mode(mode), _p(shp, C.c_int),
                                                _p(res, C.c_double), C.byref(iters), C.byref(mvps), C.byref(conv),
                                                C.byref(res0), C.byref(tol))
    if rc:
        raise np.linalg.LinAlgError("singular matrix in the MR part")
    hist = dict(iters=iters.value, mvps=mvps.value, isconverged=bool(conv.value), resnorm=res[:iters.value].copy(),
                res0=res0.value, tol=tol.value, abstol=abstol, reltol=reltol)
    return x, hist


def _run_simple(fn_name, A, b, x0, maxiter, call):
    dtype = A.nzval.dtype
    suf, ct = _suf(dtype)
    b = np.ascontiguousarray(b, dtype)
    n = A.n
    x = np.zeros(n, dtype) if x0 is None else np.array(x0, dtype, copy=True)
    maxiter = n if maxiter is None else int(maxiter)
    res = np.zeros(max(maxiter, 1), np.float64)
    iters, mvps = C.c_int64(0), C.c_int64(0)
    conv = C.c_int(0)
    res0, tol = C.c_double(0), C.c_double(0)
    call(getattr(lib(), f"{fn_name}_{suf}"), ct, b, x, maxiter, res, iters, mvps, conv, res0, tol)
    return x, dict(iters=iters.value, mvps=mvps.value, isconverged=bool(conv.value), resnorm=res[:iters.value].copy(),
                   res0=res0.value, tol=tol.value)


def chebyshev(A: CSC, b, lmin, lmax, x0=None, *, abstol=0.0, reltol=None, maxiter=None, pl_diag=None, mode="seq", shape=(1, 1)):
    """``chebyshev!(x, A, b, lmin, lmax; log=true)`` / ``chebyshev(A, b, ...)`` when ``x0 is None`` -- src/chebyshev.jl:142-169,100."""
    dtype = A.nzval.dtype
    reltol = _eps_sqrt(dtype) if reltol is None else reltol
    shp = np.asarray(shape, np.int32)
    pd = None if pl_diag is None else np.ascontiguousarray(pl_diag, dtype)

    def call(f, ct, b, x, maxiter, res, iters, mvps, conv, res0, tol):
        f(A.n, _p(A.colptr, C.c_int64), _p(A.rowval, C.c_int64), _p(A.nzval, ct), A.index_base, _p(b, ct), _p(x, ct), float(lmin),
          float(lmax), float(abstol), float(reltol), maxiter, int(x0 is None), _p(pd, ct), _mode(mode), _p(shp, C.c_int),
          _p(res, C.c_double), C.byref(iters), C.byref(mvps), C.byref(conv), C.byref(res0), C.byref(tol))
    return _run_simple("orc_chebyshev", A, b, x0, maxiter, call)


def minres(A: CSC, b, x0=None, *, skew_hermitian=False, abstol=0.0, reltol=None, maxiter=None, mode="seq", shape=(1, 1), proj_shape=None):
    """``minres!(x, A, b; log=true)`` / ``minres(A, b)`` when ``x0 is None`` -- src/minres.jl:197-230,236.  ``proj_shape``: (W, L) of
    ``proj = dot(v_curr, v_next)`` (src/minres.jl:107; mik_minres_proj_shape), default = ``shape``."""
    dtype = A.nzval.dtype
    reltol = _eps_sqrt(dtype) if reltol is None else reltol
    shp = np.asarray(tuple(shape) + tuple(proj_shape if proj_shape is not None else shape), np.int32)

    def call(f, ct, b, x, maxiter, res, iters, mvps, conv, res0, tol):
        f(A.n, _p(A.colptr, C.c_int64), _p(A.rowval, C.c_int64), _p(A.nzval, ct), A.index_base, _p(b, ct), _p(x, ct),
          int(skew_hermitian), float(abstol), float(reltol), maxiter, int(x0 is None), _mode(mode), _p(shp, C.c_int),
          _p(res, C.c_double), C.byref(iters), C.byref(mvps), C.byref(conv), C.byref(res0), C.byref(tol))
    return _run_simple("orc_minres", A, b, x0, maxiter, call)


def idrs(A: CSC, b, x0=None, *, P, s=8, pl_diag=None, abstol=0.0, reltol=None, maxiter=None, smoothing=False, mode="seq", shape=(1, 1)):
    """``idrs!(x, A, b; s, Pl, smoothing, log=true)`` / ``idrs(A, b)`` when ``x0 is None`` -- src/idrs.jl:49-64,10.  ``P`` (n x s, any layout)
    replaces the reference's ``rand!`` shadow vectors (src/idrs.jl:136); ``pl_diag`` = the diagonal of a Jacobi ``Pl``."""
    dtype = A.nzval.dtype
    reltol = _eps_sqrt(dtype) if reltol is None else reltol
    shp = np.asarray(shape, np.int32)
    Pm = np.asfortranarray(np.asarray(P, dtype).reshape(A.n, int(s)))
    pd = None if pl_diag is None else np.ascontiguousarray(pl_diag, dtype)

    def call(f, ct, b, x, maxiter, res, iters, mvps, conv, res0, tol):
        f(A.n, _p(A.colptr, C.c_int64), _p(A.rowval, C.c_int64), _p(A.nzval, ct), A.index_base, _p(b, ct), _p(x, ct), _p(Pm, ct), _p(pd, ct),
          int(s), float(abstol), float(reltol), maxiter, int(bool(smoothing)), _mode(mode), _p(shp, C.c_int),
          _p(res, C.c_double), C.byref(iters), C.byref(mvps), C.byref(conv), C.byref(res0), C.byref(tol))
    return _run_simple("orc_idrs", A, b, x0, maxiter, call)


def _csc_pair(A):
    """(m, n, colptr, rowval, nzval, tcolptr, trowval, tnzval), 0-based Int64: A (a CSC of this module or any scipy sparse matrix, m x n) and
    A' as a SparseMatrixCSC of its own -- its column scatter is `mul!(y, adjoint(A), x)` of SparseArrays bit for bit (see orc_impl.inc)."""
    S = (A.to_scipy() if isinstance(A, CSC) else A).tocsc()
    S.sort_indices()
    St = S.T.tocsc()
    St.sort_indices()
    i64 = lambda a: np.ascontiguousarray(a, np.int64)   # noqa: E731
    return (S.shape[0], S.shape[1], i64(S.indptr), i64(S.indices), np.ascontiguousarray(S.data), i64(St.indptr), i64(St.indices),
            np.ascontiguousarray(St.data))


def lsqr(A, b, x0=None, *, damp=0.0, atol=None, btol=None, conlim=None, maxiter=None, mode="seq", shape=(1, 1)):
    """``lsqr!(x, A, b; damp, atol, btol, conlim, maxiter, log=true)`` / ``lsqr(A, b)`` when ``x0 is None`` -- src/lsqr.jl:69-81,10.  A: a CSC of
    this module or a scipy sparse matrix (m x n).  History keys as the reference's: resnorm, anorm, rnorm, cnorm."""
    m, n, cp, rv, nz, tcp, trv, tnz = _csc_pair(A)
    dtype = nz.dtype
    suf, ct = _suf(dtype)
    eps_s = _eps_sqrt(dtype)
    atol = eps_s if atol is None else atol                                    # :88
    btol = eps_s if btol is None else btol
    conlim = float(dtype.type(1) / dtype.type(eps_s)) if conlim is None else conlim   # :89
    maxiter = max(m, n) if maxiter is None else int(maxiter)                  # :70
    b = np.ascontiguousarray(b, dtype)
    x = np.zeros(n, dtype) if x0 is None else np.array(x0, dtype, copy=True)
    res, an, rn, cn = (np.zeros(max(maxiter, 1)) for _ in range(4))
    iters, mvps, mtvps = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    conv = C.c_int(0)
    shp = np.asarray(shape, np.int32)
    getattr(lib(), f"orc_lsqr_{suf}")(m, n, _p(cp, C.c_int64), _p(rv, C.c_int64), _p(nz, ct), _p(tcp, C.c_int64), _p(trv, C.c_int64), _p(tnz, ct), 0,
                                      _p(b, ct), _p(x, ct), float(damp), float(atol), float(btol), float(conlim), maxiter, _mode(mode),
                                      _p(shp, C.c_int), _p(res, C.c_double), _p(an, C.c_double), _p(rn, C.c_double), _p(cn, C.c_double),
                                      C.byref(iters), C.byref(mvps), C.byref(mtvps), C.byref(conv))
    k = iters.value
    return x, dict(iters=k, mvps=mvps.value, mtvps=mtvps.value, isconverged=bool(conv.value), resnorm=res[:k].copy(), anorm=an[:k].copy(),
                   rnorm=rn[:k].copy(), cnorm=cn[:k].copy())


def lsmr(A, b, x0=None, *, lam=0.0, atol=1e-6, btol=1e-6, conlim=1e8, maxiter=None, mode="seq", shape=(1, 1)):
    """``lsmr!(x, A, b; atol, btol, conlim, maxiter, λ, log=true)`` / ``lsmr(A, b)`` when ``x0 is None`` -- src/lsmr.jl:67-82,7.  History keys as
    the reference's: anorm, rnorm, cnorm."""
    m, n, cp, rv, nz, tcp, trv, tnz = _csc_pair(A)
    dtype = nz.dtype
    suf, ct = _suf(dtype)
    maxiter = max(m, n) if maxiter is None else int(maxiter)                  # :68
    b = np.ascontiguousarray(b, dtype)
    x = np.zeros(n, dtype) if x0 is None else np.array(x0, dtype, copy=True)
    an, rn, cn = (np.zeros(max(maxiter, 1)) for _ in range(3))
    iters, mvps, mtvps = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    conv = C.c_int(0)
    shp = np.asarray(shape, np.int32)
    getattr(lib(), f"orc_lsmr_{suf}")(m, n, _p(cp, C.c_int64), _p(rv, C.c_int64), _p(nz, ct), _p(tcp, C.c_int64), _p(trv, C.c_int64), _p(tnz, ct), 0,
                                      _p(b, ct), _p(x, ct), float(lam), float(atol), float(btol), float(conlim), maxiter, _mode(mode),
                                      _p(shp, C.c_int), _p(an, C.c_double), _p(rn, C.c_double), _p(cn, C.c_double),
                                      C.byref(iters), C.byref(mvps), C.byref(mtvps), C.byref(conv))
    k = iters.value
    return x, dict(iters=k, mvps=mvps.value, mtvps=mtvps.value, isconverged=bool(conv.value), anorm=an[:k].copy(), rnorm=rn[:k].copy(),
                   cnorm=cn[:k].copy())


def qmr(A, b, x0=None, *, abstol=0.0, reltol=None, maxiter=None, mode="seq", shape=(1, 1)):
    """``qmr!(x, A, b; abstol, reltol, maxiter, log=true)`` / ``qmr(A, b)`` (initially_zero) when ``x0 is None`` -- src/qmr.jl:256-297,210."""
    m, n, cp, rv, nz, tcp, trv, tnz = _csc_pair(A)
    dtype = nz.dtype
    suf, ct = _suf(dtype)
    reltol = _eps_sqrt(dtype) if reltol is None else reltol
    maxiter = n if maxiter is None else int(maxiter)
    b = np.ascontiguousarray(b, dtype)
    x = np.zeros(n, dtype) if x0 is None else np.array(x0, dtype, copy=True)
    res = np.zeros(max(maxiter, 1))
    iters = C.c_int64(0)
    conv = C.c_int(0)
    res0, tol = C.c_double(0), C.c_double(0)
    shp = np.asarray(shape, np.int32)
    getattr(lib(), f"orc_qmr_{suf}")(n, _p(cp, C.c_int64), _p(rv, C.c_int64), _p(nz, ct), _p(tcp, C.c_int64), _p(trv, C.c_int64), _p(tnz, ct), 0,
                                     _p(b, ct), _p(x, ct), float(abstol), float(reltol), maxiter, int(x0 is None), _mode(mode), _p(shp, C.c_int),
                                     _p(res, C.c_double), C.byref(iters), C.byref(conv), C.byref(res0), C.byref(tol))
    k = iters.value
    return x, dict(iters=k, mvps=0, isconverged=bool(conv.value), resnorm=res[:k].copy(), res0=res0.value, tol=tol.value)


def powm(A: CSC, x0, *, tol=None, maxiter=None, shift=0.0, inverse=False, mode="seq", shape=(1, 1)):
    """``powm!(B, x; tol, maxiter, shift, inverse, log=true)`` -- src/simple.jl:113-142.  Returns (λ, x, history); history["resnorm"] holds the
    residual norm of every iterate (the reference reserves the key but never pushes to it, :120-130)."""
    dtype = A.nzval.dtype
    suf, ct = _suf(dtype)
    n = A.n
    tol = float(np.finfo(dtype).eps) * n ** 3 if tol is None else tol        # :114
    maxiter = n if maxiter is None else int(maxiter)
    x = np.array(x0, dtype, copy=True)
    res = np.zeros(maxiter + 2)
    iters = C.c_int64(0)
    conv = C.c_int(0)
    theta = C.c_double(0)
    shp = np.asarray(shape, np.int32)
    getattr(lib(), f"orc_powm_{suf}")(n, _p(A.colptr, C.c_int64), _p(A.rowval, C.c_int64), _p(A.nzval, ct), A.index_base, _p(x, ct), float(tol),
                                      maxiter, _mode(mode), _p(shp, C.c_int), _p(res, C.c_double), C.byref(iters), C.byref(conv), C.byref(theta))
    th = dtype.type(theta.value)
    lam = dtype.type(shift) + (dtype.type(1) / th if inverse else th)         # transform_eigenvalue  :34
    k = iters.value
    return lam, x, dict(iters=k, mvps=k, isconverged=bool(conv.value), resnorm=res[:k].copy(), tol=tol)


_omp = None


def effective_cpus() -> int:
    """Host cores this process may actually use: min(affinity mask, cgroup v2 CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def omp_lib():
    """The OpenMP CPU baseline (oracle/mik_oracle_omp.c) -- bench.py's ``cpu_baseline_omp`` leg only."""
    global _omp
    if _omp is None:
        build()
        L = C.CDLL(os.path.join(_HERE, "_build", "libmik_oracle_omp.so"))
        L.orc_omp_threads.restype = C.c_int
        L.orc_omp_set_threads.argtypes = [C.c_int]
        L.orc_omp_set_threads(effective_cpus())
        L.orc_omp_cg_f64.argtypes = [C.c_int64, _i32p, _i32p, _f64p, _f64p, _f64p, C.c_double, C.c_double, C.c_int64, _f64p]
        L.orc_omp_cg_f64.restype = C.c_int64
        _omp = L
    return _omp


def omp_cg(A: CSC, b, *, abstol=0.0, reltol=None, maxiter=None):
    """cg(A, b) with all host cores (symmetric A: its CSC arrays are read as CSR).  Returns (x, iters, resnorm, threads)."""
    L = omp_lib()
    n = A.n
    rowptr = (A.colptr - A.index_base).astype(np.int32)
    col = (A.rowval - A.index_base).astype(np.int32)
    val = np.ascontiguousarray(A.nzval, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    x = np.empty(n)
    reltol = _eps_sqrt(np.float64) if reltol is None else reltol
    maxiter = n if maxiter is None else int(maxiter)
    res = np.zeros(max(maxiter, 1))
    it = L.orc_omp_cg_f64(n, _p(rowptr, C.c_int), _p(col, C.c_int), _p(val, C.c_double), _p(b, C.c_double), _p(x, C.c_double),
                          float(abstol), float(reltol), maxiter, _p(res, C.c_double))
    return x, int(it), res[:it].copy(), int(L.orc_omp_threads())
