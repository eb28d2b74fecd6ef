"""Host-side mirror of IterativeSolvers.jl's interface for the cg! / gmres! path, over libmik.so.

Julia is not available in this environment, so this module plays the role the Julia shim
(``julia/MIK.jl``) plays for a Julia host: it keeps the package's names, argument meaning,
defaults and error behaviour, and forwards every vector operation to the C ABI.  Julia's ``f!``
is spelled ``f_`` here.  Citations are file:line in the reference checkout (v0.9.4).

    reference                                   here
    ---------------------------------------     -----------------------------------------
    SparseMatrixCSC{T,Int} operator A            HipCSR            (mul_(y, A, x), size, eltype)
    Vector{T}                                    HipVector         (dot, norm, similar, zero ...)
    cg(A, b; ...) / cg!(x, A, b; ...)            cg / cg_                     src/cg.jl:162,209
    cg_iterator!(x, A, b, Pl; ...)               cg_iterator_                 src/cg.jl:120
    CGIterable / PCGIterable + iterate           CGIterable / PCGIterable     src/cg.jl:5-100
    CGStateVariables                             CGStateVariables             src/cg.jl:114
    gmres / gmres! / gmres_iterable!             gmres / gmres_ / gmres_iterable_   src/gmres.jl
    GMRESIterable + iterate                      GMRESIterable                src/gmres.jl:31-106
    orthogonalize_and_normalize!                 orthogonalize_and_normalize_ src/orthogonalize.jl
    DGKS / ClassicalGramSchmidt / ModifiedGS     same names                   src/orthogonalize.jl:4-7
    Identity                                     Identity                     src/common.jl:28-32
    ConvergenceHistory, niters, nprods, nrests   same names                   src/history.jl
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import numpy as np

from . import _lib
from ._lib import MikError, check, dtype_code, lib

_vp = C.c_void_p


def _scalar(dtype, value):
    """A host scalar of `dtype` and a pointer to it."""
    a = np.asarray([value], dtype=dtype)
    return a, a.ctypes.data_as(_vp)


# ==============================================================================================
# context, vectors, operator  (the L1 interface: docs/src/getting_started.md:25-30)
# ==============================================================================================
class HipContext:
    """One device + one HIP stream (``mik_ctx``)."""

    def __init__(self, device: int = 0):
        h = _vp()
        check(lib().mik_ctx_create(device, C.byref(h)), "mik_ctx_create", None)
        self.handle = h
        self.device = device

    def set_stream(self, hip_stream: Optional[int]):
        check(lib().mik_ctx_set_stream(self.handle, _vp(hip_stream)), "mik_ctx_set_stream", self.handle)

    def synchronize(self):
        check(lib().mik_ctx_synchronize(self.handle), "mik_ctx_synchronize", self.handle)

    def info(self) -> dict:
        """The machine behind the context as queried by ``mik_ctx_create`` and what the selection paths derive from it (``mik_ctx_info``)."""
        d = _lib.MikDeviceInfo()
        check(lib().mik_ctx_info(self.handle, C.byref(d)), "mik_ctx_info", self.handle)
        out = {k: getattr(d, k) for k, _ in d._fields_ if k != "reserved"}
        out["arch"] = d.arch.decode()
        return out

    def reduce_shape(self, dtype):
        w, l = C.c_int(), C.c_int()
        check(lib().mik_reduce_shape(dtype_code(dtype), C.byref(w), C.byref(l)), "mik_reduce_shape", self.handle)
        return w.value, l.value

    def spmv_dot_shape(self):
        w, l = C.c_int(), C.c_int()
        check(lib().mik_spmv_dot_shape(C.byref(w), C.byref(l)), "mik_spmv_dot_shape", self.handle)
        return w.value, l.value

    def set_tuning(self, key: int, value: int) -> None:
        """Development knob of THIS context (include/mik_dev.h); results never depend on it."""
        check(lib().mik_ctx_set_tuning(self.handle, int(key), int(value)), "mik_ctx_set_tuning", self.handle)

    def spmv_long_row(self) -> int:
        """Rows with more stored entries than this use the wave-shaped row sum (include/mik.h)."""
        t = C.c_int()
        check(lib().mik_spmv_long_row(C.byref(t)), "mik_spmv_long_row", self.handle)
        return t.value

    def spmv_long_group(self) -> int:
        """Long rows are summed in groups of this many consecutive entries per lane (include/mik.h)."""
        t = C.c_int()
        check(lib().mik_spmv_long_group(C.byref(t)), "mik_spmv_long_group", self.handle)
        return t.value

    def spmv_long_segment(self) -> int:
        """Rows with more stored entries than this are summed segment by segment (include/mik.h)."""
        t = C.c_int()
        check(lib().mik_spmv_long_segment(C.byref(t)), "mik_spmv_long_segment", self.handle)
        return t.value

    def cg_shape(self, dtype):
        """(Wd, Ld, W, L): reduction shapes of the fused CG step, as the oracle's `shape` argument."""
        return self.spmv_dot_shape() + self.reduce_shape(dtype)

    def close(self):
        if getattr(self, "handle", None):
            lib().mik_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx: Optional[HipContext] = None


def default_context() -> HipContext:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = HipContext(0)
    return _default_ctx


class HipVector:
    """Device n-vector of float64/float32: what ``similar``/``zero``/``copyto!``/``dot``/``norm``
    and in-place broadcast need (src/cg.jl:51,58-59,62,124,129-130,138)."""

    def __init__(self, n: int, dtype=np.float64, ctx: Optional[HipContext] = None, *, _ptr=None, _owner=None):
        self.ctx = ctx or default_context()
        self.n = int(n)
        self.dtype = np.dtype(dtype)
        self.code = dtype_code(dtype)
        if _ptr is None:
            p = _vp()
            check(lib().mik_malloc(self.ctx.handle, self.n * self.dtype.itemsize, C.byref(p)), "mik_malloc", self.ctx.handle)
            self.ptr = p.value
            self._owner = None
            self._owns = True
        else:
            self.ptr = int(_ptr)
            self._owner = _owner          # keeps the parent allocation alive
            self._owns = False

    # -- construction ---------------------------------------------------------------------------
    @staticmethod
    def from_numpy(a, ctx: Optional[HipContext] = None) -> "HipVector":
        a = np.ascontiguousarray(a)
        v = HipVector(a.size, a.dtype, ctx)
        v.copy_from_host(a)
        return v

    @staticmethod
    def wrap(ptr: int, n: int, dtype, ctx: Optional[HipContext] = None, owner=None) -> "HipVector":
        """Adopt an existing device allocation (e.g. ``torch_tensor.data_ptr()``)."""
        return HipVector(n, dtype, ctx, _ptr=ptr, _owner=owner)

    def view(self, offset: int, n: int) -> "HipVector":
        return HipVector(n, self.dtype, self.ctx, _ptr=self.ptr + offset * self.dtype.itemsize, _owner=self)

    def similar(self) -> "HipVector":
        return HipVector(self.n, self.dtype, self.ctx)

    def zero(self) -> "HipVector":
        return self.similar().fill_(0)

    def copy(self) -> "HipVector":
        return self.similar().copyto_(self)

    # -- host <-> device ------------------------------------------------------------------------
    def copy_from_host(self, a) -> "HipVector":
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if a.size != self.n:
            raise ValueError(f"DimensionMismatch: vector has length {self.n}, host array {a.size}")
        check(lib().mik_memcpy_h2d(self.ctx.handle, _vp(self.ptr), a.ctypes.data_as(_vp), a.nbytes), "mik_memcpy_h2d", self.ctx.handle)
        return self

    def to_numpy(self) -> np.ndarray:
        out = np.empty(self.n, self.dtype)
        check(lib().mik_memcpy_d2h(self.ctx.handle, out.ctypes.data_as(_vp), _vp(self.ptr), out.nbytes), "mik_memcpy_d2h", self.ctx.handle)
        return out

    # -- vector interface -----------------------------------------------------------------------
    def _same(self, other: "HipVector"):
        if other.n != self.n or other.dtype != self.dtype:
            raise ValueError(f"DimensionMismatch: {self.n}/{self.dtype} vs {other.n}/{other.dtype}")

    def fill_(self, value) -> "HipVector":                               # x .= value
        _, p = _scalar(self.dtype, value)
        check(lib().mik_fill(self.ctx.handle, self.code, self.n, p, _vp(self.ptr)), "mik_fill", self.ctx.handle)
        return self

    def copyto_(self, src: "HipVector") -> "HipVector":                  # copyto!(self, src)
        self._same(src)
        check(lib().mik_copy(self.ctx.handle, self.code, self.n, _vp(src.ptr), _vp(self.ptr)), "mik_copy", self.ctx.handle)
        return self

    def axpy_(self, alpha, x: "HipVector") -> "HipVector":               # self .+= alpha .* x
        self._same(x)
        _, p = _scalar(self.dtype, alpha)
        check(lib().mik_axpy(self.ctx.handle, self.code, self.n, p, _vp(x.ptr), _vp(self.ptr)), "mik_axpy", self.ctx.handle)
        return self

    def xpby_(self, x: "HipVector", beta) -> "HipVector":                # self .= x .+ beta .* self
        self._same(x)
        _, p = _scalar(self.dtype, beta)
        check(lib().mik_xpby(self.ctx.handle, self.code, self.n, _vp(x.ptr), p, _vp(self.ptr)), "mik_xpby", self.ctx.handle)
        return self

    def sub_(self, x: "HipVector") -> "HipVector":                       # self .-= x
        self._same(x)
        check(lib().mik_sub(self.ctx.handle, self.code, self.n, _vp(x.ptr), _vp(self.ptr)), "mik_sub", self.ctx.handle)
        return self

    def scal_(self, alpha) -> "HipVector":                               # self .*= alpha
        _, p = _scalar(self.dtype, alpha)
        check(lib().mik_scal(self.ctx.handle, self.code, self.n, p, _vp(self.ptr)), "mik_scal", self.ctx.handle)
        return self

    def __len__(self):
        return self.n

    def __del__(self):
        try:
            if self._owns and self.ptr and self.ctx.handle:
                lib().mik_free(self.ctx.handle, _vp(self.ptr))
                self.ptr = 0
        except Exception:
            pass


def dot(x: HipVector, y: HipVector):
    """``dot(x, y)`` -- src/cg.jl:55."""
    x._same(y)
    out = np.zeros(1, x.dtype)
    check(lib().mik_dot(x.ctx.handle, x.code, x.n, _vp(x.ptr), _vp(y.ptr), out.ctypes.data_as(_vp)), "mik_dot", x.ctx.handle)
    return out[0]


def norm(x: HipVector):
    """``norm(x)`` -- src/cg.jl:62."""
    out = np.zeros(1, x.dtype)
    check(lib().mik_nrm2(x.ctx.handle, x.code, x.n, _vp(x.ptr), out.ctypes.data_as(_vp)), "mik_nrm2", x.ctx.handle)
    return out[0]


def axpy_dot_(alpha, x: Optional[HipVector], y: HipVector, z: Optional[HipVector], hints: int = 0):
    """``y .+= alpha .* x`` (skipped when x is None) and, in the same sweep, ``dot(z, y)`` -- or ``norm(y)`` when z is
    None (src/minres.jl:104+107, :109+112)."""
    out = np.zeros(1, y.dtype)
    _, pa = _scalar(y.dtype, 0 if x is None else alpha)
    check(lib().mik_axpy_dot(y.ctx.handle, y.code, y.n, pa, _vp(x.ptr if x is not None else None), _vp(y.ptr),
                             _vp(z.ptr if z is not None else None), out.ctypes.data_as(_vp), int(hints)), "mik_axpy_dot", y.ctx.handle)
    return out[0]


def axpy2_nrm2_(alpha, u: HipVector, x: HipVector, c: HipVector, r: HipVector, hints: int = 0):
    """``x .+= alpha .* u; r .-= alpha .* c; norm(r)`` in one sweep (src/chebyshev.jl:51-54)."""
    out = np.zeros(1, x.dtype)
    _, pa = _scalar(x.dtype, alpha)
    check(lib().mik_axpy2_nrm2(x.ctx.handle, x.code, x.n, pa, _vp(u.ptr), _vp(x.ptr), _vp(c.ptr), _vp(r.ptr), out.ctypes.data_as(_vp), int(hints)),
          "mik_axpy2_nrm2", x.ctx.handle)
    return out[0]


class HipMatrix:
    """Device n x cols column-major block (the Krylov basis ``V`` of src/gmres.jl:7,13)."""

    def __init__(self, n: int, cols: int, dtype=np.float64, ctx: Optional[HipContext] = None):
        self.ctx = ctx or default_context()
        self.n, self.cols = int(n), int(cols)
        self.dtype = np.dtype(dtype)
        self.ld = (self.n + 63) // 64 * 64 or 64
        self.buf = HipVector(self.ld * self.cols, dtype, self.ctx)
        self.buf.fill_(0)

    @staticmethod
    def from_numpy(a, ctx=None) -> "HipMatrix":
        a = np.asarray(a)
        m = HipMatrix(a.shape[0], a.shape[1], a.dtype, ctx)
        for j in range(a.shape[1]):
            m.col(j).copy_from_host(a[:, j])
        return m

    def col(self, j: int) -> HipVector:
        return self.buf.view(j * self.ld, self.n)

    def to_numpy(self) -> np.ndarray:
        return np.stack([self.col(j).to_numpy() for j in range(self.cols)], axis=1)


class HipCSR:
    """The operator ``A``: a SparseMatrixCSC uploaded as device CSR (``mik_csr``).

    ``HipCSR(n_rows, n_cols, colptr, rowval, nzval, index_base=1)`` takes exactly the fields of a
    Julia ``SparseMatrixCSC{T,Int}`` (test/laplace_matrix.jl:12); Int32 index arrays -- ``SparseMatrixCSC{T,Int32}``,
    test/gmres.jl:38 -- go through ``mik_csr_create_i32``."""

    def __init__(self, n_rows, n_cols, ptr, idx, val, *, index_base=1, is_csc=True, ctx: Optional[HipContext] = None):
        self.ctx = ctx or default_context()
        val = np.ascontiguousarray(val)
        self.dtype = val.dtype
        self.code = dtype_code(val.dtype)
        self.n_rows, self.n_cols, self.nnz = int(n_rows), int(n_cols), int(val.size)
        h = _vp()
        if np.asarray(ptr).dtype == np.int32 and np.asarray(idx).dtype == np.int32:
            ptr, idx = np.ascontiguousarray(ptr), np.ascontiguousarray(idx)
            check(lib().mik_csr_create_i32(self.ctx.handle, self.code, self.n_rows, self.n_cols, self.nnz,
                                           ptr.ctypes.data_as(C.POINTER(C.c_int32)), idx.ctypes.data_as(C.POINTER(C.c_int32)),
                                           val.ctypes.data_as(_vp), int(index_base), int(bool(is_csc)), C.byref(h)),
                  "mik_csr_create_i32", self.ctx.handle)
        else:
            ptr = np.ascontiguousarray(ptr, np.int64)
            idx = np.ascontiguousarray(idx, np.int64)
            check(lib().mik_csr_create(self.ctx.handle, self.code, self.n_rows, self.n_cols, self.nnz,
                                       ptr.ctypes.data_as(C.POINTER(C.c_int64)), idx.ctypes.data_as(C.POINTER(C.c_int64)),
                                       val.ctypes.data_as(_vp), int(index_base), int(bool(is_csc)), C.byref(h)),
                  "mik_csr_create", self.ctx.handle)
        self.handle = h

    @classmethod
    def from_device(cls, n_rows, n_cols, nnz, ptr_dev: int, idx_dev: int, val_dev: int, dtype, *, index_base=1, is_csc=True,
                    ctx: Optional[HipContext] = None) -> "HipCSR":
        """The same operator from arrays that already live in device memory (raw device pointers of an Int64 ``ptr`` /
        ``idx`` and a ``val`` array of ``dtype`` -- a ROCSparseMatrixCSC, torch tensors): ``mik_csr_create`` detects the
        placement and starts its device-side pipeline from them; no host copy.  The arrays are only read during the call."""
        self = cls.__new__(cls)
        self.ctx = ctx or default_context()
        self.dtype = np.dtype(dtype)
        self.code = dtype_code(self.dtype)
        self.n_rows, self.n_cols, self.nnz = int(n_rows), int(n_cols), int(nnz)
        h = _vp()
        check(lib().mik_csr_create(self.ctx.handle, self.code, self.n_rows, self.n_cols, self.nnz,
                                   C.cast(_vp(int(ptr_dev)), C.POINTER(C.c_int64)), C.cast(_vp(int(idx_dev)), C.POINTER(C.c_int64)),
                                   _vp(int(val_dev)), int(index_base), int(bool(is_csc)), C.byref(h)),
              "mik_csr_create", self.ctx.handle)
        self.handle = h
        return self

    @staticmethod
    def from_scipy(m, ctx=None) -> "HipCSR":
        m = m.tocsc()
        m.sort_indices()
        return HipCSR(m.shape[0], m.shape[1], m.indptr, m.indices, m.data, index_base=0, is_csc=True, ctx=ctx)

    def compact(self) -> bool:
        """Release the CSR arrays of an operator that runs on one of the sliced layouts (``mik_csr_compact``); False (nothing
        released) if the operator needs them."""
        code = lib().mik_csr_compact(self.handle)
        if code == 5:
            return False
        check(code, "mik_csr_compact", self.ctx.handle)
        return True

    def size(self, d: Optional[int] = None):
        return (self.n_rows, self.n_cols) if d is None else (self.n_rows, self.n_cols)[d - 1]

    def eltype(self):
        return self.dtype

    def __matmul__(self, x: HipVector) -> HipVector:                     # A * v
        y = HipVector(self.n_rows, self.dtype, self.ctx)
        return mul_(y, self, x)

    LAYOUTS = ("csr-rowblock", "jagged-slices", "(retired-2)", "(retired)", "sliced-ell+slice-offsets+row-masks",
               "slice-offsets+slice-values+row-masks", "wide-slice-values+row-masks")

    def layout(self) -> str:
        """Device layout ``mul_`` uses for this operator (``mik_csr_layout``); results do not depend on it."""
        out = C.c_int()
        check(lib().mik_csr_layout(self.handle, C.byref(out)), "mik_csr_layout", self.ctx.handle)
        return self.LAYOUTS[out.value]

    def set_layout(self, layout: str = "auto") -> "HipCSR":
        """``"csr"``: run ``mul_`` and the iterables created afterwards on the plain CSR arrays; ``"auto"``: the layout chosen at
        upload (``mik_csr_set_layout``).  Results are bit-identical either way."""
        check(lib().mik_csr_set_layout(self.handle, {"csr": 0, "auto": -1}[layout]), "mik_csr_set_layout", self.ctx.handle)
        return self

    def spmv_kernel(self) -> str:
        """Name of the SpMV kernel ``mul_`` launches for this operator now (``mik_spmv_kernel``; for profiles / the bench line)."""
        buf = C.create_string_buffer(64)
        check(lib().mik_spmv_kernel(self.handle, buf, 64), "mik_spmv_kernel", self.ctx.handle)
        return buf.value.decode()

    def spmv_stored_bytes(self) -> int:
        """Bytes one ``mul_`` launch actually streams in the active layout: operator data + x once + y once."""
        out = C.c_int64()
        check(lib().mik_csr_stored_bytes(self.handle, C.byref(out)), "mik_csr_stored_bytes", self.ctx.handle)
        s = np.dtype(self.dtype).itemsize
        return int(out.value) + (self.n_cols + self.n_rows) * s

    def spmv_algorithmic_bytes(self) -> int:
        """nnz*(s+4) + (n+1)*4 + 2*n*s."""
        s = self.dtype.itemsize
        return self.nnz * (s + 4) + (self.n_rows + 1) * 4 + self.n_cols * s + self.n_rows * s

    def time_spmv(self, x: HipVector, y: HipVector, reps: int = 20, fused_dot: bool = False) -> float:
        ms = C.c_double()
        check(lib().mik_time_spmv(self.ctx.handle, self.handle, _vp(x.ptr), _vp(y.ptr), int(fused_dot), reps, C.byref(ms)),
              "mik_time_spmv", self.ctx.handle)
        return ms.value

    def __del__(self):
        try:
            if getattr(self, "handle", None) and self.ctx.handle:
                lib().mik_csr_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def mul_(y: HipVector, A, x: HipVector) -> HipVector:
    """``mul!(y, A, x)`` -- src/cg.jl:54."""
    if x.n != A.n_cols or y.n != A.n_rows or x.dtype != A.dtype or y.dtype != A.dtype:
        raise ValueError("DimensionMismatch in mul_(y, A, x)")
    if isinstance(A, LinearOperator):
        A.mul(y, x)
        return y
    check(lib().mik_spmv(A.ctx.handle, A.handle, _vp(x.ptr), _vp(y.ptr)), "mik_spmv", A.ctx.handle)
    return y


class LinearOperator:
    """Any operator the reference accepts: something with ``mul!(y, A, x)``, ``eltype`` and ``size``
    (docs/src/getting_started.md:25-30) -- e.g. the LinearMap of test/gmres.jl:59-66.  ``mul(y, x)`` receives two device
    vectors and must leave ``y = A * x`` ordered on the context's stream (any composition of this module's calls does)."""

    def __init__(self, n: int, dtype, mul, ctx: Optional[HipContext] = None):
        self.n_rows = self.n_cols = int(n)
        self.dtype = np.dtype(dtype)
        self.mul = mul
        self.ctx = ctx or default_context()

    def size(self, d: Optional[int] = None):
        return (self.n_rows, self.n_cols) if d is None else self.n_rows

    def eltype(self):
        return self.dtype

    def __matmul__(self, x: HipVector) -> HipVector:
        return mul_(HipVector(self.n_rows, self.dtype, self.ctx), self, x)


class _Bound:
    """ctypes structures + callbacks that carry a Python operator / preconditioner across the C ABI (kept alive by the
    iterable that owns them).  An exception inside a callback is parked and re-raised by the caller of the C entry."""

    def __init__(self, ctx: HipContext, n: int, dtype):
        self.ctx, self.n, self.dtype = ctx, int(n), np.dtype(dtype)
        self.error = None
        self.keep = []

    def _vec(self, ptr):
        return HipVector.wrap(ptr, self.n, self.dtype, self.ctx)

    def operator(self, A) -> "_lib.MikOperator":
        if isinstance(A, HipCSR):
            return _lib.MikOperator(A.code, A.n_rows, A.handle, _lib.MUL_FN(), None)

        def cb(_user, x, y):
            try:
                A.mul(self._vec(y), self._vec(x)) if isinstance(A, LinearOperator) else mul_(self._vec(y), A, self._vec(x))
                return 0
            except BaseException as e:               # never let an exception cross the C frame
                self.error = e
                return 1
        fn = _lib.MUL_FN(cb)
        self.keep.append(fn)
        return _lib.MikOperator(dtype_code(self.dtype), self.n, None, fn, None)

    def precond(self, P):
        """None for Identity(); else a MikPrecond (diagonal fused into the sweeps, or ``ldiv_`` as a callback)."""
        if P is None or isinstance(P, Identity):
            return None
        if isinstance(P, JacobiPrec):
            return _lib.MikPrecond(P.diagonal.ptr, _lib.LDIV_FN(), None)
        if not callable(getattr(P, "ldiv_", None)):
            raise MikError(5, "preconditioner", "needs ldiv_(y, x) (docs/src/preconditioning.md:5-14)")

        def cb(_user, y, x):
            try:
                P.ldiv_(self._vec(y), self._vec(x))
                return 0
            except BaseException as e:
                self.error = e
                return 1
        fn = _lib.LDIV_FN(cb)
        self.keep.append(fn)
        return _lib.MikPrecond(None, fn, None)

    def check(self, code, where):
        if code and self.error is not None:
            err, self.error = self.error, None
            raise err
        check(code, where, self.ctx.handle)


# ==============================================================================================
# common.jl / preconditioners
# ==============================================================================================
class Identity:
    """No-op preconditioner -- src/common.jl:28-32."""

    def ldiv_(self, y: HipVector, x: Optional[HipVector] = None):
        if x is None:
            return y                                                     # ldiv!(::Identity, x) = x
        return y.copyto_(x)                                              # ldiv!(y, ::Identity, x)


class JacobiPrec:
    """``ldiv!(y, P::JacobiPrec, x) = y .= x ./ P.diagonal`` -- the fixture of test/cg.jl:14-18."""

    def __init__(self, diagonal: HipVector):
        self.diagonal = diagonal

    def ldiv_(self, y: HipVector, x: Optional[HipVector] = None):
        x = y if x is None else x
        check(lib().mik_divide(y.ctx.handle, y.code, y.n, _vp(x.ptr), _vp(self.diagonal.ptr), _vp(y.ptr)), "mik_divide", y.ctx.handle)
        return y


def zerox(A: HipCSR, b: HipVector) -> HipVector:
    """``zerox(A, b)`` -- src/common.jl:18-23."""
    return HipVector(A.size(2), np.result_type(A.dtype, b.dtype), b.ctx).fill_(0)


# ==============================================================================================
# history.jl
# ==============================================================================================
class ConvergenceHistory:
    """``ConvergenceHistory`` -- src/history.jl:54-66; ``partial=True`` stores nothing per iteration."""

    def __init__(self, partial: bool = False, restart=None):
        self.mvps = 0
        self.mtvps = 0
        self.iters = 0
        self.restart = restart
        self.isconverged = False
        self.partial = partial
        self.data = {}

    def __getitem__(self, key):
        return self.data[key]

    def __setitem__(self, key, val):
        self.data[key] = val

    def reserve_(self, key, n):                                          # src/history.jl:181-201
        if not self.partial:
            self.data[key] = np.empty(int(n), np.float64)

    def nextiter_(self, mvps=0, mtvps=0):                                # src/history.jl:211-215
        self.iters += 1
        self.mvps += mvps
        self.mtvps += mtvps

    def push_(self, key, val):                                           # src/history.jl:139-142
        if self.partial:
            self.data[key] = val
        else:
            self.data[key][self.iters - 1] = val

    def setconv(self, val: bool):
        self.isconverged = bool(val)

    def shrink_(self):                                                   # src/history.jl:196-208
        if not self.partial:
            for k, v in list(self.data.items()):
                if isinstance(v, np.ndarray):
                    self.data[k] = v[: self.iters].copy()


def niters(ch: ConvergenceHistory) -> int:
    return ch.iters


def nprods(ch: ConvergenceHistory) -> int:
    return ch.mvps + ch.mtvps


def nrests(ch: ConvergenceHistory) -> int:
    return int(math.ceil(ch.iters / ch.restart))


# ==============================================================================================
# cg.jl
# ==============================================================================================
class CGStateVariables:
    """``CGStateVariables(u, r, c)`` -- src/cg.jl:114-118."""

    def __init__(self, u: HipVector, r: HipVector, c: HipVector):
        self.u, self.r, self.c = u, r, c


class CGIterable:
    """``CGIterable`` (src/cg.jl:5-16) / ``PCGIterable`` with a diagonal ``Pl`` (src/cg.jl:18-30),
    driven by the fused device step (``mik_cg``)."""

    def __init__(self, A: HipCSR, x: HipVector, b: HipVector, statevars: CGStateVariables, Pl, *, abstol, reltol,
                 maxiter, initially_zero):
        self.A, self.x, self.b = A, x, b
        self.u, self.r, self.c = statevars.u, statevars.r, statevars.c
        self.Pl = Pl
        for v in (x, b, self.u, self.r, self.c):
            if v.n != A.n_rows or v.dtype != A.dtype:
                raise ValueError("DimensionMismatch in cg_iterator_")
        h = _vp()
        if isinstance(A, HipCSR) and isinstance(Pl, (Identity, JacobiPrec)):
            diag = Pl.diagonal.ptr if isinstance(Pl, JacobiPrec) else None
            check(lib().mik_cg_create(A.ctx.handle, A.handle, _vp(x.ptr), _vp(b.ptr), _vp(self.u.ptr), _vp(self.r.ptr),
                                      _vp(self.c.ptr), _vp(diag), float(abstol), float(reltol), int(maxiter),
                                      int(bool(initially_zero)), C.byref(h)), "mik_cg_create", A.ctx.handle)
        else:
            # any operator with mul!, any Pl with ldiv! (mik_cg_create_op): the fused sweeps stay, A / Pl come back as callbacks
            self._bound = _Bound(A.ctx, A.n_rows, A.dtype)
            op, pl = self._bound.operator(A), self._bound.precond(Pl)
            self._bound.check(lib().mik_cg_create_op(A.ctx.handle, C.byref(op), C.byref(pl) if pl is not None else None, _vp(x.ptr), _vp(b.ptr),
                                                     _vp(self.u.ptr), _vp(self.r.ptr), _vp(self.c.ptr), float(abstol), float(reltol), int(maxiter),
                                                     int(bool(initially_zero)), C.byref(h)), "mik_cg_create_op")
        self.handle = h
        self.maxiter = int(maxiter)
        self._refresh()

    def _refresh(self):
        res, prev, tol = C.c_double(), C.c_double(), C.c_double()
        mx, mv = C.c_int64(), C.c_int64()
        conv = C.c_int()
        check(lib().mik_cg_state(self.handle, C.byref(res), C.byref(prev), C.byref(tol), C.byref(mx), C.byref(mv), C.byref(conv)),
              "mik_cg_state", self.A.ctx.handle)
        self.residual, self.prev_residual, self.tol = res.value, prev.value, tol.value
        self.mv_products = mv.value

    def _check(self, code, where):
        b = getattr(self, "_bound", None)
        b.check(code, where) if b is not None else check(code, where, self.A.ctx.handle)

    def converged(self) -> bool:                                         # src/cg.jl:32
        return self.residual <= self.tol

    def start(self) -> int:                                              # src/cg.jl:34
        return 0

    def done(self, iteration: int) -> bool:                              # src/cg.jl:36
        return iteration >= self.maxiter or self.converged()

    def iterate(self, iteration: Optional[int] = None):
        """``iterate(it, iteration)`` -> ``None`` or ``(residual, iteration + 1)`` -- src/cg.jl:43-66."""
        iteration = self.start() if iteration is None else iteration
        res = C.c_double()
        done = C.c_int()
        self._check(lib().mik_cg_iterate(self.handle, int(iteration), C.byref(res), C.byref(done)), "mik_cg_iterate")
        if done.value:
            return None
        self.prev_residual, self.residual = self.residual, res.value
        self.mv_products += 1
        return self.residual, iteration + 1

    def fused_x(self) -> bool:
        """True if ``x .+= alpha .* u`` rides on the sweep over u that opens the next step (``mik_cg_fused_x``)."""
        out = C.c_int()
        self._check(lib().mik_cg_fused_x(self.handle, C.byref(out)), "mik_cg_fused_x")
        return bool(out.value)

    def iterate_many(self, iteration: int, max_steps: int) -> np.ndarray:
        """Up to ``max_steps`` ``iterate`` calls with one host synchronisation; returns the residuals."""
        out = np.empty(max(int(max_steps), 1), np.float64)
        nd = C.c_int64()
        self._check(lib().mik_cg_iterate_many(self.handle, int(iteration), int(max_steps), out.ctypes.data_as(C.POINTER(C.c_double)),
                                              C.byref(nd)), "mik_cg_iterate_many")
        self._refresh()
        return out[: nd.value].copy()

    def profile(self, enable: int = -1):
        """In-loop HIP-event timing of the SpMV launch: returns (total_ms, launches) gathered so far,
        then switches timing on (1, resets totals) / off (0) / leaves it (-1)."""
        ms, cnt = C.c_double(), C.c_int64()
        check(lib().mik_cg_profile(self.handle, int(enable), C.byref(ms), C.byref(cnt)), "mik_cg_profile", self.A.ctx.handle)
        return ms.value, cnt.value

    def profile_kernels(self):
        """After ``profile(2)``: {kernel: (total_ms, launches)} for the SpMV, ``u = r + beta u`` and the x / r update."""
        ms = (C.c_double * 3)()
        cnt = (C.c_int64 * 3)()
        check(lib().mik_cg_profile_kernels(self.handle, ms, cnt), "mik_cg_profile_kernels", self.A.ctx.handle)
        return {k: (ms[i], cnt[i]) for i, k in enumerate(("spmv", "xpby", "update"))}

    def __iter__(self):
        iteration = self.start()
        while True:
            nxt = self.iterate(iteration)
            if nxt is None:
                return
            residual, iteration = nxt
            yield residual

    def __del__(self):
        try:
            if getattr(self, "handle", None) and self.A.ctx.handle:
                lib().mik_cg_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


PCGIterable = CGIterable   # same handle; the diagonal Pl selects the src/cg.jl:72-100 branch


class GenericCGIterable:
    """The reference's ``iterate(::CGIterable)`` / ``iterate(::PCGIterable)`` spelled with the L1
    entry points only (mul_, dot, norm, broadcast forms) -- i.e. what the unmodified package does
    with a device vector type (SURVEY.md section 8b plug point 1).  Works with any ``Pl`` that has
    ``ldiv_(y, x)``."""

    def __init__(self, A, x, b, statevars, Pl, *, abstol, reltol, maxiter, initially_zero):
        self.A, self.x, self.Pl = A, x, Pl
        self.u, self.r, self.c = statevars.u, statevars.r, statevars.c
        T = x.dtype.type
        self.u.fill_(0)                                                  # src/cg.jl:129
        self.r.copyto_(b)                                                # :130
        if initially_zero:
            self.mv_products = 0
        else:
            self.mv_products = 1
            mul_(self.c, A, x)                                           # :137
            self.r.sub_(self.c)                                          # :138
        self.residual = norm(self.r)                                     # :140
        self.tol = max(T(reltol) * self.residual, T(abstol))             # :141
        self.prev_residual = T(1)
        self.rho = T(1)
        self.maxiter = int(maxiter)
        self.pcg = not isinstance(Pl, Identity)

    def converged(self):
        return self.residual <= self.tol

    def start(self):
        return 0

    def done(self, iteration):
        return iteration >= self.maxiter or self.converged()

    def iterate(self, iteration=None):
        iteration = 0 if iteration is None else iteration
        if self.done(iteration):
            return None
        if not self.pcg:
            beta = self.residual * self.residual / (self.prev_residual * self.prev_residual)   # :50
            self.u.xpby_(self.r, beta)                                   # :51
            mul_(self.c, self.A, self.u)                                 # :54
            alpha = self.residual * self.residual / dot(self.u, self.c)  # :55
        else:
            self.Pl.ldiv_(self.c, self.r)                                # :79
            rho_prev = self.rho
            self.rho = dot(self.c, self.r)                               # :82
            beta = self.rho / rho_prev                                   # :85
            self.u.xpby_(self.c, beta)                                   # :86
            mul_(self.c, self.A, self.u)                                 # :89
            alpha = self.rho / dot(self.u, self.c)                       # :90
        self.x.axpy_(alpha, self.u)                                      # :58
        self.r.axpy_(-alpha, self.c)                                     # :59
        self.prev_residual = self.residual
        self.residual = norm(self.r)                                     # :62
        self.mv_products += 1
        return self.residual, iteration + 1

    def __iter__(self):
        iteration = 0
        while True:
            nxt = self.iterate(iteration)
            if nxt is None:
                return
            residual, iteration = nxt
            yield residual


def _default_reltol(b: HipVector) -> float:
    return float(np.sqrt(np.finfo(b.dtype).eps))


def cg_iterator_(x: HipVector, A: HipCSR, b: HipVector, Pl=None, *, abstol=0.0, reltol=None, maxiter=None,
                 statevars: Optional[CGStateVariables] = None, initially_zero: bool = False, fused: bool = True):
    """``cg_iterator!(x, A, b, Pl; ...)`` -- src/cg.jl:120-155."""
    Pl = Identity() if Pl is None else Pl
    reltol = _default_reltol(b) if reltol is None else reltol
    maxiter = A.size(2) if maxiter is None else maxiter
    if statevars is None:
        statevars = CGStateVariables(x.zero(), x.similar(), x.similar())  # :124
    kw = dict(abstol=abstol, reltol=reltol, maxiter=maxiter, initially_zero=initially_zero)
    if fused:
        return CGIterable(A, x, b, statevars, Pl, **kw)
    return GenericCGIterable(A, x, b, statevars, Pl, **kw)


def cg_(x: HipVector, A: HipCSR, b: HipVector, *, abstol=0.0, reltol=None, maxiter=None, log: bool = False,
        statevars: Optional[CGStateVariables] = None, verbose: bool = False, Pl=None, **kwargs):
    """``cg!(x, A, b; ...)`` -> ``x`` or ``(x, history)`` -- src/cg.jl:209-242."""
    reltol = _default_reltol(b) if reltol is None else reltol
    maxiter = A.size(2) if maxiter is None else maxiter
    history = ConvergenceHistory(partial=not log)                         # :218
    history["abstol"] = abstol
    history["reltol"] = reltol
    if log:
        history.reserve_("resnorm", maxiter + 1)                          # :221
    iterable = cg_iterator_(x, A, b, Pl, abstol=abstol, reltol=reltol, maxiter=maxiter, statevars=statevars, **kwargs)
    if log:
        history.mvps = iterable.mv_products                               # :227
    for iteration, _item in enumerate(iterable, start=1):                 # :229
        if log:
            history.nextiter_(mvps=1)                                     # :231
            history.push_("resnorm", iterable.residual)                   # :232
        if verbose:
            print("%3d\t%1.2e" % (iteration, iterable.residual))
    if verbose:
        print()
    if log:
        history.setconv(iterable.converged())                             # :238
        history.shrink_()                                                 # :239
    return (iterable.x, history) if log else iterable.x


def cg(A: HipCSR, b: HipVector, **kwargs):
    """``cg(A, b; ...)`` -- src/cg.jl:162."""
    return cg_(zerox(A, b), A, b, initially_zero=True, **kwargs)


# ==============================================================================================
# orthogonalize.jl
# ==============================================================================================
class OrthogonalizationMethod:
    code = None


class DGKS(OrthogonalizationMethod):
    code = _lib.MIK_DGKS


class ClassicalGramSchmidt(OrthogonalizationMethod):
    code = _lib.MIK_CGS


class ModifiedGramSchmidt(OrthogonalizationMethod):
    code = _lib.MIK_MGS


def orthogonalize_and_normalize_(V, k: int, w: HipVector, h: np.ndarray, method=None):
    """``orthogonalize_and_normalize!(view(V, :, 1:k), w, h, method)`` -> nrm -- src/orthogonalize.jl:13-79.
    ``h`` is a host array of at least k entries, written in place.  ``V`` may also be a list of ``HipVector`` s -- the
    vector-of-vectors method of src/orthogonalize.jl:53-65 (ModifiedGramSchmidt only, like the reference)."""
    method = ModifiedGramSchmidt() if method is None else method           # :10-11
    if isinstance(V, (list, tuple)):
        if not isinstance(method, ModifiedGramSchmidt):
            raise TypeError("MethodError: orthogonalize_and_normalize!(::Vector{Vector}, ...) is defined for ModifiedGramSchmidt only")
        if k > len(V) or any(v.n != w.n or v.dtype != w.dtype for v in V[:k]):
            raise ValueError("DimensionMismatch in orthogonalize_and_normalize_")
        ptrs = (_vp * max(k, 1))(*[_vp(v.ptr) for v in V[:k]])
        hh = np.zeros(max(k, 1), w.dtype)
        nrm = np.zeros(1, w.dtype)
        check(lib().mik_orthogonalize_vectors(w.ctx.handle, dtype_code(w.dtype), w.n, int(k), ptrs, _vp(w.ptr), hh.ctypes.data_as(_vp),
                                              nrm.ctypes.data_as(_vp)), "mik_orthogonalize_vectors", w.ctx.handle)
        h[:k] = hh[:k]
        return nrm[0]
    if w.n != V.n or w.dtype != V.dtype or k > V.cols:
        raise ValueError("DimensionMismatch in orthogonalize_and_normalize_")
    hh = np.zeros(max(k, 1), V.dtype)
    nrm = np.zeros(1, V.dtype)
    check(lib().mik_orthogonalize(V.ctx.handle, dtype_code(V.dtype), V.n, int(k), _vp(V.buf.ptr), V.ld, _vp(w.ptr),
                                  hh.ctypes.data_as(_vp), nrm.ctypes.data_as(_vp), method.code), "mik_orthogonalize", V.ctx.handle)
    h[:k] = hh[:k]
    return nrm[0]


def gemv_n_(y: HipVector, V: HipMatrix, k: int, c: np.ndarray, alpha=1.0, col0: int = 0) -> HipVector:
    """``mul!(y, view(V, :, col0+1 : col0+k), c, alpha, 1)`` -- src/gmres.jl:275, src/bicgstabl.jl:127-129."""
    c = np.ascontiguousarray(c[:k], V.dtype)
    _, pa = _scalar(V.dtype, alpha)
    check(lib().mik_gemv_n(V.ctx.handle, dtype_code(V.dtype), V.n, int(k), _vp(V.col(col0).ptr), V.ld, c.ctypes.data_as(_vp), pa, _vp(y.ptr)),
          "mik_gemv_n", V.ctx.handle)
    return y


def gemv_t_(V: HipMatrix, k: int, w: HipVector, col0: int = 0) -> np.ndarray:
    """``adjoint(view(V, :, col0+1 : col0+k)) * w`` -- src/orthogonalize.jl:15, one column of src/bicgstabl.jl:121."""
    h = np.zeros(max(k, 1), V.dtype)
    check(lib().mik_gemv_t(V.ctx.handle, dtype_code(V.dtype), V.n, int(k), _vp(V.col(col0).ptr), V.ld, _vp(w.ptr), h.ctypes.data_as(_vp)),
          "mik_gemv_t", V.ctx.handle)
    return h[:k]


def gram_(V: HipMatrix, k: int, col0: int = 0) -> np.ndarray:
    """``adjoint(V[:, col0+1 : col0+k]) * V[:, col0+1 : col0+k]`` in one pass (k <= 5) -- src/bicgstabl.jl:120."""
    M = np.zeros((k, k), V.dtype, order="F")
    check(lib().mik_gram(V.ctx.handle, dtype_code(V.dtype), V.n, int(k), _vp(V.col(col0).ptr), V.ld, M.ctypes.data_as(_vp)),
          "mik_gram", V.ctx.handle)
    return M


def lu_solve_(A: np.ndarray, b: np.ndarray) -> np.ndarray:
    """``ldiv!(x, lu!(A), b)`` for a small dense host matrix -- src/bicgstabl.jl:124-125.  Overwrites both."""
    if not A.flags.f_contiguous or A.dtype != b.dtype or A.shape[0] != A.shape[1] or b.size != A.shape[0]:
        raise ValueError("lu_solve_: A must be a square F-ordered array, b a vector of the same dtype")
    code = lib().mik_lu_solve(dtype_code(A.dtype), A.ctypes.data_as(_vp), A.shape[0], A.shape[0], b.ctypes.data_as(_vp))
    if code == 8:                                                            # MIK_ERR_SINGULAR
        raise np.linalg.LinAlgError("SingularException")
    check(code, "mik_lu_solve", None)
    return b


def hessenberg_ldiv_(H: np.ndarray, rhs: np.ndarray):
    """``ldiv!(FastHessenberg(H), rhs)`` (host) -- src/hessenberg.jl:15-46.  ``H`` must be Fortran-ordered."""
    if not H.flags.f_contiguous or H.dtype != rhs.dtype or H.shape[0] != H.shape[1] + 1 or rhs.size != H.shape[0]:
        raise ValueError("hessenberg_ldiv_: H must be an F-ordered (m+1) x m array, rhs of length m+1, same dtype")
    check(lib().mik_hessenberg_ldiv(dtype_code(H.dtype), H.ctypes.data_as(_vp), H.shape[0], H.shape[1], rhs.ctypes.data_as(_vp)),
          "mik_hessenberg_ldiv", None)
    return rhs


# ==============================================================================================
# gmres.jl
# ==============================================================================================
class GMRESIterable:
    """``GMRESIterable`` -- src/gmres.jl:31-49, driven by ``mik_gmres`` (device Krylov basis, host Hessenberg)."""

    def __init__(self, x: HipVector, A: HipCSR, b: HipVector, *, abstol, reltol, restart, maxiter, initially_zero, orth_meth,
                 Pl=None, Pr=None):
        if x.n != A.n_rows or b.n != A.n_rows or x.dtype != A.dtype or b.dtype != A.dtype:
            raise ValueError("DimensionMismatch in gmres_iterable_")
        self.A, self.x, self.b = A, x, b
        self.Pl, self.Pr = Pl, Pr
        self.restart, self.maxiter = int(restart), int(maxiter)
        self.orth_meth = orth_meth
        h = _vp()
        simple = lambda P: P is None or isinstance(P, (Identity, JacobiPrec))
        if isinstance(A, HipCSR) and simple(Pl) and simple(Pr):
            pl = Pl.diagonal.ptr if isinstance(Pl, JacobiPrec) else None
            pr = Pr.diagonal.ptr if isinstance(Pr, JacobiPrec) else None
            check(lib().mik_gmres_create(A.ctx.handle, A.handle, _vp(x.ptr), _vp(b.ptr), _vp(pl), _vp(pr), float(abstol), float(reltol), int(restart),
                                         int(maxiter), int(bool(initially_zero)), orth_meth.code, C.byref(h)), "mik_gmres_create", A.ctx.handle)
        else:
            # any operator with mul!, any Pl / Pr with ldiv! (mik_gmres_create_op), e.g. test/gmres.jl:28-35 and :59-66
            self._bound = _Bound(A.ctx, A.n_rows, A.dtype)
            op, pl, pr = self._bound.operator(A), self._bound.precond(Pl), self._bound.precond(Pr)
            self._bound.check(lib().mik_gmres_create_op(A.ctx.handle, C.byref(op), C.byref(pl) if pl is not None else None,
                                                        C.byref(pr) if pr is not None else None, _vp(x.ptr), _vp(b.ptr), float(abstol), float(reltol),
                                                        int(restart), int(maxiter), int(bool(initially_zero)), orth_meth.code, C.byref(h)),
                              "mik_gmres_create_op")
        self.handle = h
        self._refresh()

    def _refresh(self):
        res, tol, beta = C.c_double(), C.c_double(), C.c_double()
        k = C.c_int()
        mv = C.c_int64()
        conv = C.c_int()
        check(lib().mik_gmres_state(self.handle, C.byref(res), C.byref(tol), C.byref(beta), C.byref(k), C.byref(mv), C.byref(conv)),
              "mik_gmres_state", self.A.ctx.handle)
        self.residual_current, self.tol, self.beta, self.k, self.mv_products = res.value, tol.value, beta.value, k.value, mv.value

    def converged(self) -> bool:                                         # src/gmres.jl:51
        return self.residual_current <= self.tol

    def start(self) -> int:
        return 0

    def done(self, iteration: int) -> bool:                              # src/gmres.jl:55
        return iteration >= self.maxiter or self.converged()

    def iterate(self, iteration: Optional[int] = None):
        """``iterate(g, iteration)`` -- src/gmres.jl:57-106."""
        iteration = 0 if iteration is None else iteration
        res = C.c_double()
        done = C.c_int()
        code = lib().mik_gmres_iterate(self.handle, int(iteration), C.byref(res), C.byref(done))
        b = getattr(self, "_bound", None)
        b.check(code, "mik_gmres_iterate") if b is not None else check(code, "mik_gmres_iterate", self.A.ctx.handle)
        if done.value:
            return None
        self._refresh()
        return self.residual_current, iteration + 1

    def iterate_many(self, iteration: int, max_steps: int) -> np.ndarray:
        """Up to ``max_steps`` ``iterate`` calls inside the library (``mik_gmres_iterate_many``); returns the residuals."""
        out = np.empty(max(int(max_steps), 1), np.float64)
        nd = C.c_int64()
        code = lib().mik_gmres_iterate_many(self.handle, int(iteration), int(max_steps), out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(nd))
        b = getattr(self, "_bound", None)
        b.check(code, "mik_gmres_iterate_many") if b is not None else check(code, "mik_gmres_iterate_many", self.A.ctx.handle)
        self._refresh()
        return out[: nd.value].copy()

    def __iter__(self):
        iteration = 0
        while True:
            nxt = self.iterate(iteration)
            if nxt is None:
                return
            residual, iteration = nxt
            yield residual

    def __del__(self):
        try:
            if getattr(self, "handle", None) and self.A.ctx.handle:
                lib().mik_gmres_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def gmres_iterable_(x: HipVector, A: HipCSR, b: HipVector, *, Pl=None, Pr=None, abstol=0.0, reltol=None, restart=None,
                    maxiter=None, initially_zero: bool = False, orth_meth: Optional[OrthogonalizationMethod] = None):
    """``gmres_iterable!(x, A, b; ...)`` -- src/gmres.jl:108-136."""
    for P, name in ((Pl, "Pl"), (Pr, "Pr")):
        if P is not None and not isinstance(P, (Identity, JacobiPrec)) and not callable(getattr(P, "ldiv_", None)):
            raise MikError(5, "gmres_iterable_", f"{name} needs ldiv_(y, x) (docs/src/preconditioning.md:5-14)")
    reltol = _default_reltol(b) if reltol is None else reltol
    restart = min(20, A.size(2)) if restart is None else restart           # :113
    maxiter = A.size(2) if maxiter is None else maxiter                    # :114
    orth_meth = ModifiedGramSchmidt() if orth_meth is None else orth_meth  # :116
    return GMRESIterable(x, A, b, abstol=abstol, reltol=reltol, restart=restart, maxiter=maxiter, initially_zero=initially_zero,
                         orth_meth=orth_meth, Pl=Pl, Pr=Pr)


def gmres_(x: HipVector, A: HipCSR, b: HipVector, *, Pl=None, Pr=None, abstol=0.0, reltol=None, restart=None, maxiter=None,
           log: bool = False, initially_zero: bool = False, verbose: bool = False, orth_meth=None):
    """``gmres!(x, A, b; ...)`` -> ``x`` or ``(x, history)`` -- src/gmres.jl:184-222."""
    reltol = _default_reltol(b) if reltol is None else reltol
    restart = min(20, A.size(2)) if restart is None else restart
    maxiter = A.size(2) if maxiter is None else maxiter
    history = ConvergenceHistory(partial=not log, restart=restart)          # :195
    history["abstol"] = abstol
    history["reltol"] = reltol
    if log:
        history.reserve_("resnorm", maxiter)                               # :198
    iterable = gmres_iterable_(x, A, b, Pl=Pl, Pr=Pr, abstol=abstol, reltol=reltol, maxiter=maxiter, restart=restart,
                               initially_zero=initially_zero, orth_meth=orth_meth)
    if verbose:
        print("=== gmres ===\n%4s\t%4s\t%7s" % ("rest", "iter", "resnorm"))
    for iteration, residual in enumerate(iterable, start=1):               # :207
        if log:
            history.nextiter_()                                            # :209
            history.mvps = iterable.mv_products                            # :210
            history.push_("resnorm", residual)                             # :211
        if verbose:
            print("%3d\t%3d\t%1.2e" % (1 + (iteration - 1) // restart, 1 + (iteration - 1) % restart, residual))
    if verbose:
        print()
    history.setconv(iterable.converged())                                  # :218
    if log:
        history.shrink_()                                                  # :219
    return (x, history) if log else x


def gmres(A: HipCSR, b: HipVector, **kwargs):
    """``gmres(A, b; ...)`` -- src/gmres.jl:143."""
    return gmres_(zerox(A, b), A, b, initially_zero=True, **kwargs)


# ==============================================================================================
# bicgstabl.jl  (composed from the L1 / L2 entry points)
# ==============================================================================================
class BiCGStabIterable:
    """``BiCGStabIterable`` -- src/bicgstabl.jl:5-23; construction follows ``bicgstabl_iterator!`` (:25-73).
    ``r_shadow`` replaces the reference's ``rand(T, n)`` (:38) so that runs are reproducible; by default it
    is the hashed vector of ``fixtures.hashed_rhs`` shifted into (0, 1)."""

    def __init__(self, x: HipVector, A: HipCSR, b: HipVector, l: int = 2, *, Pl=None, max_mv_products, abstol, reltol,
                 initial_zero, r_shadow: Optional[HipVector] = None, fused: bool = True):
        from . import fixtures
        T = x.dtype.type
        self.fused = bool(fused)          # one-pass Gram matrix and one-sweep MR update (same bits) for l <= 4
        n = A.size(1)
        self.A, self.l, self.x = A, int(l), x
        self.Pl = Identity() if Pl is None else Pl
        if not isinstance(self.Pl, (Identity, JacobiPrec)):
            raise MikError(5, "bicgstabl_iterator_", "Pl must be Identity() or a diagonal JacobiPrec on the device path")
        self.mv_products = 0
        self.r_shadow = r_shadow if r_shadow is not None else HipVector.from_numpy((fixtures.hashed_rhs(n) + 0.5).astype(x.dtype), x.ctx)
        self.rs = HipMatrix(n, self.l + 1, x.dtype, x.ctx)                  # :39
        self.us = HipMatrix(n, self.l + 1, x.dtype, x.ctx)                  # :40 zeros
        residual = self.rs.col(0)
        if initial_zero:
            residual.copyto_(b)                                              # :46
        else:
            mul_(residual, A, x)                                             # :48
            residual.xpby_(b, T(-1))                                         # residual .= b .- residual  :49
            self.mv_products += 1
        self._ldiv(residual)                                                 # :55
        self.gamma = np.zeros(self.l, x.dtype)                               # :58
        self.omega = self.sigma = T(1)                                       # :59
        self.residual = norm(residual)                                       # :61
        self.M = np.zeros((self.l + 1, self.l + 1), x.dtype, order="F")      # :66
        self.tol = max(T(reltol) * self.residual, T(abstol))                 # :69
        self.max_mv_products = int(max_mv_products)
        # fused, l <= 4, a HipCSR operator: the whole outer iteration is one C call with its scalars on the device
        # (mik_bicgstab_step); omega, sigma, gamma and M then live there and the host copies above stay at their initial values
        self._step = None
        if self.fused and self.l <= 4 and isinstance(A, HipCSR):
            h = _vp()
            d = self.Pl.diagonal.ptr if isinstance(self.Pl, JacobiPrec) else None
            check(lib().mik_bicgstab_create(x.ctx.handle, A.handle, self.l, _vp(x.ptr), _vp(self.rs.col(0).ptr), self.rs.ld, _vp(self.us.col(0).ptr),
                                            self.us.ld, _vp(self.r_shadow.ptr), _vp(d), C.byref(h)), "mik_bicgstab_create", x.ctx.handle)
            self._step = h

    def _ldiv(self, v: HipVector):
        if isinstance(self.Pl, JacobiPrec):
            self.Pl.ldiv_(v)

    def dot_shape(self):
        """(W, L) of the reduction tree of ``sigma = dot(r_shadow, A u)`` (src/bicgstabl.jl:100) and of ``rho`` from the second column on
        (:89): the SpMV-dot shape where the whole-iteration call forms them in the SpMV launch (``mik_bicgstab_dot_shape``), the vector
        shape otherwise -- what the oracle's ``dot_shape`` takes."""
        if self._step is None:
            return self.x.ctx.reduce_shape(self.x.dtype)
        w, l = C.c_int(), C.c_int()
        check(lib().mik_bicgstab_dot_shape(self._step, C.byref(w), C.byref(l)), "mik_bicgstab_dot_shape", self.x.ctx.handle)
        return w.value, l.value

    def converged(self) -> bool:                                             # :75
        return self.residual <= self.tol

    def start(self) -> int:
        return 0

    def done(self, iteration: int) -> bool:                                  # :77
        return self.mv_products >= self.max_mv_products or self.converged()

    def iterate(self, iteration: Optional[int] = None):
        """``iterate(it, iteration)`` -- src/bicgstabl.jl:79-134."""
        iteration = 0 if iteration is None else iteration
        if self.done(iteration):
            return None
        l, rs, us = self.l, self.rs, self.us
        if self._step is not None:
            out = np.zeros(1, self.x.dtype)
            rc = lib().mik_bicgstab_step(self._step, out.ctypes.data_as(_vp))
            if rc == 8:                                                      # MIK_ERR_SINGULAR (never an invalid-argument code)
                raise np.linalg.LinAlgError("SingularException")
            check(rc, "mik_bicgstab_step", self.x.ctx.handle)
            self.mv_products += 2 * l                                        # :115
            self.residual = out[0]
            return self.residual, iteration + 1
        self.sigma = -self.omega * self.sigma                                # :85
        for j in range(l):                                                   # BiCG part  :88
            rho = dot(self.r_shadow, rs.col(j))                              # :89
            beta = rho / self.sigma                                          # :90
            for q in range(j + 1):
                us.col(q).xpby_(rs.col(q), -beta)                            # us = rs - beta*us  :93
            mul_(us.col(j + 1), self.A, us.col(j))                           # :97
            self._ldiv(us.col(j + 1))                                        # :98
            self.sigma = dot(self.r_shadow, us.col(j + 1))                   # :100
            alpha = rho / self.sigma                                         # :101
            for q in range(j + 1):
                rs.col(q).axpy_(-alpha, us.col(q + 1))                       # rs -= alpha*us  :103
            mul_(rs.col(j + 1), self.A, rs.col(j))                           # :107
            self._ldiv(rs.col(j + 1))                                        # :108
            self.x.axpy_(alpha, us.col(0))                                   # :111
        self.mv_products += 2 * l                                            # :115
        fused = self.fused and l + 1 <= 5
        if fused:
            self.M[:, :] = gram_(rs, l + 1)                                  # M = rs' * rs  :120, one pass
        else:
            for c in range(l + 1):                                           # M = rs' * rs  :120
                self.M[:, c] = gemv_t_(rs, l + 1, rs.col(c))
        Msub = np.asfortranarray(self.M[1:, 1:].copy())
        self.gamma = lu_solve_(Msub, self.M[1:, 0].copy())                   # :123-125
        self.omega = self.gamma[l - 1]                                       # :131
        if fused:                                                            # :127-129, :132 in one sweep
            out = np.zeros(1, self.x.dtype)
            g = np.ascontiguousarray(self.gamma, self.x.dtype)
            check(lib().mik_bicgstab_mr_update(self.x.ctx.handle, self.x.code, self.x.n, l, _vp(us.col(0).ptr), us.ld, _vp(rs.col(0).ptr),
                                               rs.ld, _vp(self.x.ptr), g.ctypes.data_as(_vp), out.ctypes.data_as(_vp)),
                  "mik_bicgstab_mr_update", self.x.ctx.handle)
            self.residual = out[0]
            return self.residual, iteration + 1
        gemv_n_(us.col(0), us, l, self.gamma, -1.0, col0=1)                  # :127
        gemv_n_(self.x, rs, l, self.gamma, 1.0, col0=0)                      # :128
        gemv_n_(rs.col(0), rs, l, self.gamma, -1.0, col0=1)                  # :129
        self.residual = norm(rs.col(0))                                      # :132
        return self.residual, iteration + 1

    def __iter__(self):
        iteration = 0
        while (nxt := self.iterate(iteration)) is not None:
            residual, iteration = nxt
            yield residual

    def __del__(self):
        try:
            if getattr(self, "_step", None) is not None and self.x.ctx.handle:
                lib().mik_bicgstab_destroy(self._step)
                self._step = None
        except Exception:
            pass


def bicgstabl_iterator_(x: HipVector, A: HipCSR, b: HipVector, l: int = 2, *, Pl=None, max_mv_products=None, abstol=0.0,
                        reltol=None, initial_zero: bool = False, r_shadow: Optional[HipVector] = None, fused: bool = True):
    """``bicgstabl_iterator!(x, A, b, l; ...)`` -- src/bicgstabl.jl:25-73."""
    reltol = _default_reltol(b) if reltol is None else reltol
    max_mv_products = A.size(2) if max_mv_products is None else max_mv_products
    return BiCGStabIterable(x, A, b, l, Pl=Pl, max_mv_products=max_mv_products, abstol=abstol, reltol=reltol,
                            initial_zero=initial_zero, r_shadow=r_shadow, fused=fused)


def bicgstabl_(x: HipVector, A: HipCSR, b: HipVector, l: int = 2, *, abstol=0.0, reltol=None, max_mv_products=None,
               log: bool = False, verbose: bool = False, Pl=None, **kwargs):
    """``bicgstabl!(x, A, b, l; ...)`` -> ``x`` or ``(x, history)`` -- src/bicgstabl.jl:181-219."""
    reltol = _default_reltol(b) if reltol is None else reltol
    max_mv_products = A.size(2) if max_mv_products is None else max_mv_products
    history = ConvergenceHistory(partial=not log)
    history["abstol"] = abstol
    history["reltol"] = reltol
    if log:
        history.reserve_("resnorm", max_mv_products)                        # :194
    iterable = bicgstabl_iterator_(x, A, b, l, Pl=Pl, abstol=abstol, reltol=reltol, max_mv_products=max_mv_products, **kwargs)
    if log:
        history.mvps = iterable.mv_products                                  # :202
    for iteration, _item in enumerate(iterable, start=1):                    # :205
        if log:
            history.nextiter_()
            history.mvps = iterable.mv_products                              # :208
            history.push_("resnorm", iterable.residual)
        if verbose:
            print("%3d\t%1.2e" % (iteration, iterable.residual))
    if verbose:
        print()
    if log:
        history.setconv(iterable.converged())
        history.shrink_()
    return (iterable.x, history) if log else iterable.x


def bicgstabl(A: HipCSR, b: HipVector, l: int = 2, **kwargs):
    """``bicgstabl(A, b, l; ...)`` -- src/bicgstabl.jl:142."""
    return bicgstabl_(zerox(A, b), A, b, l, initial_zero=True, **kwargs)


# ==============================================================================================
# chebyshev.jl, minres.jl  (composed from the L1 entry points)
# ==============================================================================================
def givens_algorithm(f, g, dtype=np.float64):
    """``LinearAlgebra.givensAlgorithm(f, g)`` -> (c, s, r) on the host -- src/minres.jl:129."""
    fa, ga, out = np.asarray([f], dtype), np.asarray([g], dtype), np.zeros(3, dtype)
    check(lib().mik_givens(dtype_code(dtype), fa.ctypes.data_as(_vp), ga.ctypes.data_as(_vp), out.ctypes.data_as(_vp)), "mik_givens", None)
    return out[0], out[1], out[2]


class ChebyshevIterable:
    """``ChebyshevIterable`` -- src/chebyshev.jl:5-22, construction per ``chebyshev_iterable!`` (:59-91).  Follows the
    v0.9.4 source as written: ``start = 0`` with the ``iteration == 1`` branch (:26, :37) and
    ``u .= c .+ beta .* c`` (:45)."""

    def __init__(self, x, A, b, lmin, lmax, *, abstol, reltol, maxiter, Pl=None, initially_zero=False, fused=True):
        T = x.dtype.type
        self.fused = bool(fused)          # one sweep per statement group (same bits) instead of one L1 call per statement
        self.Pl = Identity() if Pl is None else Pl
        if not isinstance(self.Pl, (Identity, JacobiPrec)):
            raise MikError(5, "chebyshev_iterable_", "Pl must be Identity() or a diagonal JacobiPrec on the device path")
        self.A, self.x = A, x
        self.l_avg = (T(lmax) + T(lmin)) / T(2)                              # :66
        self.l_diff = (T(lmax) - T(lmin)) / T(2)                             # :67
        self.r = x.similar().copyto_(b)                                      # :70-71
        self.u = x.zero()
        self.c = x.similar()
        if initially_zero:
            self.mv_products = 0
        else:
            self.mv_products = 1
            mul_(self.c, A, x)                                               # :80
            self.r.sub_(self.c)                                              # :81
        self.resnorm = norm(self.r)                                          # :83
        self.tol = max(T(reltol) * self.resnorm, T(abstol))                  # :84
        self.alpha = T(0)
        self.maxiter = int(maxiter)

    def converged(self):
        return self.resnorm <= self.tol

    def start(self):
        return 0

    def done(self, iteration):
        return iteration >= self.maxiter or self.converged()

    def iterate(self, iteration=None):
        iteration = 0 if iteration is None else iteration
        if self.done(iteration):
            return None
        T = self.x.dtype.type
        if self.fused:
            first = iteration == 1                                           # :37
            if first:
                self.alpha = T(2) / self.l_avg
                beta = T(0)
            else:
                h = (self.l_diff * self.alpha) / T(2)
                beta = h * h                                                 # :41
                self.alpha = T(1) / (self.l_avg - beta)                      # :42
            _, pb = _scalar(self.x.dtype, beta)
            d = self.Pl.diagonal.ptr if isinstance(self.Pl, JacobiPrec) else None
            check(lib().mik_cheb_direction(self.x.ctx.handle, self.x.code, self.x.n, _vp(self.r.ptr), _vp(d), pb, int(first), _vp(self.u.ptr)),
                  "mik_cheb_direction", self.x.ctx.handle)                   # :35-45
            mul_(self.c, self.A, self.u)                                     # :48
            self.mv_products += 1
            self.resnorm = axpy2_nrm2_(self.alpha, self.u, self.x, self.c, self.r, hints=7)   # :51-54; x, c, u are not re-read
            return self.resnorm, iteration + 1
        self.Pl.ldiv_(self.c, self.r)                                        # :35
        if iteration == 1:                                                   # :37
            self.alpha = T(2) / self.l_avg
            self.u.copyto_(self.c)
        else:
            h = (self.l_diff * self.alpha) / T(2)
            beta = h * h                                                     # :41
            self.alpha = T(1) / (self.l_avg - beta)                          # :42
            self.u.copyto_(self.c).axpy_(beta, self.c)                       # u .= c .+ beta .* c  :45
        mul_(self.c, self.A, self.u)                                         # :48
        self.mv_products += 1
        self.x.axpy_(self.alpha, self.u)                                     # :51
        self.r.axpy_(-self.alpha, self.c)                                    # :52
        self.resnorm = norm(self.r)                                          # :54
        return self.resnorm, iteration + 1

    def __iter__(self):
        iteration = 0
        while (nxt := self.iterate(iteration)) is not None:
            resnorm, iteration = nxt
            yield resnorm


def chebyshev_iterable_(x, A, b, lmin, lmax, *, abstol=0.0, reltol=None, maxiter=None, Pl=None, initially_zero=False, fused=True):
    """``chebyshev_iterable!`` -- src/chebyshev.jl:59-91."""
    return ChebyshevIterable(x, A, b, lmin, lmax, abstol=abstol, reltol=_default_reltol(b) if reltol is None else reltol,
                             maxiter=A.size(2) if maxiter is None else maxiter, Pl=Pl, initially_zero=initially_zero, fused=fused)


def _drive(iterable, history, log, verbose, per_iter_mvps=None):
    for iteration, resnorm in enumerate(iterable, start=1):
        history.nextiter_(**({} if per_iter_mvps is None else {"mvps": per_iter_mvps}))
        if per_iter_mvps is None:
            history.mvps = iterable.mv_products
        history.push_("resnorm", resnorm)
        if verbose:
            print("%3d\t%1.2e" % (iteration, resnorm))
    history.setconv(iterable.converged())
    if log:
        history.shrink_()


def chebyshev_(x, A, b, lmin, lmax, *, abstol=0.0, reltol=None, Pl=None, maxiter=None, log=False, verbose=False,
               initially_zero=False):
    """``chebyshev!(x, A, b, lmin, lmax; ...)`` -- src/chebyshev.jl:142-169."""
    reltol = _default_reltol(b) if reltol is None else reltol
    maxiter = A.size(2) if maxiter is None else maxiter
    history = ConvergenceHistory(partial=not log)
    history["abstol"], history["reltol"] = abstol, reltol
    history.reserve_("resnorm", maxiter)                                     # :154
    it = chebyshev_iterable_(x, A, b, lmin, lmax, abstol=abstol, reltol=reltol, maxiter=maxiter, Pl=Pl, initially_zero=initially_zero)
    history.mvps = it.mv_products                                            # :160
    _drive(it, history, log, verbose)
    return (x, history) if log else x


def chebyshev(A, b, lmin, lmax, **kwargs):
    """``chebyshev(A, b, lmin, lmax; ...)`` -- src/chebyshev.jl:100-101."""
    return chebyshev_(zerox(A, b), A, b, lmin, lmax, initially_zero=True, **kwargs)


class MINRESIterable:
    """``MINRESIterable`` -- src/minres.jl:6-36, construction per ``minres_iterable!`` (:38-87); real element types."""

    def __init__(self, x, A, b, *, initially_zero=False, skew_hermitian=False, abstol, reltol, maxiter, fused=True):
        T = x.dtype.type
        self.fused = bool(fused)          # one sweep per statement group (same bits) instead of one L1 call per statement
        self.A, self.x, self.skew = A, x, bool(skew_hermitian)
        self.v_prev, self.v_curr, self.v_next = x.similar(), x.similar().copyto_(b), x.similar()       # :47-50
        self.w_prev, self.w_curr, self.w_next = x.zero(), x.zero(), x.zero()                            # :51-53
        self.mv_products = 0
        if not initially_zero:
            mul_(self.v_next, A, x)                                          # :60
            self.v_curr.axpy_(T(-1), self.v_next)                            # :61
            self.mv_products = 1
        self.resnorm = norm(self.v_curr)                                     # :65
        self.tol = max(T(reltol) * self.resnorm, T(abstol))                  # :66
        self.H = np.zeros(4, x.dtype)                                        # :70
        self.rhs = np.array([self.resnorm, 0], x.dtype)                      # :71
        self.v_curr.scal_(T(1) / self.resnorm)                               # :74
        self.c_prev, self.s_prev, self.c_curr, self.s_curr = T(1), T(0), T(1), T(0)
        self.maxiter = int(maxiter)
        # fused, a HipCSR operator: the whole iteration is one C call with H, rhs and the rotations on the device (mik_minres_step);
        # the host copies above then stay at their initial values
        self._step = None
        if self.fused and isinstance(A, HipCSR):
            h = _vp()
            check(lib().mik_minres_create(x.ctx.handle, A.handle, _vp(x.ptr), _vp(self.v_prev.ptr), _vp(self.v_curr.ptr), _vp(self.v_next.ptr),
                                          _vp(self.w_prev.ptr), _vp(self.w_curr.ptr), _vp(self.w_next.ptr), float(self.resnorm), int(self.skew), C.byref(h)),
                  "mik_minres_create", x.ctx.handle)
            self._step = h

    def proj_shape(self):
        """(W, L) of the reduction tree of ``proj = dot(v_curr, v_next)`` (src/minres.jl:107): the SpMV-dot shape where the whole-iteration
        call forms it in the SpMV launch (``mik_minres_proj_shape``), the vector shape otherwise -- what the oracle's ``proj_shape`` takes."""
        if self._step is None:
            return self.x.ctx.reduce_shape(self.x.dtype)
        w, l = C.c_int(), C.c_int()
        check(lib().mik_minres_proj_shape(self._step, C.byref(w), C.byref(l)), "mik_minres_proj_shape", self.x.ctx.handle)
        return w.value, l.value

    def converged(self):
        return self.resnorm <= self.tol                                      # :89

    def start(self):
        return 1                                                             # :91

    def done(self, iteration):
        return iteration > self.maxiter or self.converged()                  # :93

    def iterate(self, iteration=None):
        """``iterate(m, iteration)`` -- src/minres.jl:95-159."""
        iteration = 1 if iteration is None else iteration
        if self.done(iteration):
            return None
        T, H, rhs = self.x.dtype.type, self.H, self.rhs
        if self._step is not None:
            out = np.zeros(1, self.x.dtype)
            check(lib().mik_minres_step(self._step, int(iteration), out.ctypes.data_as(_vp)), "mik_minres_step", self.x.ctx.handle)
            self.v_prev, self.v_curr, self.v_next = self.v_curr, self.v_next, self.v_prev       # :145 (the handle rotates its pointers alike)
            self.w_prev, self.w_curr, self.w_next = self.w_curr, self.w_next, self.w_prev       # :146
            self.resnorm = out[0]                                            # :154
            self.mv_products += 1
            return self.resnorm, iteration + 1
        mul_(self.v_next, self.A, self.v_curr)                               # :102
        if self.fused:
            proj = axpy_dot_(-H[1], self.v_prev if iteration > 1 else None, self.v_next, self.v_curr, hints=1)   # :104, :107; v_prev is dead
            H[2] = proj
            H[3] = axpy_dot_(-proj, self.v_curr, self.v_next, None)          # :109, :112
        else:
            if iteration > 1:
                self.v_next.axpy_(-H[1], self.v_prev)                        # :104
            proj = dot(self.v_curr, self.v_next)                             # :107
            H[2] = proj
            self.v_next.axpy_(-proj, self.v_curr)                            # :109
            H[3] = norm(self.v_next)                                         # :112
            self.v_next.scal_(T(1) / H[3])                                   # :113
        inv_h3 = T(1) / H[3]
        if iteration > 2:                                                    # :116-119
            H[0] = self.s_prev * H[1]
            H[1] = self.c_prev * H[1]
        if iteration > 1:                                                    # :122-126
            tmp = -self.s_curr * H[1] + self.c_curr * H[2]
            H[1] = self.c_curr * H[1] + self.s_curr * H[2]
            H[2] = tmp
        c, s, H[2] = givens_algorithm(H[2], H[3], self.x.dtype)              # :129
        rhs[1] = -s * rhs[0]                                                 # :132
        rhs[0] = c * rhs[0]                                                  # :133
        if self.fused:                                                       # :113, :136-142 in one sweep
            dt = self.x.dtype
            sc = [_scalar(dt, v) for v in (inv_h3, -H[1], -H[0], T(1) / H[2], rhs[0])]
            check(lib().mik_minres_update(self.x.ctx.handle, self.x.code, self.x.n, sc[0][1], _vp(self.v_next.ptr), _vp(self.v_curr.ptr),
                                          sc[1][1], _vp(self.w_curr.ptr if iteration > 1 else None),
                                          sc[2][1], _vp(self.w_prev.ptr if iteration > 2 else None),
                                          sc[3][1], _vp(self.w_next.ptr), sc[4][1], _vp(self.x.ptr), 3), "mik_minres_update", self.x.ctx.handle)
        else:
            self.w_next.copyto_(self.v_curr)                                 # :136
            if iteration > 1:
                self.w_next.axpy_(-H[1], self.w_curr)                        # :137
            if iteration > 2:
                self.w_next.axpy_(-H[0], self.w_prev)                        # :138
            self.w_next.scal_(T(1) / H[2])                                   # :139
            self.x.axpy_(rhs[0], self.w_next)                                # :142
        self.v_prev, self.v_curr, self.v_next = self.v_curr, self.v_next, self.v_prev       # :145
        self.w_prev, self.w_curr, self.w_next = self.w_curr, self.w_next, self.w_prev       # :146
        self.c_prev, self.s_prev, self.c_curr, self.s_curr = self.c_curr, self.s_curr, c, s  # :147
        rhs[0] = rhs[1]                                                      # :148
        H[1] = -H[3] if self.skew else H[3]                                  # :151
        self.resnorm = abs(rhs[1])                                           # :154
        self.mv_products += 1
        return self.resnorm, iteration + 1

    def __iter__(self):
        iteration = 1
        while (nxt := self.iterate(iteration)) is not None:
            resnorm, iteration = nxt
            yield resnorm

    def __del__(self):
        try:
            if getattr(self, "_step", None) is not None and self.x.ctx.handle:
                lib().mik_minres_destroy(self._step)
                self._step = None
        except Exception:
            pass


def minres_iterable_(x, A, b, *, initially_zero=False, skew_hermitian=False, abstol=0.0, reltol=None, maxiter=None, fused=True):
    """``minres_iterable!`` -- src/minres.jl:38-87."""
    return MINRESIterable(x, A, b, initially_zero=initially_zero, skew_hermitian=skew_hermitian, abstol=abstol,
                          reltol=_default_reltol(b) if reltol is None else reltol, maxiter=A.size(2) if maxiter is None else maxiter,
                          fused=fused)


def minres_(x, A, b, *, skew_hermitian=False, verbose=False, log=False, abstol=0.0, reltol=None, maxiter=None, initially_zero=False):
    """``minres!(x, A, b; ...)`` -- src/minres.jl:197-230."""
    reltol = _default_reltol(b) if reltol is None else reltol
    maxiter = A.size(2) if maxiter is None else maxiter
    history = ConvergenceHistory(partial=not log)
    history["abstol"], history["reltol"] = abstol, reltol
    if log:
        history.reserve_("resnorm", maxiter)
    it = minres_iterable_(x, A, b, skew_hermitian=skew_hermitian, abstol=abstol, reltol=reltol, maxiter=maxiter, initially_zero=initially_zero)
    mv0 = it.mv_products
    if log:
        history.mvps = mv0                                                   # :211
    n_it = 0
    for iteration, resnorm in enumerate(it, start=1):                        # :214
        n_it += 1
        if log:
            history.nextiter_(mvps=1)                                        # :216
            history.push_("resnorm", resnorm)
        if verbose:
            print("%3d\t%1.2e" % (iteration, resnorm))
    if log:
        history.setconv(it.converged())
        history.shrink_()
    return (it.x, history) if log else it.x


def minres(A, b, **kwargs):
    """``minres(A, b; ...)`` -- src/minres.jl:236."""
    return minres_(zerox(A, b), A, b, initially_zero=True, **kwargs)
