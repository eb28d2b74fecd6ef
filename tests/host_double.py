"""A numpy stand-in for the device vector / operator types of api.py (test infrastructure): the statement-by-statement mirrors of the solvers
(IDRSIterable with fused=False, lsqr_, lsmr_) run on it on a CPU box, every vector statement evaluated with the oracle's SEQ primitives, so
their control flow and scalar arithmetic can be compared with the C oracle bit for bit without a GPU."""
import ctypes as C

import numpy as np


class FakeCtx:
    handle = None


class FakeVector:
    def __init__(self, a):
        self.a = np.ascontiguousarray(a)
        self.n, self.dtype, self.ctx = self.a.size, self.a.dtype, FakeCtx()

    @property
    def ptr(self):
        return 0

    def similar(self):
        return FakeVector(np.empty_like(self.a))

    def zero(self):
        return FakeVector(np.zeros_like(self.a))

    def copyto_(self, src):
        self.a[:] = src.a
        return self

    def fill_(self, v):
        self.a[:] = v
        return self

    def axpy_(self, alpha, x):                    # self .+= alpha .* x
        t = self.dtype.type(alpha) * x.a
        self.a[:] = self.a + t
        return self

    def xpby_(self, x, beta):                     # self .= x .+ beta .* self
        t = self.dtype.type(beta) * self.a
        self.a[:] = x.a + t
        return self

    def sub_(self, x):
        self.a[:] = self.a - x.a
        return self

    def scal_(self, alpha):
        self.a[:] = self.a * self.dtype.type(alpha)
        return self

    def to_numpy(self):
        return self.a.copy()


class FakeMatrix:
    def __init__(self, n, cols, dtype=np.float64, ctx=None):
        self.n, self.cols, self.ld = int(n), int(cols), int(n)
        self.m = np.zeros((cols, n), dtype)

    @staticmethod
    def from_numpy(a, ctx=None):
        a = np.asarray(a)
        out = FakeMatrix(a.shape[0], a.shape[1], a.dtype)
        out.m[:] = a.T
        return out

    def col(self, j):
        v = FakeVector.__new__(FakeVector)
        v.a = self.m[j]
        v.n, v.dtype, v.ctx = v.a.size, v.a.dtype, FakeCtx()
        return v


class FakeOperator:
    """A (scipy sparse, m x n) with its adjoint; products by the oracle's column scatter"""

    def __init__(self, orc, S, adj=None):
        S = S.tocsc()
        S.sort_indices()
        self.orc, self.S = orc, S
        self.n_rows, self.n_cols = S.shape
        self.dtype = S.dtype
        self.ctx = FakeCtx()
        self.adj = adj if adj is not None else FakeOperator(orc, S.T.tocsc(), adj=self)

    def size(self, d=None):
        return (self.n_rows, self.n_cols) if d is None else (self.n_rows, self.n_cols)[d - 1]

    def mul(self, y, x):
        S = self.S
        suf = "f64" if self.dtype == np.float64 else "f32"
        ct = C.c_double if self.dtype == np.float64 else C.c_float
        cp, rv = S.indptr.astype(np.int64), S.indices.astype(np.int64)
        xa = np.ascontiguousarray(x.a)
        out = np.empty(self.n_rows, self.dtype)
        getattr(self.orc.lib(), f"orc_csc_spmv_{suf}")(self.n_rows, self.n_cols, cp.ctypes.data_as(C.POINTER(C.c_int64)), rv.ctypes.data_as(C.POINTER(C.c_int64)),
                                                       S.data.ctypes.data_as(C.POINTER(ct)), 0, xa.ctypes.data_as(C.POINTER(ct)), out.ctypes.data_as(C.POINTER(ct)))
        y.a[:] = out
        return y


def patch(monkeypatch, api, orc):
    """route api.py's L1 functions to the oracle's SEQ primitives"""
    monkeypatch.setattr(api, "norm", lambda x: x.dtype.type(orc.nrm2(np.ascontiguousarray(x.a))))
    monkeypatch.setattr(api, "dot", lambda x, y: x.dtype.type(orc.dot(np.ascontiguousarray(x.a), np.ascontiguousarray(y.a))))
    monkeypatch.setattr(api, "mul_", lambda y, A, x: A.mul(y, x))
    monkeypatch.setattr(api, "HipMatrix", FakeMatrix)
