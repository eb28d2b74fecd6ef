"""Solvers BEYOND the scope contract: IDR(s), LSQR, LSMR, QMR and the power method.

SURVEY.md section 2 marks ``src/idrs.jl``, ``src/lsqr.jl``, ``src/lsmr.jl``, ``src/qmr.jl`` and ``src/simple.jl`` OUT OF SCOPE; these
host mirrors were written in round 5 and are kept here, apart from ``api.py`` (which holds only the section-8 rows), unjudged and
unadvertised.  They run on the same L1 / L2 entry points of libmik.so as the section-8 solvers (IDR(s) also on ``mik_idrs_*``), are
tested against their oracle restatements (``tests/test_idrs.py``, ``test_lsqr_lsmr.py``, ``test_qmr.py``, ``test_powm.py``) and are benchmarked
only by ``bench.py --extras``.  Bit parity is against the repository's own oracle; the reference's ``LowerTriangular \\`` in IDR(s)
dispatches to BLAS ``trsv``, whose operation order may differ by ulps from the forward substitution used here and in the oracle.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from ._lib import MikError, check, lib
from .api import (ConvergenceHistory, HipCSR, HipMatrix, HipVector, Identity, JacobiPrec, _default_reltol, _scalar, _vp, dot,     # noqa: F401
                  givens_algorithm, mul_, norm, zerox)


def with_adjoint(n_rows, n_cols, colptr, rowval, nzval, *, index_base=1, ctx=None) -> HipCSR:
    """The operator of a ``SparseMatrixCSC`` together with its adjoint (``adjoint(A)`` below; real element types): the SAME three arrays read
    as a CSR matrix are A' (n_cols x n_rows) -- row j of A' is column j of A, entries in storage order, which is the order
    ``mul!(y, adjoint(A), x)`` of SparseArrays sums them in.  Two device operators (a second copy of the arrays in HBM); no transpose is
    ever formed for the adjoint."""
    A = HipCSR(n_rows, n_cols, colptr, rowval, nzval, index_base=index_base, is_csc=True, ctx=ctx)
    A.adj = HipCSR(n_cols, n_rows, colptr, rowval, nzval, index_base=index_base, is_csc=False, ctx=A.ctx)
    A.adj.adj = A
    return A


def with_adjoint_from_scipy(m, ctx=None) -> HipCSR:
    m = m.tocsc()
    m.sort_indices()
    return with_adjoint(m.shape[0], m.shape[1], m.indptr, m.indices, m.data, index_base=0, ctx=ctx)


class IDRSIterable:
    """``IDRSIterable`` -- src/idrs.jl:84-113, construction per ``idrs_iterable!`` (:116-147); real element types.  ``P`` replaces the
    reference's ``rand!`` shadow vectors (:136) when reproducibility is wanted: a ``HipMatrix`` or an n x s array; default: uniform [0, 1)
    numbers like ``rand!``.  The iteration state is the pair ``(iter, step)`` of the reference (:164)."""

    def __init__(self, log, X, A, C_, s, Pl, abstol, reltol, maxiter, *, smoothing=False, verbose=False, P=None, fused=True):
        T = X.dtype.type
        self.log, self.X, self.A, self.s, self.smoothing, self.verbose = log, X, A, int(s), bool(smoothing), bool(verbose)
        self.Pl = Identity() if Pl is None else Pl
        self.abstol, self.reltol, self.maxiter = abstol, reltol, maxiter
        n = X.n
        self.R = X.similar()
        mul_(self.R, A, X)                                                   # R = C - A*X  :119
        self.R.xpby_(C_, T(-1))
        self.normR = norm(self.R)                                            # :120
        self.tol = max(T(reltol) * self.normR, T(abstol))                    # :121
        if self.smoothing:                                                   # :123-126
            self.X_s, self.R_s, self.T_s = X.similar().copyto_(X), X.similar().copyto_(self.R), X.zero()
        else:
            self.X_s = self.R_s = self.T_s = None
        if P is None:
            P = np.random.default_rng().random((n, self.s)).astype(X.dtype)  # :136
        if isinstance(P, HipMatrix):
            if P.n != n or P.cols < self.s or P.dtype != X.dtype or P.ctx is not X.ctx:
                raise MikError(3, "IDRSIterable", "P must be an n x s block (n = %d, s = %d) of X's element type on X's context" % (n, self.s))
        else:
            P = np.asarray(P, X.dtype)
            if P.size != n * self.s:
                raise MikError(3, "IDRSIterable", "P must hold n x s = %d x %d entries" % (n, self.s))
            P = HipMatrix.from_numpy(P.reshape(n, self.s), X.ctx)
        self.P = P
        self.U, self.G = HipMatrix(n, self.s, X.dtype, X.ctx), HipMatrix(n, self.s, X.dtype, X.ctx)   # :137-138
        self.Q, self.V = X.zero(), X.zero()                                  # :139-140
        self.M = np.eye(self.s, dtype=X.dtype, order="F")                    # :142
        self.f = np.zeros(self.s, X.dtype)                                   # :143
        self.c = np.zeros(self.s, X.dtype)
        self.omega = T(1)                                                    # :146
        # fused, a HipCSR operator, Identity / diagonal Pl, s <= 32: one C call per step (mik_idrs_step); M, f and omega then live in the handle
        self._step = None
        if fused and isinstance(A, HipCSR) and isinstance(self.Pl, (Identity, JacobiPrec)) and self.s <= 32:
            h = _vp()
            d = self.Pl.diagonal.ptr if isinstance(self.Pl, JacobiPrec) else None
            check(lib().mik_idrs_create(X.ctx.handle, A.handle, self.s, _vp(X.ptr), _vp(self.R.ptr), _vp(self.P.col(0).ptr), self.P.ld,
                                        _vp(self.U.col(0).ptr), self.U.ld, _vp(self.G.col(0).ptr), self.G.ld, _vp(d),
                                        _vp(self.X_s.ptr if self.smoothing else None), _vp(self.R_s.ptr if self.smoothing else None),
                                        float(self.normR), C.byref(h)), "mik_idrs_create", X.ctx.handle)
            self._step = h

    def _ldiv(self, v):
        if not isinstance(self.Pl, Identity):
            self.Pl.ldiv_(v)

    def _smooth(self):                                                       # :226-235, :257-266
        self.T_s.copyto_(self.R_s).sub_(self.R)
        gamma = dot(self.R_s, self.T_s) / dot(self.T_s, self.T_s)
        self.R_s.axpy_(-gamma, self.T_s)
        self.T_s.copyto_(self.X_s).sub_(self.X)                              # X_s .- X (T_s is free again)
        self.X_s.axpy_(-gamma, self.T_s)
        self.normR = norm(self.R_s)

    def state(self):
        """(omega, M, f) as the iteration holds them (the handle's copies on the fused path)."""
        if self._step is None:
            return self.omega, self.M.copy(), self.f.copy()
        om, M, f = np.zeros(1, self.X.dtype), np.zeros((self.s, self.s), self.X.dtype, order="F"), np.zeros(self.s, self.X.dtype)
        check(lib().mik_idrs_state(self._step, om.ctypes.data_as(_vp), M.ctypes.data_as(_vp), f.ctypes.data_as(_vp)), "mik_idrs_state", self.X.ctx.handle)
        return om[0], M, f

    def iterate(self, state=None):
        """``iterate(it, (iter, step))`` -- src/idrs.jl:164-272."""
        it, step = (1, 1) if state is None else state
        T = self.X.dtype.type
        s, P, U, G, M, f = self.s, self.P, self.U, self.G, self.M, self.f
        if self.normR < self.tol or it > self.maxiter:                       # :168
            if self.log is not None:
                self.log.setconv(bool(0 <= self.normR < self.tol))
            if self.smoothing:
                self.X.copyto_(self.X_s)                                     # :171-173
            return None
        if self._step is not None:
            out = np.zeros(1, self.X.dtype)
            check(lib().mik_idrs_step(self._step, int(step), out.ctypes.data_as(_vp)), "mik_idrs_step", self.X.ctx.handle)
            self.normR = out[0]
            nextstep = step + 1 if step <= s else 1
        elif step <= s:
            if step == 1:
                for i in range(s):
                    f[i] = dot(P.col(i), self.R)                             # :179-181
            k = step - 1
            c = f[k:].copy()                                                 # c = LowerTriangular(M[k:s,k:s]) \ f[k:s]  :187
            for j in range(k, s):
                c[j - k] = c[j - k] / M[j, j]
                for i in range(j + 1, s):
                    c[i - k] = c[i - k] - M[i, j] * c[j - k]
            self.V.copyto_(G.col(k)).scal_(c[0])                             # :188
            self.Q.copyto_(U.col(k)).scal_(c[0])                             # :189
            for i in range(k + 1, s):                                        # :191-194
                self.V.axpy_(c[i - k], G.col(i))
                self.Q.axpy_(c[i - k], U.col(i))
            self.V.xpby_(self.R, T(-1))                                      # V .= R .- V  :197
            self._ldiv(self.V)                                               # :200
            U.col(k).copyto_(self.Q).axpy_(self.omega, self.V)               # :202
            mul_(G.col(k), self.A, U.col(k))                                 # :203
            for i in range(k):                                               # :207-211
                alpha = dot(P.col(i), G.col(k)) / M[i, i]
                G.col(k).axpy_(-alpha, G.col(i))
                U.col(k).axpy_(-alpha, U.col(i))
            for i in range(k, s):
                M[i, k] = dot(P.col(i), G.col(k))                            # :215-217
            beta = f[k] / M[k, k]                                            # :221
            self.R.axpy_(-beta, G.col(k))                                    # :222
            self.X.axpy_(beta, U.col(k))                                     # :223
            self.normR = norm(self.R)                                        # :225
            if self.smoothing:
                self._smooth()
            for i in range(k + 1, s):
                f[i] = f[i] - beta * M[i, k]                                 # :237-239
            nextstep = step + 1
        else:                                                                # step == s + 1  :242
            self.V.copyto_(self.R)                                           # :246
            self._ldiv(self.V)                                               # :249
            mul_(self.Q, self.A, self.V)                                     # :251
            ns, nt, ts = norm(self.R), norm(self.Q), dot(self.Q, self.R)     # omega(Q, R)  :70-82
            rho = abs(ts / (nt * ns))
            omega = ts / (nt * nt)
            if float(rho) < math.sqrt(2.) / 2:
                omega = omega * T(math.sqrt(2.) / 2) / rho
            self.omega = T(omega)
            self.R.axpy_(-self.omega, self.Q)                                # :253
            self.X.axpy_(self.omega, self.V)                                 # :254
            self.normR = norm(self.R)                                        # :256
            if self.smoothing:
                self._smooth()
            nextstep = 1
        if self.log is not None:
            self.log.nextiter_(mvps=1)                                       # :268-269
            self.log.push_("resnorm", self.normR)
        if self.verbose:
            print("%3d\t%3d\t%1.2e" % (it, step, self.normR))
        return self.normR, (it + 1, nextstep)

    def __iter__(self):
        state = (1, 1)
        while (nxt := self.iterate(state)) is not None:
            normR, state = nxt
            yield normR

    def __del__(self):
        try:
            if getattr(self, "_step", None) is not None and self.X.ctx.handle:
                lib().mik_idrs_destroy(self._step)
                self._step = None
        except Exception:
            pass


def idrs_iterable_(log, X, A, C_, s, Pl, abstol, reltol, maxiter, *, smoothing=False, verbose=False, P=None, fused=True):
    """``idrs_iterable!(log, X, A, C, s, Pl, abstol, reltol, maxiter; smoothing, verbose)`` -- src/idrs.jl:116-147."""
    return IDRSIterable(log, X, A, C_, s, Pl, abstol, reltol, maxiter, smoothing=smoothing, verbose=verbose, P=P, fused=fused)


def idrs_(x, A, b, *, s=8, Pl=None, abstol=0.0, reltol=None, maxiter=None, log=False, **kwargs):
    """``idrs!(x, A, b; s, Pl, abstol, reltol, maxiter, log, smoothing, verbose)`` -- src/idrs.jl:49-64 (and idrs_method!, :150-162)."""
    reltol = _default_reltol(b) if reltol is None else reltol
    maxiter = A.size(2) if maxiter is None else maxiter
    history = ConvergenceHistory(partial=not log)
    history["abstol"], history["reltol"] = abstol, reltol
    if log:
        history.reserve_("resnorm", maxiter)
    if kwargs.get("verbose"):
        print("=== idrs ===\n%4s\t%4s\t%7s" % ("iter", "step", "resnorm"))
    it = idrs_iterable_(history, x, A, b, s, Pl, abstol, reltol, maxiter, **kwargs)
    for _ in it:                                                             # reduce((_, r) -> r, iterable; init = iterable.normR)  :158
        pass
    if log:
        history.shrink_()
    return (it.X, history) if log else it.X


def idrs(A, b, **kwargs):
    """``idrs(A, b; s = 8, ...)`` -- src/idrs.jl:10."""
    return idrs_(zerox(A, b), A, b, **kwargs)


# ==============================================================================================
# lsqr.jl / lsmr.jl: rectangular operators and adjoint products
# ==============================================================================================
def adjoint(A):
    """``adjoint(A)`` (src/lsqr.jl:120, src/lsmr.jl:113): the operator ``with_adjoint`` uploaded next to A."""
    adj = getattr(A, "adj", None)
    if adj is None:
        raise MikError(5, "adjoint", "this operator was uploaded without its adjoint: create it with extras.with_adjoint(...) / with_adjoint_from_scipy(m)")
    return adj


def _eps(dtype):
    return np.finfo(np.dtype(dtype)).eps


def _hypot(T, a, b):
    """hypot in the element type through the C library (numpy's ufunc calls hypot / hypotf; math.hypot is Python's own algorithm and differs in
    the last bit now and then)"""
    return np.hypot(T(a), T(b))


def xpby_nrm2_(x, beta, y):
    """``y .= x .+ beta .* y`` and ``norm(y)`` in one sweep (``mik_xpby_nrm2``): the bidiagonalisation updates of LSQR / LSMR."""
    out = np.zeros(1, y.dtype)
    _, pb = _scalar(y.dtype, beta)
    check(lib().mik_xpby_nrm2(y.ctx.handle, y.code, y.n, _vp(x.ptr), pb, _vp(y.ptr), out.ctypes.data_as(_vp)), "mik_xpby_nrm2", y.ctx.handle)
    return out[0]


def lsqr_(x, A, b, *, damp=0, atol=None, btol=None, conlim=None, maxiter=None, verbose=False, log=False, fused=True):
    """``lsqr!(x, A, b; damp, atol, btol, conlim, maxiter, verbose, log)`` -- src/lsqr.jl:69-81 and lsqr_method! (:87-224), statement by
    statement as written in v0.9.4.  ``fused`` (device vectors): the two bidiagonalisation updates carry their norms (``mik_xpby_nrm2``) and the tail
    :189-192 is one sweep (``mik_lsqr_update``: wrho is never stored) -- same per-element operations, same bits; otherwise every vector statement
    is one L1 call (mul_, xpby_, scal_, axpy_, norm).  damp and the tolerances are taken in the element type (what the defaults are)."""
    T = x.dtype.type
    fused = bool(fused) and isinstance(x, HipVector)
    m, n = A.size(1), A.size(2)
    maxiter = max(m, n) if maxiter is None else int(maxiter)                 # :70
    history = ConvergenceHistory(partial=not log)
    for key in ("resnorm", "anorm", "rnorm", "cnorm"):
        history.reserve_(key, maxiter)                                       # :76
    if x.n != n or b.n != m:
        raise MikError(3, "lsqr_", "x should be of length %d, b of length %d" % (n, m))          # :95-96
    if not np.isfinite(float(norm(x))):
        raise MikError(1, "lsqr_", "Initial guess for x must be finite")    # :102-104 (a finite norm <=> every entry finite, up to overflow of the sum)
    sq = T(np.sqrt(_eps(x.dtype)))
    atol = sq if atol is None else T(atol)                                   # :88
    btol = sq if btol is None else T(btol)
    conlim = T(1) / sq if conlim is None else T(conlim)                      # :89
    damp = T(damp)
    if verbose:
        print("=== lsqr ===\n%4s\t%7s\t\t%7s\t\t%7s\t\t%7s" % ("iter", "resnorm", "anorm", "cnorm", "rnorm"))
    itn = istop = 0
    ctol = T(1) / conlim if conlim > 0 else T(0)                             # :105
    Anorm = Acond = ddnorm = res2 = xnorm = xxnorm = z = sn2 = T(0)          # :106
    cs2 = T(-1)
    dampsq = damp * damp                                                     # :108
    tmpm, tmpn = b.similar(), x.similar()                                    # :109-110
    history["atol"], history["btol"], history["ctol"] = atol, btol, ctol
    u = b.similar()
    mul_(u, A, x)
    u.xpby_(b, T(-1))                                                        # u = b - A*x  :116
    v = x.similar().copyto_(x)                                               # :117
    beta = norm(u)                                                           # :118
    alpha = T(0)
    At = adjoint(A)                                                          # :120
    if beta > 0:
        history.mtvps = 1
        u.scal_(T(1) / beta)
        mul_(v, At, u)
        alpha = norm(v)
    if alpha > 0:
        v.scal_(T(1) / alpha)
    w = x.similar().copyto_(v)                                               # :130
    wrho = x.similar()
    Arnorm = alpha * beta                                                    # :133
    if Arnorm == 0:
        if log:
            history.shrink_()                                                # lsqr! still runs shrink! after the early return (:79)
        return (x, history) if log else x                                    # :134-136
    rhobar = alpha
    phibar = bnorm = rnorm = beta                                            # :138-139
    while itn < maxiter and not history.isconverged:                         # :141
        history.nextiter_(mvps=1)
        itn += 1
        mul_(tmpm, A, v)                                                     # :150
        if fused:
            beta = xpby_nrm2_(tmpm, -alpha, u)                               # :151-152 in one sweep
        else:
            u.xpby_(tmpm, -alpha)                                            # u .= -alpha .* u .+ tmpm
            beta = norm(u)
        if beta > 0:
            history.mtvps += 1
            u.scal_(T(1) / beta)
            Anorm = np.sqrt(Anorm * Anorm + alpha * alpha + beta * beta + dampsq)       # :156
            mul_(tmpn, At, u)
            if fused:
                alpha = xpby_nrm2_(tmpn, -beta, v)                           # :159-160 in one sweep
            else:
                v.xpby_(tmpn, -beta)                                         # v .= -beta .* v .+ tmpn
                alpha = norm(v)
            if alpha > 0:
                v.scal_(T(1) / alpha)
        rhobar1 = np.sqrt(rhobar * rhobar + dampsq)                          # :168-172
        cs1 = rhobar / rhobar1
        sn1 = damp / rhobar1
        psi = sn1 * phibar
        phibar = cs1 * phibar
        rho = np.sqrt(rhobar1 * rhobar1 + beta * beta)                       # :176-183
        cs = rhobar1 / rho
        sn = beta / rho
        theta = sn * alpha
        rhobar = -cs * alpha
        phi = cs * phibar
        phibar = sn * phibar
        tau = sn * phi
        t1 = phi / rho                                                       # :186-187
        t2 = -theta / rho
        if fused:                                                            # :189-192 in one sweep, wrho never stored
            outn = np.zeros(1, x.dtype)
            sc = [_scalar(x.dtype, val) for val in (t1, t2, T(1) / rho)]
            check(lib().mik_lsqr_update(x.ctx.handle, x.code, x.n, sc[0][1], sc[1][1], sc[2][1], _vp(x.ptr), _vp(w.ptr), _vp(v.ptr),
                                        outn.ctypes.data_as(_vp)), "mik_lsqr_update", x.ctx.handle)
            ddnorm = ddnorm + outn[0]
        else:
            x.axpy_(t1, w)                                                   # x .+= t1*w
            w.xpby_(v, t2)                                                   # w = t2 .* w .+ v
            wrho.copyto_(w).scal_(T(1) / rho)                                # wrho .= w .* inv(rho)
            ddnorm = ddnorm + norm(wrho)                                     # ddnorm += norm(wrho)  (as written)
        delta = sn2 * rho                                                    # :196-205
        gambar = -cs2 * rho
        rhs = phi - delta * z
        zbar = rhs / gambar
        xnorm = np.sqrt(xxnorm + zbar * zbar)
        gamma = np.sqrt(gambar * gambar + theta * theta)
        cs2 = gambar / gamma
        sn2 = theta / gamma
        z = rhs / gamma
        xxnorm = xxnorm + z * z
        Acond = Anorm * np.sqrt(ddnorm)                                      # :211-216
        res1 = phibar * phibar
        res2 = res2 + psi * psi
        rnorm = np.sqrt(res1 + res2)
        Arnorm = alpha * abs(tau)
        r1sq = rnorm * rnorm - dampsq * xxnorm                               # :224-227
        r1norm = np.sqrt(abs(r1sq))
        if r1sq < 0:
            r1norm = -r1norm
        history.push_("resnorm", r1norm)
        with np.errstate(divide="ignore", invalid="ignore"):
            test1 = rnorm / bnorm                                            # :233-238
            test2 = Arnorm / (Anorm * rnorm)
            test3 = T(1) / Acond
            t1 = test1 / (T(1) + Anorm * xnorm / bnorm)
            rtol = btol + atol * Anorm * xnorm / bnorm
        history.push_("cnorm", test3)
        history.push_("anorm", test2)
        history.push_("rnorm", test1)
        if verbose:
            print("%3d\t%1.2e\t%1.2e\t%1.2e\t%1.2e" % (itn, r1norm, test2, test3, test1))
        if itn >= maxiter:                                                   # :248-259
            istop = 7
        if T(1) + test3 <= 1:
            istop = 6
        if T(1) + test2 <= 1:
            istop = 5
        if T(1) + t1 <= 1:
            istop = 4
        if test3 <= ctol:
            istop = 3
        if test2 <= atol:
            istop = 2
        if test1 <= rtol:
            istop = 1
        history.setconv(istop > 0)
    if log:
        history.shrink_()
    return (x, history) if log else x


def lsqr(A, b, **kwargs):
    """``lsqr(A, b; ...)`` -- src/lsqr.jl:10 (x = zerox(A, b): length size(A, 2))."""
    return lsqr_(HipVector(A.size(2), b.dtype, b.ctx).fill_(0), A, b, **kwargs)


def lsmr_(x, A, b, *, atol=1e-6, btol=1e-6, conlim=1e8, maxiter=None, lam=0, verbose=False, log=False, fused=True):
    """``lsmr!(x, A, b; atol, btol, conlim, maxiter, λ, verbose, log)`` -- src/lsmr.jl:67-82 and lsmr_method! (:86-287), statement by statement
    as written in v0.9.4.  atol / btol are Float64 like the defaults, so that with Float32 data ``rtol`` and the comparisons against them
    promote, as does everything downstream of ``minrbar = 1e100`` (condA, test3) -- like the reference.  ``fused`` (device vectors): the
    bidiagonalisation updates carry their norms (``mik_xpby_nrm2``) and :199-201 with ``norm(x)`` (:242) are one sweep (``mik_lsmr_update``)."""
    T = x.dtype.type
    fused = bool(fused) and isinstance(x, HipVector)
    m, n = A.size(1), A.size(2)
    maxiter = max(m, n) if maxiter is None else int(maxiter)                 # :68
    history = ConvergenceHistory(partial=not log)
    for key in ("anorm", "rnorm", "cnorm"):
        history.reserve_(key, maxiter)                                       # :72
    if x.n != n or b.n != m:
        raise MikError(3, "lsmr_", "x has length %d but should have length %d (b: %d, %d)" % (x.n, n, b.n, m))
    if verbose:
        print("=== lsmr ===\n%4s\t%7s\t\t%7s\t\t%7s" % ("iter", "anorm", "cnorm", "rnorm"))
    u = b.similar().copyto_(b)                                               # btmp  :76-77
    v, h, hbar = x.similar(), x.similar(), x.similar()                       # :78
    atol, btol = np.float64(atol), np.float64(btol)
    ctol = T(1.0 / conlim) if conlim > 0 else T(0)                           # :105
    lam = T(lam)
    tmp_u, tmp_v = b.similar(), x.similar()
    mul_(tmp_u, A, x)                                                        # :108
    u.sub_(tmp_u)                                                            # b .-= tmp_u; u = b
    with np.errstate(divide="ignore", invalid="ignore"):
        beta = norm(u)
        u.scal_(T(1) / beta)                                                 # :112
        At = adjoint(A)
        mul_(v, At, u)                                                       # :114
        alpha = norm(v)
        v.scal_(T(1) / alpha)                                                # :116
    history["atol"], history["btol"], history["ctol"] = atol, btol, ctol
    zetabar = alpha * beta                                                   # :123-128
    alphabar = alpha
    rho = rhobar = cbar = T(1)
    sbar = T(0)
    h.copyto_(v)                                                             # :130
    hbar.fill_(0)
    betadd = beta                                                            # :134-140
    betad = T(0)
    rhodold = T(1)
    tautildeold = thetatilde = zeta = d = T(0)
    normA2 = alpha * alpha                                                   # :144
    maxrbar = T(0)
    minrbar = np.float64(1e100)                                              # :146
    normb = beta
    istop = 0
    normAr = alpha * beta
    it = 0
    history.mvps = 1                                                         # :154-155
    history.mtvps = 1
    if normAr != 0:
        while it < maxiter:
            history.nextiter_(mvps=1)
            it += 1
            mul_(tmp_u, A, v)                                                # :160
            if fused:
                beta = xpby_nrm2_(tmp_u, -alpha, u)                          # :161-162 in one sweep
            else:
                u.xpby_(tmp_u, -alpha)                                       # u .= tmp_u .+ u .* -α
                beta = norm(u)
            if beta > 0:
                history.mtvps += 1
                u.scal_(T(1) / beta)
                mul_(tmp_v, At, u)                                           # :166
                if fused:
                    alpha = xpby_nrm2_(tmp_v, -beta, v)                      # :167-168 in one sweep
                else:
                    v.xpby_(tmp_v, -beta)                                    # v .= tmp_v .+ v .* -β
                    alpha = norm(v)
                with np.errstate(divide="ignore"):
                    v.scal_(T(1) / alpha)
            alphahat = _hypot(T, alphabar, lam)     # :175-177
            chat = alphabar / alphahat
            shat = lam / alphahat
            rhoold = rho                                                     # :180-185
            rho = _hypot(T, alphahat, beta)
            c = alphahat / rho
            s_ = beta / rho
            thetanew = s_ * alpha
            alphabar = c * alpha
            rhobarold = rhobar                                               # :188-196
            zetaold = zeta
            thetabar = sbar * rho
            rhotemp = cbar * rho
            rhobar = _hypot(T, cbar * rho, thetanew)
            cbar = cbar * rho / rhobar
            sbar = thetanew / rhobar
            zeta = cbar * zetabar
            zetabar = -sbar * zetabar
            normx_fused = None
            if fused:                                                        # :199-201 and norm(x) (:242) in one sweep
                outn = np.zeros(1, x.dtype)
                sc = [_scalar(x.dtype, val) for val in (-thetabar * rho / (rhoold * rhobarold), zeta / (rho * rhobar), -thetanew / rho)]
                check(lib().mik_lsmr_update(x.ctx.handle, x.code, x.n, sc[0][1], sc[1][1], sc[2][1], _vp(hbar.ptr), _vp(h.ptr), _vp(x.ptr), _vp(v.ptr),
                                            outn.ctypes.data_as(_vp)), "mik_lsmr_update", x.ctx.handle)
                normx_fused = outn[0]
            else:
                hbar.xpby_(h, -thetabar * rho / (rhoold * rhobarold))        # hbar .= hbar .* (...) .+ h  :199
                x.axpy_(zeta / (rho * rhobar), hbar)                         # :200
                h.xpby_(v, -thetanew / rho)                                  # h .= h .* (-θnew / ρ) .+ v  :201
            betaacute = chat * betadd                                        # :206-211
            betacheck = -shat * betadd
            betahat = c * betaacute
            betadd = -s_ * betaacute
            thetatildeold = thetatilde                                       # :214-220
            rhotildeold = _hypot(T, rhodold, thetabar)
            ctildeold = rhodold / rhotildeold
            stildeold = thetabar / rhotildeold
            thetatilde = stildeold * rhobar
            rhodold = ctildeold * rhobar
            betad = -stildeold * betad + ctildeold * betahat
            tautildeold = (zetaold - thetatildeold * tautildeold) / rhotildeold          # :222-225
            taud = (zeta - thetatilde * tautildeold) / rhodold
            d = d + betacheck * betacheck
            e = betad - taud
            normr = np.sqrt(d + e * e + betadd * betadd)
            normA2 = normA2 + beta * beta                                    # :228-230
            normA = np.sqrt(normA2)
            normA2 = normA2 + alpha * alpha
            maxrbar = max(maxrbar, rhobarold)                                # :233-237
            if it > 1:
                minrbar = min(minrbar, np.float64(rhobarold))
            condA = np.float64(max(maxrbar, rhotemp)) / min(minrbar, np.float64(rhotemp))
            normAr = abs(zetabar)                                            # :241-242
            normx = norm(x) if normx_fused is None else normx_fused
            with np.errstate(divide="ignore", invalid="ignore"):
                test1 = normr / normb                                        # :245-247
                test2 = normAr / (normA * normr)
                test3 = np.float64(1.0) / condA
                t1 = test1 / (T(1) + normA * normx / normb)                  # :252
                rtol = btol + atol * np.float64(normA) * np.float64(normx) / np.float64(normb)   # :253
            history.push_("cnorm", test3)
            history.push_("anorm", test2)
            history.push_("rnorm", test1)
            if verbose:
                print("%3d\t%1.2e\t%1.2e\t%1.2e" % (it, test2, test3, test1))
            if it >= maxiter:                                                # :254-260
                istop = 7
                break
            if np.float64(1.0) + test3 <= 1:
                istop = 6
                break
            if T(1) + test2 <= 1:
                istop = 5
                break
            if T(1) + t1 <= 1:
                istop = 4
                break
            if test3 <= np.float64(ctol):
                istop = 3
                break
            if np.float64(test2) <= atol:
                istop = 2
                break
            if np.float64(test1) <= rtol:
                istop = 1
                break
    history.setconv(istop not in (3, 6, 7))                                  # :285
    if log:
        history.shrink_()
    return (x, history) if log else x


def lsmr(A, b, **kwargs):
    """``lsmr(A, b; ...)`` -- src/lsmr.jl:7."""
    return lsmr_(HipVector(A.size(2), b.dtype, b.ctx).fill_(0), A, b, **kwargs)


# ==============================================================================================
# qmr.jl
# ==============================================================================================
class LanczosDecomp:
    """``LanczosDecomp`` -- src/qmr.jl:5-58: the two-sided Lanczos process on A and adjoint(A); real element types."""

    def __init__(self, x, A, b, *, initially_zero=False, fused=True):
        T = x.dtype.type
        self.fused = bool(fused) and isinstance(x, HipVector)   # two axpy! (+ the dot behind them) per sweep, both scalings in one (same bits)
        self.A, self.At = A, adjoint(A)                                      # :51
        self.v_prev, self.v_curr, self.v_next = x.zero(), x.similar().copyto_(b), x.similar()        # :27-29
        if not initially_zero:
            mul_(self.v_next, A, x)                                          # :33
            self.v_curr.axpy_(T(-1), self.v_next)                            # :34
        self.resnorm = norm(self.v_curr)                                     # :36
        with np.errstate(divide="ignore"):
            self.v_curr.scal_(T(1) / self.resnorm)                           # :37
        self.w_prev, self.w_curr, self.w_next = x.zero(), x.similar().copyto_(self.v_curr), x.similar()   # :39-41
        self.alpha = self.beta_prev = self.beta_curr = self.delta = T(0)     # :43-46

    def iterate(self, iteration=1):
        """``iterate(l::LanczosDecomp, iteration)`` -- src/qmr.jl:62-101; None on a breakdown (delta == 0), vectors unrotated."""
        T = self.v_curr.dtype.type
        mul_(self.v_next, self.A, self.v_curr)                               # :67
        self.alpha = dot(self.v_next, self.w_curr)                           # :69
        if self.fused:
            _axpy2_dot(self.v_next, -self.alpha, self.v_curr, -self.beta_curr, self.v_prev if iteration > 1 else None, None)      # :70-72
            mul_(self.w_next, self.At, self.w_curr)                          # :75
            vw = _axpy2_dot(self.w_next, -self.alpha, self.w_curr, -self.delta, self.w_prev if iteration > 1 else None, self.v_next)   # :76-81
        else:
            self.v_next.axpy_(-self.alpha, self.v_curr)                      # :70
            if iteration > 1:
                self.v_next.axpy_(-self.beta_curr, self.v_prev)              # :72
            mul_(self.w_next, self.At, self.w_curr)                          # :75
            self.w_next.axpy_(-self.alpha, self.w_curr)                      # :76
            if iteration > 1:
                self.w_next.axpy_(-self.delta, self.w_prev)                  # :78
            vw = dot(self.v_next, self.w_next)                               # :81
        self.delta = np.sqrt(abs(vw))                                        # :82
        if self.delta == 0:
            return None                                                      # :83-85
        self.beta_prev = self.beta_curr                                      # :87-88
        self.beta_curr = vw / self.delta
        if self.fused:
            sa, sb = _scalar(self.v_next.dtype, T(1) / self.delta), _scalar(self.v_next.dtype, T(1) / self.beta_curr)
            check(lib().mik_scal2(self.v_next.ctx.handle, self.v_next.code, self.v_next.n, sa[1], _vp(self.v_next.ptr), sb[1], _vp(self.w_next.ptr)),
                  "mik_scal2", self.v_next.ctx.handle)                       # :90-91
        else:
            self.v_next.scal_(T(1) / self.delta)                             # :90
            self.w_next.scal_(T(1) / self.beta_curr)                         # :91
        self.w_next, self.w_curr, self.w_prev = self.w_prev, self.w_next, self.w_curr     # :93
        self.v_next, self.v_curr, self.v_prev = self.v_prev, self.v_next, self.v_curr     # :94
        return None, iteration + 1


def _axpy2_dot(y, a, x1, b, x2, z):
    """``y .+= a .* x1; y .+= b .* x2`` (x2 may be None) and ``dot(y, z)`` (z may be None) in one sweep (``mik_axpy2_dot``)."""
    out = np.zeros(1, y.dtype)
    sa, sb = _scalar(y.dtype, a), _scalar(y.dtype, b)
    check(lib().mik_axpy2_dot(y.ctx.handle, y.code, y.n, sa[1], _vp(x1.ptr), sb[1], _vp(x2.ptr if x2 is not None else None), _vp(y.ptr),
                              _vp(z.ptr if z is not None else None), out.ctypes.data_as(_vp)), "mik_axpy2_dot", y.ctx.handle)
    return out[0]


class QMRIterable:
    """``QMRIterable`` -- src/qmr.jl:103-121, construction per ``qmr_iterable!`` (:123-154).  ``fused`` (device vectors): the Lanczos step's
    axpy! pairs, its scalings and the tail :188-197 are single sweeps (``mik_axpy2_dot``, ``mik_scal2``, ``mik_qmr_update``); same bits."""

    def __init__(self, x, A, b, *, abstol, reltol, maxiter, initially_zero=False, fused=True):
        T = x.dtype.type
        self.x = x
        self.fused = bool(fused) and isinstance(x, HipVector)
        self.lanczos = LanczosDecomp(x, A, b, initially_zero=initially_zero, fused=fused)  # :131
        self.resnorm = self.lanczos.resnorm
        self.g = np.array([self.resnorm, 0], x.dtype)                        # :134
        self.H = np.zeros(4, x.dtype)
        self.c_prev, self.s_prev, self.c_curr, self.s_curr = T(1), T(0), T(1), T(0)       # :137-138
        self.p_prev, self.p_curr = x.zero(), x.zero()                        # :140-141
        self.tol = max(T(reltol) * self.lanczos.resnorm, T(abstol))          # :143
        self.maxiter = int(maxiter)

    def converged(self):
        return self.resnorm <= self.tol                                      # :156

    def start(self):
        return 1

    def done(self, iteration):
        return iteration > self.maxiter or self.converged()                  # :158

    def iterate(self, iteration=None):
        """``iterate(q::QMRIterable, iteration)`` -- src/qmr.jl:160-207 (the Lanczos step's return value is ignored, :165, as written)."""
        iteration = 1 if iteration is None else iteration
        if self.done(iteration):
            return None
        T, H, g, lz = self.x.dtype.type, self.H, self.g, self.lanczos
        lz.iterate(iteration)                                                # :165
        H[1] = lz.beta_prev                                                  # :167-169
        H[2] = lz.alpha
        H[3] = lz.delta
        if iteration > 2:                                                    # :171-174
            H[0] = self.s_prev * H[1]
            H[1] = self.c_prev * H[1]
        if iteration > 1:                                                    # :176-180
            tmp = -self.s_curr * H[1] + self.c_curr * H[2]
            H[1] = self.c_curr * H[1] + self.s_curr * H[2]
            H[2] = tmp
        c, s, H[2] = givens_algorithm(H[2], H[3], self.x.dtype)              # :183
        g[1] = -s * g[0]                                                     # :185-186
        g[0] = c * g[0]
        if self.fused:                                                       # :188-197 in one sweep; the names rotate instead of the two copies
            with np.errstate(divide="ignore"):
                sc = [_scalar(self.x.dtype, val) for val in (-H[1], -H[0], T(1) / H[2], g[0])]
            check(lib().mik_qmr_update(self.x.ctx.handle, self.x.code, self.x.n, _vp(lz.v_prev.ptr), sc[0][1], _vp(self.p_curr.ptr if iteration > 1 else None),
                                       sc[1][1], _vp(self.p_prev.ptr if iteration > 2 else None), sc[2][1], sc[3][1], _vp(self.x.ptr), _vp(self.p_prev.ptr)),
                  "mik_qmr_update", self.x.ctx.handle)
            self.p_prev, self.p_curr = self.p_curr, self.p_prev
        else:
            lz.v_next.copyto_(lz.v_prev)                                     # we need v_m, not v_m+1  :188
            if iteration > 1:
                lz.v_next.axpy_(-H[1], self.p_curr)                          # :189
            if iteration > 2:
                lz.v_next.axpy_(-H[0], self.p_prev)                          # :190
            with np.errstate(divide="ignore"):
                lz.v_next.scal_(T(1) / H[2])                                 # :191
            self.x.axpy_(g[0], lz.v_next)                                    # :193
            self.p_prev.copyto_(self.p_curr)                                 # :196-197
            self.p_curr.copyto_(lz.v_next)
        self.c_prev, self.s_prev, self.c_curr, self.s_curr = self.c_curr, self.s_curr, c, s           # :195
        g[0] = g[1]                                                          # :198
        self.resnorm = abs(g[1])                                             # :200
        return self.resnorm, iteration + 1

    def __iter__(self):
        iteration = 1
        while (nxt := self.iterate(iteration)) is not None:
            resnorm, iteration = nxt
            yield resnorm


def qmr_iterable_(x, A, b, *, abstol=0.0, reltol=None, maxiter=None, initially_zero=False, lookahead=False, fused=True):
    """``qmr_iterable!`` -- src/qmr.jl:123-154."""
    return QMRIterable(x, A, b, abstol=abstol, reltol=_default_reltol(b) if reltol is None else reltol,
                       maxiter=A.size(2) if maxiter is None else maxiter, initially_zero=initially_zero, fused=fused)


def qmr_(x, A, b, *, abstol=0.0, reltol=None, maxiter=None, lookahead=False, log=False, initially_zero=False, verbose=False, fused=True):
    """``qmr!(x, A, b; ...)`` -- src/qmr.jl:256-297."""
    reltol = _default_reltol(b) if reltol is None else reltol
    maxiter = A.size(2) if maxiter is None else maxiter
    history = ConvergenceHistory(partial=not log)
    history["abstol"], history["reltol"] = abstol, reltol
    if log:
        history.reserve_("resnorm", maxiter)
    it = qmr_iterable_(x, A, b, abstol=abstol, reltol=reltol, maxiter=maxiter, initially_zero=initially_zero, fused=fused)
    if verbose:
        print("=== qmr ===\n%4s\t%7s" % ("iter", "resnorm"))
    for iteration, residual in enumerate(it, start=1):
        if log:
            history.nextiter_()                                              # :283
            history.push_("resnorm", residual)
        if verbose:
            print("%3d\t%1.2e" % (iteration, residual))
    if log:
        history.setconv(it.converged())
        history.shrink_()
    return (x, history) if log else x


def qmr(A, b, **kwargs):
    """``qmr(A, b; ...)`` -- src/qmr.jl:210."""
    return qmr_(zerox(A, b), A, b, initially_zero=True, **kwargs)


# ==============================================================================================
# simple.jl: power method
# ==============================================================================================
class PowerMethodIterable:
    """``PowerMethodIterable`` -- src/simple.jl:5-14, construction per ``powm_iterable!`` (:36-39); real element types."""

    def __init__(self, A, x, *, tol, maxiter):
        T = x.dtype.type
        self.A, self.x, self.tol, self.maxiter = A, x, T(tol), int(maxiter)
        self.theta = T(0)
        self.r, self.Ax = x.similar(), x.similar()
        self.residual = np.finfo(x.dtype).max                                # floatmax  :38

    def converged(self):
        return self.residual <= self.tol                                     # :15

    def start(self):
        return 0

    def done(self, iteration):
        return iteration > self.maxiter or self.converged()                  # :17

    def iterate(self, iteration=None):
        """``iterate(p::PowerMethodIterable, iteration)`` -- src/simple.jl:19-32."""
        iteration = 0 if iteration is None else iteration
        if self.done(iteration):
            return None
        T = self.x.dtype.type
        mul_(self.Ax, self.A, self.x)                                        # :22
        self.theta = dot(self.x, self.Ax)                                    # :25
        self.r.copyto_(self.Ax)                                              # :26
        self.r.axpy_(-self.theta, self.x)                                    # :27
        self.residual = norm(self.r)                                         # :28
        self.x.copyto_(self.Ax)                                              # :31
        with np.errstate(divide="ignore"):
            self.x.scal_(T(1) / norm(self.x))                                # :32
        return self.residual, iteration + 1

    def __iter__(self):
        iteration = 0
        while (nxt := self.iterate(iteration)) is not None:
            residual, iteration = nxt
            yield residual


def powm_iterable_(A, x, *, tol=None, maxiter=None):
    """``powm_iterable!(A, x; tol, maxiter)`` -- src/simple.jl:36-39."""
    tol = np.finfo(x.dtype).eps * A.size(2) ** 3 if tol is None else tol
    return PowerMethodIterable(A, x, tol=tol, maxiter=A.size(1) if maxiter is None else maxiter)


def powm_(B, x, *, tol=None, maxiter=None, shift=0, inverse=False, log=False, verbose=False):
    """``powm!(B, x; tol, maxiter, shift, inverse, log, verbose)`` -- src/simple.jl:113-142: (λ, x[, history]).  With ``log`` the history also
    carries the residual norm of every iterate (the reference reserves ``:resnorm`` but never pushes to it)."""
    T = x.dtype.type
    tol = np.finfo(x.dtype).eps * B.size(2) ** 3 if tol is None else tol     # :114
    maxiter = B.size(1) if maxiter is None else maxiter
    history = ConvergenceHistory(partial=not log)
    history["tol"] = tol
    history.reserve_("resnorm", maxiter + 1)
    if verbose:
        print("=== powm ===\n%4s\t%7s" % ("iter", "resnorm"))
    it = powm_iterable_(B, x, tol=tol, maxiter=maxiter)
    for iteration, residual in enumerate(it, start=1):
        history.nextiter_(mvps=1)                                            # :128
        history.push_("resnorm", residual)
        if verbose:
            print("%3d\t%1.2e" % (iteration, residual))
    history.setconv(it.converged())
    if log:
        history.shrink_()
    with np.errstate(divide="ignore"):
        lam = T(shift) + (T(1) / it.theta if inverse else it.theta)          # transform_eigenvalue  :34
    return (lam, it.x, history) if log else (lam, it.x)


def invpowm_(B, x0, **kwargs):
    """``invpowm!(B, x0; shift = σ, ...)`` -- src/simple.jl:185: B has the action of inv(A - σI) (any LinearOperator)."""
    return powm_(B, x0, inverse=True, **kwargs)
