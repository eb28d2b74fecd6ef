"""IDR(s) (src/idrs.jl): the reference's property tests (test/idrs.jl) against the oracle, a second restatement in Python scalar
loops bit for bit against the oracle's SEQ mode (CPU), and bit-level parity of the device path -- one C call per step, mik_idrs_step --
and of its statement-by-statement form against the oracle's TREE mode (GPU).  The reference draws its shadow vectors with rand! (:136);
here they are an input everywhere."""
import math

import numpy as np
import pytest


def sprand_plus(rng, n, density, shift, dtype):
    import scipy.sparse as sp
    A = (sp.random(n, n, density=density, random_state=rng, format="csc") + shift * sp.identity(n)).tocsc().astype(dtype)   # test/idrs.jl:36,46
    A.sort_indices()
    return A


# ---- oracle: what test/idrs.jl checks ------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("smoothing", [False, True])
def test_oracle_idrs_dense(orc, dtype, smoothing):
    rng = np.random.default_rng(1234567)
    n = 10
    Ad = (rng.random((n, n)) + n * np.eye(n)).astype(dtype)                  # test/idrs.jl:16-17
    b = rng.random(n).astype(dtype)
    reltol = float(np.sqrt(np.finfo(dtype).eps))
    x, h = orc.idrs(orc.CSC.from_dense(Ad), b, P=rng.random((n, 8)), reltol=reltol, smoothing=smoothing)
    assert h["isconverged"]                                                  # :23, :31
    assert np.linalg.norm(Ad @ x - b) / np.linalg.norm(b) <= (2 if smoothing else 1) * reltol      # :24, :32


def test_oracle_idrs_sparse_and_preconditioned(orc):
    rng = np.random.default_rng(7)
    A = sprand_plus(rng, 10, 0.5, 10, np.float64)
    b = rng.random(10)
    x, h = orc.idrs(orc.CSC.from_scipy(A), b, P=rng.random((10, 8)))
    assert h["isconverged"] and np.linalg.norm(A @ x - b) / np.linalg.norm(b) <= np.sqrt(np.finfo(np.float64).eps)     # :40-42
    A = sprand_plus(rng, 1000, 0.1, 30, np.float64)                          # :46
    b = rng.random(1000)
    P = rng.random((1000, 8))
    x, h = orc.idrs(orc.CSC.from_scipy(A), b, P=P)
    assert h["isconverged"] and np.linalg.norm(A @ x - b) / np.linalg.norm(b) <= np.sqrt(np.finfo(np.float64).eps)     # :50-52
    xp, hp = orc.idrs(orc.CSC.from_scipy(A), b, P=P, pl_diag=A.diagonal())  # (the reference uses an incomplete LU; here the diagonal)
    assert hp["isconverged"] and np.linalg.norm(A @ xp - b) / np.linalg.norm(b) <= np.sqrt(np.finfo(np.float64).eps)   # :55-57
    assert np.allclose(x, xp, rtol=1e-3)                                     # :59


def test_oracle_idrs_maxiter_and_termination(orc):
    rng = np.random.default_rng(3)
    x, h = orc.idrs(orc.CSC.from_dense(rng.random((5, 5))), rng.random(5), P=rng.random((5, 8)), maxiter=2)
    assert h["iters"] == 2 and len(h["resnorm"]) == 2                        # test/idrs.jl:65-69
    for dtype in (np.float32, np.float64):                                   # :84-107
        A3 = np.array([[2, -1, 0], [-1, 2, -1], [0, -1, 2]], dtype)
        b3 = np.ones(3, dtype)
        x0 = np.linalg.solve(A3.astype(np.float64), b3.astype(np.float64)).astype(dtype)
        pert = (10 * np.sqrt(np.finfo(dtype).eps) * np.array([-1.0, 1.0, -1.0])).astype(dtype)
        P = rng.random((3, 8)).astype(dtype)
        x, ch = orc.idrs(orc.CSC.from_dense(A3), b3, x0 + pert, P=P)
        assert 2 <= ch["iters"] <= 3                                         # :98
        r0 = float(np.linalg.norm(A3 @ (x0 + pert) - b3))
        x, ch = orc.idrs(orc.CSC.from_dense(A3), b3, x0 + pert, P=P, abstol=2 * r0, reltol=0.0)
        assert ch["iters"] == 0                                              # :105
    A5, b5 = rng.random((5, 5)), rng.random(5)                               # "Near solution (#222)"  :72-82
    P5 = rng.random((5, 8))
    x1, _ = orc.idrs(orc.CSC.from_dense(A5), b5, rng.random(5), P=P5)
    x2, _ = orc.idrs(orc.CSC.from_dense(A5), b5, x1, P=P5)
    assert np.allclose(x2, x1)


# ---- a second restatement, written separately from the same reference lines ---------------------------
def py_idrs(A, b, x, P, s, abstol, reltol, maxiter, smoothing, pl_diag):
    """src/idrs.jl:116-147 and :164-272 with Python floats, one rounded operation per statement, left-to-right sums"""
    n = len(b)

    def mul(v):                                                              # mul!(y, A::SparseMatrixCSC, v)
        y = [0.0] * n
        for j in range(n):
            for p in range(int(A.colptr[j]) - A.index_base, int(A.colptr[j + 1]) - A.index_base):
                i = int(A.rowval[p]) - A.index_base
                y[i] = y[i] + float(A.nzval[p]) * v[j]
        return y

    def dot(u, v):
        t = 0.0
        for a, c in zip(u, v):
            t = t + a * c
        return t

    def nrm(u):
        return math.sqrt(dot(u, u))

    X = list(map(float, x))
    R = [bi - yi for bi, yi in zip(map(float, b), mul(X))]                   # :119
    normR = nrm(R)
    tol = max(reltol * normR, abstol)                                        # :121
    X_s, R_s = (list(X), list(R)) if smoothing else (None, None)
    Pc = [[float(P[i][j]) for i in range(n)] for j in range(s)]
    U = [[0.0] * n for _ in range(s)]
    G = [[0.0] * n for _ in range(s)]
    M = [[1.0 if i == j else 0.0 for j in range(s)] for i in range(s)]
    f = [0.0] * s
    omega = 1.0
    hist = []
    it, step = 1, 1
    while not (normR < tol or it > maxiter):                                 # :168
        if step <= s:
            if step == 1:
                f = [dot(Pc[i], R) for i in range(s)]                        # :179-181
            k = step - 1
            c = f[k:]                                                        # :187
            for j in range(k, s):
                c[j - k] = c[j - k] / M[j][j]
                for i in range(j + 1, s):
                    c[i - k] = c[i - k] - M[i][j] * c[j - k]
            V = [c[0] * g for g in G[k]]                                     # :188
            Q = [c[0] * u for u in U[k]]                                     # :189
            for i in range(k + 1, s):
                V = [v + c[i - k] * g for v, g in zip(V, G[i])]              # :192
                Q = [q + c[i - k] * u for q, u in zip(Q, U[i])]              # :193
            V = [r - v for r, v in zip(R, V)]                                # :197
            if pl_diag is not None:
                V = [v / float(d) for v, d in zip(V, pl_diag)]               # :200
            U[k] = [q + omega * v for q, v in zip(Q, V)]                     # :202
            G[k] = mul(U[k])                                                 # :203
            for i in range(k):                                               # :207-211
                alpha = dot(Pc[i], G[k]) / M[i][i]
                G[k] = [g - alpha * gi for g, gi in zip(G[k], G[i])]
                U[k] = [u - alpha * ui for u, ui in zip(U[k], U[i])]
            for i in range(k, s):
                M[i][k] = dot(Pc[i], G[k])                                   # :215-217
            beta = f[k] / M[k][k]                                            # :221
            R = [r - beta * g for r, g in zip(R, G[k])]                      # :222
            X = [xv + beta * u for xv, u in zip(X, U[k])]                    # :223
            normR = nrm(R)                                                   # :225
            nextstep = step + 1
        else:
            V = list(R)                                                      # :246
            if pl_diag is not None:
                V = [v / float(d) for v, d in zip(V, pl_diag)]
            Q = mul(V)                                                       # :251
            ns, nt, ts = nrm(R), nrm(Q), dot(Q, R)                           # :72-74
            rho = abs(ts / (nt * ns))
            omega = ts / (nt * nt)
            if rho < math.sqrt(2.) / 2:
                omega = omega * (math.sqrt(2.) / 2) / rho
            R = [r - omega * q for r, q in zip(R, Q)]                        # :253
            X = [xv + omega * v for xv, v in zip(X, V)]                      # :254
            normR = nrm(R)
            nextstep = 1
        if smoothing:                                                        # :226-235
            T_s = [a - r for a, r in zip(R_s, R)]
            gamma = dot(R_s, T_s) / dot(T_s, T_s)
            R_s = [a - gamma * t for a, t in zip(R_s, T_s)]
            X_s = [a - gamma * (a - xv) for a, xv in zip(X_s, X)]
            normR = nrm(R_s)
        if step <= s:
            k = step - 1
            for i in range(k + 1, s):
                f[i] = f[i] - beta * M[i][k]                                 # :237-239
        hist.append(normR)
        it, step = it + 1, nextstep
    return (X_s if smoothing else X), hist, bool(0 <= normR < tol)


@pytest.mark.parametrize("s,smoothing,precond", [(3, False, False), (1, False, False), (4, True, False), (3, False, True), (8, True, True)])
def test_python_idrs_equals_the_c_oracle_bit_for_bit(orc, s, smoothing, precond):
    rng = np.random.default_rng(11 + s)
    n = 9
    A = sprand_plus(rng, n, 0.5, 4, np.float64)
    Ao = orc.CSC.from_scipy(A)
    b, x0, P = rng.random(n), rng.random(n), rng.random((n, s))
    d = A.diagonal() if precond else None
    reltol = 1e-10
    xo, ho = orc.idrs(Ao, b, x0, P=P, s=s, pl_diag=d, reltol=reltol, maxiter=40, smoothing=smoothing)
    xp, hp, conv = py_idrs(Ao, b, x0, P, s, 0.0, reltol, 40, smoothing, d)
    assert ho["iters"] == len(hp) > s + 1 and ho["isconverged"] == conv and ho["mvps"] == len(hp)
    assert np.array_equal(ho["resnorm"], np.array(hp)) and np.array_equal(xo, np.array(xp))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("s,smoothing,precond,start", [(4, False, False, False), (3, True, False, True), (8, False, True, True), (1, True, True, False)])
def test_python_mirror_equals_the_c_oracle_on_a_host_double(pkg, orc, monkeypatch, dtype, s, smoothing, precond, start):
    """IDRSIterable (fused=False: the statement-by-statement form the device path also carries) with every vector statement evaluated by the
    oracle's SEQ primitives (tests/host_double.py): history, counters and solution equal the C restatement bit for bit, fp64 and fp32."""
    from importlib import import_module
    from host_double import FakeOperator, FakeVector, patch
    api = import_module(pkg.__name__ + ".extras")
    patch(monkeypatch, api, orc)
    rng = np.random.default_rng(31 + s)
    n = 30
    S = sprand_plus(rng, n, 0.3, 5, dtype)
    b, P = rng.random(n).astype(dtype), rng.random((n, s)).astype(dtype)
    x0 = rng.random(n).astype(dtype) if start else np.zeros(n, dtype)
    d = S.diagonal().astype(dtype) if precond else None

    class Diag:
        def ldiv_(self, v):
            v.a[:] = v.a / d
            return v
    xo, ho = orc.idrs(orc.CSC.from_scipy(S), b, x0 if start else None, P=P, s=s, pl_diag=d, maxiter=60, smoothing=smoothing)
    x, ch = api.idrs_(FakeVector(x0.copy()), FakeOperator(orc, S), FakeVector(b), s=s, Pl=Diag() if precond else None, maxiter=60, log=True,
                      smoothing=smoothing, P=P, fused=False)
    assert ch.iters == ho["iters"] >= min(s + 2, 8) and ch.mvps == ho["mvps"] and ch.isconverged == ho["isconverged"]
    assert np.array_equal(ch["resnorm"], ho["resnorm"]) and np.array_equal(x.to_numpy(), xo)


# ---- device ------------------------------------------------------------------------------------------
def _system(pkg, orc, name, dtype):
    import scipy.sparse as sp
    if name == "advdiff":
        A = orc.advdiff(12, 300.0)[0].astype(dtype)                          # non-symmetric: what IDR(s) is for
        b = pkg.fixtures.advection_dominated(12, 300.0)[4].astype(dtype)
    else:
        L0 = orc.laplace(11, 3)
        A = orc.CSC.from_scipy((L0.to_scipy() + 0.5 * sp.eye(L0.n)).tocsc()).astype(dtype)
        b = orc.hashed_rhs(A.n).astype(dtype)
    return A, b


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("name,s,smoothing,precond,start", [("advdiff", 8, False, False, False), ("advdiff", 4, True, False, True), ("advdiff", 8, False, True, True),
                                                           ("laplace", 1, False, False, False), ("laplace", 3, True, True, False), ("laplace", 8, True, False, True),
                                                           ("advdiff", 32, False, False, False)])
def test_idrs_device_bit_exact(pkg, orc, ctx, dtype, name, s, smoothing, precond, start):
    """idrs / idrs! through mik_idrs_step (one C call per step) and statement by statement through the L1 entry points: history, solution,
    iteration count and isconverged equal the oracle's (TREE mode, the device's reduction shape) bit for bit -- Identity and diagonal Pl,
    residual smoothing, zero and non-zero starting vectors, s = 1 (no bi-orthogonalisation), the reference's default s = 8, and the largest
    shadow space the fused path takes (32)."""
    A, b = _system(pkg, orc, name, dtype)
    n = A.n
    rng = np.random.default_rng(5)
    P = rng.random((n, s)).astype(dtype)
    x0 = rng.standard_normal(n).astype(dtype) if start else None
    S = A.to_scipy()
    d = (np.abs(S.diagonal()) * (1 + 0.1 * np.cos(np.arange(n)))).astype(dtype) if precond else None
    maxiter = 150
    xo, ho = orc.idrs(A, b, x0, P=P, s=s, pl_diag=d, maxiter=maxiter, smoothing=smoothing, mode="tree", shape=ctx.reduce_shape(dtype))
    assert ho["iters"] > 2 * (s + 1) or ho["isconverged"]
    dA = pkg.HipCSR(n, n, A.colptr, A.rowval, A.nzval)
    for fused in (True, False):
        kw = dict(s=s, P=P, maxiter=maxiter, smoothing=smoothing, log=True, fused=fused)
        if precond:
            kw["Pl"] = pkg.JacobiPrec(pkg.HipVector.from_numpy(d))
        if start:
            x, ch = pkg.extras.idrs_(pkg.HipVector.from_numpy(x0), dA, pkg.HipVector.from_numpy(b), **kw)
        else:
            x, ch = pkg.extras.idrs(dA, pkg.HipVector.from_numpy(b), **kw)
        assert ch.iters == ho["iters"] and ch.mvps == ho["mvps"] and ch.isconverged == ho["isconverged"], fused
        assert np.array_equal(ch["resnorm"], ho["resnorm"]), fused
        assert np.array_equal(x.to_numpy(), xo), fused
    if ho["isconverged"] and dtype == np.float64 and not smoothing:          # (the smoothed recurrence may drift from the true residual: both sides alike)
        assert np.linalg.norm(S @ xo - b) / np.linalg.norm(b) <= 3e-7


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,scale", [(np.float64, 1e-140), (np.float32, 1e-14), (np.float64, 1e137), (np.float32, 1e13)])
def test_idrs_device_on_badly_scaled_systems(pkg, orc, ctx, dtype, scale):
    """b (and with it R, U-updates and every norm) scaled outside the range where a plain sum of squares is safe (fp64: [2^-900, 2^900], fp32:
    [2^-70, 2^100]) but far enough inside the exponent range that omega's product norm(t) * norm(s) stays finite: the norms take the scaled
    recomputation on both sides (mik_safe_norm_slow / the oracle's safe_nrm_), everything else is the same arithmetic."""
    A, b = _system(pkg, orc, "advdiff", dtype)
    b = (b.astype(np.float64) * scale).astype(dtype)
    n = A.n
    P = np.random.default_rng(6).random((n, 4)).astype(dtype)
    xo, ho = orc.idrs(A, b, P=P, s=4, maxiter=40, mode="tree", shape=ctx.reduce_shape(dtype))
    assert np.all(np.isfinite(ho["resnorm"])) and ho["iters"] == 40
    dA = pkg.HipCSR(n, n, A.colptr, A.rowval, A.nzval)
    for fused in (True, False):
        x, ch = pkg.extras.idrs(dA, pkg.HipVector.from_numpy(b), s=4, P=P, maxiter=40, log=True, fused=fused)
        assert ch.iters == ho["iters"] and np.array_equal(ch["resnorm"], ho["resnorm"]) and np.array_equal(x.to_numpy(), xo), fused


@pytest.mark.gpu
def test_idrs_device_beyond_1024_segments(pkg, orc, ctx):
    """2.1 M unknowns: the bi-orthogonalisation chain can no longer finalise the previous dot inside the sweep (k_map_with takes up to 1024
    segment sums) and runs its separate finaliser; 14 steps (a full cycle of s = 4 twice and more) bit for bit, fused and statement by statement"""
    A = orc.laplace(129, 3)
    b = orc.hashed_rhs(A.n)
    P = np.random.default_rng(8).random((A.n, 4))
    xo, ho = orc.idrs(A, b, P=P, s=4, maxiter=14, reltol=0.0, mode="tree", shape=ctx.reduce_shape(np.float64))
    dA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval)
    for layout in ("auto", "csr"):
        dA.set_layout(layout)
        x, ch = pkg.extras.idrs(dA, pkg.HipVector.from_numpy(b), s=4, P=P, maxiter=14, reltol=0.0, log=True)
        assert ch.iters == 14 and np.array_equal(ch["resnorm"], ho["resnorm"]) and np.array_equal(x.to_numpy(), xo), layout


@pytest.mark.gpu
def test_idrs_state_and_argument_checks(pkg, orc, ctx):
    A, b = _system(pkg, orc, "advdiff", np.float64)
    n = A.n
    P = np.random.default_rng(9).random((n, 5))
    dA = pkg.HipCSR(n, n, A.colptr, A.rowval, A.nzval)
    its = [pkg.extras.idrs_iterable_(None, pkg.zerox(dA, pkg.HipVector.from_numpy(b)), dA, pkg.HipVector.from_numpy(b), 5, None, 0.0, 1e-8, 100, P=P, fused=f)
           for f in (True, False)]
    for it in its:
        state = (1, 1)
        for _ in range(9):                                                   # one cycle and a half
            _, state = it.iterate(state)
        assert state == (10, 4)
    (om1, M1, f1), (om2, M2, f2) = its[0].state(), its[1].state()
    assert om1 == om2 and np.array_equal(M1, M2) and np.array_equal(f1, f2) and np.all(np.triu(M1, 1) == 0) and om1 != 1.0
    import ctypes as C
    x = pkg.HipVector.from_numpy(np.zeros(n))
    hnd = C.c_void_p()
    lib = pkg.lib()
    args = (ctx.handle, dA.handle, 40, x.ptr, x.ptr, x.ptr, n, x.ptr, n, x.ptr, n, None, None, None, 1.0, C.byref(hnd))
    assert lib.mik_idrs_create(*args) == 5 and b"s = 40" in lib.mik_last_error(ctx.handle)                          # MIK_ERR_NOTIMPL
    args = (ctx.handle, dA.handle, 4, x.ptr, x.ptr, x.ptr, n, x.ptr, n, x.ptr, n, None, x.ptr, None, 1.0, C.byref(hnd))
    assert lib.mik_idrs_create(*args) == 1                                                                          # smoothing takes both vectors
    assert lib.mik_idrs_step(None, 1, None) == 1 and lib.mik_idrs_destroy(None) == 0
