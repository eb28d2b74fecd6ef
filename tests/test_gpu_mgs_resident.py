"""orthogonalize_and_normalize!(..., ModifiedGramSchmidt()) (src/orthogonalize.jl:67-79) in its "resident w" form (csrc/mik_mgs_res.h): beyond
2048 reduction segments one workgroup per compute unit keeps its part of w in registers and LDS between the passes.  Same arithmetic as the
multi-launch chain (MIK_KNOB_GS = 6 keeps the chain): residual history, solution and counters must agree BIT FOR BIT -- segments per workgroup odd
and even, vector lengths odd and even, workgroups with nothing but whole segments (the branch-free body) and with tails, fp64 and fp32 -- and with
the oracle where the oracle finishes in seconds.  GPU box only."""
import ctypes as C

import numpy as np
import pytest

from conftest import KN

pytestmark = pytest.mark.gpu


def form(pkg, it):
    s, g, xl, to = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    assert pkg.lib().mik_dev_gmres_form(it.handle, C.byref(s), C.byref(g), C.byref(xl), C.byref(to)) == 0
    return {"single": s.value, "G": g.value, "xl": xl.value, "timeouts": to.value}


def run(pkg, ctx, A, b, knob, restart, inner):
    ctx.set_tuning(KN.GS, knob)
    try:
        it = pkg.gmres_iterable_(pkg.zerox(A, b), A, b, restart=restart, orth_meth=pkg.ModifiedGramSchmidt(), initially_zero=True, reltol=0.0, maxiter=inner)
        f0 = form(pkg, it)
        h = it.iterate_many(0, inner)
        return h, it.x.to_numpy(), it.mv_products, f0, form(pkg, it)
    finally:
        ctx.set_tuning(KN.GS, 0)


@pytest.mark.parametrize("dtype,N", [(np.float64, 140), (np.float64, 161), (np.float64, 200), (np.float32, 200), (np.float32, 231)])
def test_resident_form_equals_the_chain_bit_for_bit(pkg, orc, ctx, dtype, N):
    """advection_dominated(N): 140 -> 11 segments per workgroup (odd: half of the last round is empty), 161 -> odd n (the last 16-byte group is half
    outside), 200 -> 31 per workgroup and a short last workgroup; fp32: 2048-element segments"""
    n, cp, rv, nz, b = pkg.fixtures.advection_dominated(N, 300.0)
    A = pkg.HipCSR(n, n, cp, rv, nz.astype(dtype), index_base=1)
    db = pkg.HipVector.from_numpy(b.astype(dtype))
    W, L = ctx.reduce_shape(dtype)
    nseg = -(-n // (256 * W * L))
    cus = ctx.info()["compute_units"]
    assert nseg > 8 * min(cus, 256)
    h1, x1, mv1, f1, f1b = run(pkg, ctx, A, db, 0, 7, 17)
    h0, x0, mv0, f0, _ = run(pkg, ctx, A, db, 6, 7, 17)
    assert f1["single"] == 1 and f1["G"] == -(-nseg // cus) and f1b["timeouts"] == 0           # the resident form ran, nothing timed out
    assert f0["single"] == 0                                                                    # MIK_KNOB_GS = 6: the chain
    assert np.array_equal(h1, h0) and np.array_equal(x1, x0) and mv1 == mv0
    h2, x2, _, _, f2 = run(pkg, ctx, A, db, 0, 7, 17)                                            # and again: same bits from run to run
    assert np.array_equal(h1, h2) and np.array_equal(x1, x2) and f2["timeouts"] == 0
    if N == 140:
        Ao = orc.CSC(n, cp, rv, nz, 1)
        xo, ho = orc.gmres(Ao, b, restart=7, orth_meth="mgs", maxiter=17, reltol=0.0, mode="tree", shape=(W, L))
        assert np.array_equal(h1, np.asarray(ho["resnorm"])) and np.array_equal(x1, xo)


@pytest.mark.parametrize("dtype,machine", [(np.float64, 0), (np.float32, 128 | (8 << 16))])
def test_resident_form_branch_free_body_on_whole_workgroups(pkg, ctx, dtype, machine):
    """the 256^3 Laplacian of configs[1]: 64 whole segments per workgroup in fp64 (every workgroup runs the body without bounds tests); fp32 planned for
    128 compute units has 64 per workgroup as well.  gmres!(restart = 6), 9 inner iterations, against the chain."""
    n, cp, rv, nz = pkg.fixtures.laplace_matrix(256, 3)
    ctx.set_tuning(KN.MACHINE, machine)
    try:
        A = pkg.HipCSR(n, n, cp, rv, nz.astype(dtype), index_base=1)
        b = pkg.HipVector.from_numpy(pkg.fixtures.hashed_rhs(n).astype(dtype))
        h1, x1, mv1, f1, f1b = run(pkg, ctx, A, b, 0, 6, 9)
        h0, x0, mv0, f0, _ = run(pkg, ctx, A, b, 6, 6, 9)
    finally:
        ctx.set_tuning(KN.MACHINE, 0)
    assert f1["single"] == 1 and f1["G"] == 64 and f1b["timeouts"] == 0 and f0["single"] == 0
    assert np.array_equal(h1, h0) and np.array_equal(x1, x0) and mv1 == mv0


def test_resident_form_timeout_falls_back_to_the_chain(pkg, ctx):
    n, cp, rv, nz, b = pkg.fixtures.advection_dominated(140, 300.0)
    A = pkg.HipCSR(n, n, cp, rv, nz, index_base=1)
    db = pkg.HipVector.from_numpy(b)
    h1, x1, _, _, _ = run(pkg, ctx, A, db, 0, 5, 8)
    pkg.lib().mik_set_tuning(KN.GS_TIMEOUT, 1)
    try:
        h2, x2, _, _, f2 = run(pkg, ctx, A, db, 0, 5, 8)
    finally:
        pkg.lib().mik_set_tuning(KN.GS_TIMEOUT, 0)
    assert f2["timeouts"] >= 1 and f2["single"] == 0 and np.array_equal(h1, h2) and np.array_equal(x1, x2)


@pytest.mark.parametrize("precond", ["none", "pl", "pl+pr"])
def test_resident_form_long_cycles_restarts_and_preconditioners(pkg, ctx, precond):
    """restart = 30 (columns k = 1 ... 30: up to 31 hand-offs per launch), two restarts, a start vector, and the three expand! methods of
    src/gmres.jl:285-304 (Identity / Pl / Pl and Pr as diagonal preconditioners) in front of the resident form -- against the chain, bit for bit"""
    n, cp, rv, nz, b = pkg.fixtures.advection_dominated(140, 300.0)
    A = pkg.HipCSR(n, n, cp, rv, nz, index_base=1)
    db = pkg.HipVector.from_numpy(b)
    rng = np.random.default_rng(11)
    x0 = rng.standard_normal(n)
    d1 = pkg.HipVector.from_numpy(1.0 + rng.random(n))
    d2 = pkg.HipVector.from_numpy(0.5 + rng.random(n))
    kw = dict(restart=30, orth_meth=pkg.ModifiedGramSchmidt(), reltol=0.0, maxiter=65)
    if precond != "none":
        kw["Pl"] = pkg.JacobiPrec(d1)
    if precond == "pl+pr":
        kw["Pr"] = pkg.JacobiPrec(d2)
    out = []
    for knob in (0, 6):
        ctx.set_tuning(KN.GS, knob)
        try:
            it = pkg.gmres_iterable_(pkg.HipVector.from_numpy(x0), A, db, **kw)
            f = form(pkg, it)
            h = it.iterate_many(0, 65)
            out.append((h, it.x.to_numpy(), it.mv_products, f, form(pkg, it)))
        finally:
            ctx.set_tuning(KN.GS, 0)
    (h1, x1, mv1, f1, f1b), (h0, x0_, mv0, f0, _) = out
    assert f1["single"] == 1 and f1b["timeouts"] == 0 and f0["single"] == 0 and h1.size == 65
    assert np.array_equal(h1, h0) and np.array_equal(x1, x0_) and mv1 == mv0


def test_resident_form_only_while_most_of_w_fits(pkg, ctx):
    """planned for 32 compute units: 2,680 segments -> 84 per workgroup (0.62 of w on the chip: resident form), 4,076 segments -> 128 per workgroup
    (0.41: the chain is faster there, scripts/micro/mgs_resident_big.py) -- decided at mik_gmres_create; same bits either way"""
    for N, resident in ((140, True), (161, False)):
        n, cp, rv, nz, b = pkg.fixtures.advection_dominated(N, 300.0)
        A = pkg.HipCSR(n, n, cp, rv, nz, index_base=1)
        db = pkg.HipVector.from_numpy(b)
        h_ref, x_ref, _, _, _ = run(pkg, ctx, A, db, 6, 6, 10)
        ctx.set_tuning(KN.MACHINE, 32 | (1 << 16))
        try:
            h, x, _, f, fb = run(pkg, ctx, A, db, 0, 6, 10)
        finally:
            ctx.set_tuning(KN.MACHINE, 0)
        assert (f["single"], f["G"]) == ((1, -(-(-(-n // 1024)) // 32)) if resident else (0, 0)), (N, f)
        assert fb["timeouts"] == 0 and np.array_equal(h, h_ref) and np.array_equal(x, x_ref)
