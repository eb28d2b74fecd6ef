"""CG with the head of the next step enqueued ahead of the host wait, and with x .+= alpha .* u carried by the next sweep over u
(csrc/mik_krylov.hip, cg_enqueue_head; OpXpbyX): x, r and the
residual history after any mix of iterate / iterate_many calls equal those of the plain protocol (development knob MIK_KNOB_NO_LOOKAHEAD)
and the oracle, bit for bit -- including stops by tolerance, by maxiter, and a solve that is continued after a pause.
GPU box only."""
import numpy as np
import pytest

from conftest import KN

pytestmark = pytest.mark.gpu


def run(pkg, A, b, schedule, knobs, Pl=None, **kw):
    L = pkg.lib()
    for k, v in knobs.items():
        L.mik_set_tuning(k, v)
    try:
        dA = pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval, index_base=A.index_base)
        x = pkg.HipVector(A.n, b.dtype).fill_(0)
        it = pkg.cg_iterator_(x, dA, pkg.HipVector.from_numpy(b), Pl, initially_zero=True, **kw)
        hist, snaps, k = [], [], 0
        for steps in schedule:
            if steps == 1:
                nxt = it.iterate(k)
                if nxt is None:
                    break
                hist.append(nxt[0]); k += 1
            else:
                r = it.iterate_many(k, steps)
                hist.extend(r.tolist()); k += r.size
                if r.size < steps:
                    break
            snaps.append((x.to_numpy(), it.r.to_numpy()))          # what a caller sees between two calls
        run.last_u = it.u.to_numpy()
        return np.array(hist), snaps, it.converged if hasattr(it, "converged") else None
    finally:
        for k2 in knobs:
            L.mik_set_tuning(k2, 0)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("pcg", [False, True])
@pytest.mark.parametrize("N", [9, 10])
def test_lookahead_changes_nothing_the_caller_can_see(pkg, orc, ctx, dtype, pcg, N):
    A = orc.laplace(N, 3).astype(dtype)
    b = orc.hashed_rhs(A.n).astype(dtype)
    Pl = pkg.JacobiPrec(pkg.HipVector.from_numpy(np.full(A.n, 6.0, dtype))) if pcg else None
    schedule = [1, 1, 3, 1, 7, 1, 1, 25, 1, 1]
    kw = dict(reltol=0.0, maxiter=10 ** 6)
    h1, s1, _ = run(pkg, A, b, schedule, {}, Pl, **kw)
    # no look-ahead; x updated by the step's own sweep; neither; one row per lane
    for knobs in ({KN.NO_LOOKAHEAD: 1}, {KN.CG_STEP: KN.X_IN_STEP}, {KN.NO_LOOKAHEAD: 1, KN.CG_STEP: KN.X_IN_STEP}, {KN.SDIA_KERNEL: 3}, {KN.CG_STEP: KN.PCG_THREE_SWEEPS},
                  {KN.CG_STEP: KN.PCG_THREE_SWEEPS, KN.NO_LOOKAHEAD: 1}, {KN.HOST_WAIT: 3}):
        h0, s0, _ = run(pkg, A, b, schedule, knobs, Pl, **kw)
        assert np.array_equal(h1, h0) and len(s1) == len(s0), knobs
        for (x1, r1), (x0, r0) in zip(s1, s0):
            assert np.array_equal(x1, x0) and np.array_equal(r1, r0), knobs
    if not pcg:
        _, ho = orc.cg(A, b, maxiter=len(h1), mode="tree", shape=ctx.cg_shape(dtype), reltol=0.0)
        assert np.array_equal(h1, np.asarray(ho["resnorm"], dtype=np.float64)[:len(h1)])


@pytest.mark.parametrize("N", [8, 9])
def test_lookahead_at_the_stopping_tests(pkg, orc, ctx, N):
    A = orc.laplace(N, 3)
    b = orc.hashed_rhs(A.n)
    for kw in (dict(reltol=1e-6, maxiter=10 ** 6), dict(reltol=0.0, maxiter=13), dict(reltol=1e-3, maxiter=10 ** 6)):
        for schedule in ([1] * 200, [4] * 60, [1, 5, 1, 9] * 20):
            h1, s1, _ = run(pkg, A, b, schedule, {}, **kw)
            u1 = run.last_u
            for knobs in ({KN.NO_LOOKAHEAD: 1}, {KN.CG_STEP: KN.X_IN_STEP}, {KN.NO_LOOKAHEAD: 1, KN.CG_STEP: KN.X_IN_STEP}):
                h0, s0, _ = run(pkg, A, b, schedule, knobs, **kw)
                assert np.array_equal(h1, h0) and len(s1) == len(s0) and len(h1) > 0, knobs
                assert np.array_equal(s1[-1][0], s0[-1][0]) and np.array_equal(s1[-1][1], s0[-1][1]), knobs
                # the caller's u is the last direction: a head that ran ahead of a stopped iteration leaves it alone
                assert np.array_equal(u1, run.last_u), knobs
            if kw["maxiter"] == 13:
                assert len(h1) == 13
