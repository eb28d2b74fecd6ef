"""The machine is queried, not assumed: mik_ctx_info reports what hipDeviceGetAttribute said, and the selection paths
-- workgroups of the single-launch Gram-Schmidt (one per compute unit), its XCD-local form, the XCD strip maps of the banded SpMV --
derive their caps from it.  The development knob MIK_KNOB_MACHINE plans for another shape (32 CUs, 1 XCD = a CPX partition) on the
same device: the forms that do not fit are never launched (no time-out), the results keep the bits of the oracle
(src/orthogonalize.jl:67-79 has one result whatever the form).  GPU box only."""
import ctypes as C

import numpy as np
import pytest

from conftest import KN

pytestmark = pytest.mark.gpu

CPX = 32 | (1 << 16)            # compute units | XCDs << 16


def upload(pkg, A):
    return pkg.HipCSR(A.n, A.n, A.colptr, A.rowval, A.nzval, index_base=A.index_base)


def form(pkg, it):
    s, g, xl, to = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    assert pkg.lib().mik_dev_gmres_form(it.handle, C.byref(s), C.byref(g), C.byref(xl), C.byref(to)) == 0
    return {"single": s.value, "G": g.value, "xl": xl.value, "timeouts": to.value}


def test_ctx_info_reports_the_queried_machine(pkg, ctx):
    d = ctx.info()
    assert d["compute_units"] >= 1 and d["xcds"] >= 1 and d["wavefront_size"] == 64 and d["arch"].startswith("gfx950")
    assert d["compute_units"] % d["xcds"] == 0
    assert d["planned_compute_units"] == d["compute_units"] and d["planned_xcds"] == d["xcds"]
    assert d["resident_workgroup_cap"] == d["compute_units"] and d["gs_single_launch_max_segments"] == 8 * d["compute_units"]
    assert d["xcd_maps"] == (1 if d["xcds"] == 8 else 0)
    assert d["gs_xcd_local_max_workgroups"] == (4 * d["compute_units"] // d["xcds"] if d["xcds"] == 8 else 0)
    assert d["mgs_resident_max_segments"] == 86 * d["compute_units"]
    assert d["sweep_grid_cap"] == 32 * d["compute_units"] and d["hbm_bytes"] > 2 ** 30 and d["lds_bytes_per_cu"] >= 65536
    ctx.set_tuning(KN.MACHINE, CPX)
    try:
        o = ctx.info()
        assert (o["compute_units"], o["xcds"]) == (d["compute_units"], d["xcds"])              # the query is untouched ...
        assert (o["planned_compute_units"], o["planned_xcds"], o["xcd_maps"]) == (32, 1, 0)    # ... the plan follows the override
        assert o["resident_workgroup_cap"] == 32 and o["gs_single_launch_max_segments"] == 256 and o["gs_xcd_local_max_workgroups"] == 0
    finally:
        ctx.set_tuning(KN.MACHINE, 0)


@pytest.mark.parametrize("orth", ["mgs", "cgs", "dgks"])
@pytest.mark.parametrize("N,expect", [(12, "G1"), (40, "G2"), (60, "G8"), (70, "chains")])
def test_gmres_forms_follow_the_machine_shape_bit_exact(pkg, orc, ctx, N, expect, orth):
    """advection_dominated(N) planned for 32 CUs / 1 XCD: up to 32 segments one per workgroup, up to 256 with G = 2 / 4 / 8, beyond that the
    multi-launch chains (Modified Gram-Schmidt: the resident-w form on 31 workgroups) -- chosen at mik_gmres_create, never through the time-out; history, x and counters equal the run planned for
    the real machine and (MGS / CGS) the oracle."""
    A, b = orc.advdiff(N, 200.0)
    W, L = ctx.reduce_shape(np.float64)
    nseg = -(-A.n // (256 * W * L))
    M = {"mgs": pkg.ModifiedGramSchmidt(), "cgs": pkg.ClassicalGramSchmidt(), "dgks": pkg.DGKS()}[orth]
    db = pkg.HipVector.from_numpy(b)
    kw = dict(restart=9, orth_meth=M, maxiter=24, reltol=1e-10, initially_zero=True)
    dA = upload(pkg, A)
    it0 = pkg.gmres_iterable_(pkg.zerox(dA, db), dA, db, **kw)
    h0 = it0.iterate_many(0, 24)
    f0 = form(pkg, it0)
    x0 = it0.x.to_numpy()
    cus = ctx.info()["compute_units"]
    if nseg <= 8 * min(cus, 256):
        assert f0["single"] == 1 and f0["timeouts"] == 0
    ctx.set_tuning(KN.MACHINE, CPX)
    try:
        dB = upload(pkg, A)                                   # (the workgroup map of an operator is chosen at upload: identity here)
        it1 = pkg.gmres_iterable_(pkg.zerox(dB, db), dB, db, **kw)
        f_before = form(pkg, it1)
        h1 = it1.iterate_many(0, 24)
        f1 = form(pkg, it1)
        x1 = it1.x.to_numpy()
    finally:
        ctx.set_tuning(KN.MACHINE, 0)
    want = {"G1": (1, 1), "G2": (1, 2), "G8": (1, 8), "chains": (0, 0)}[expect]
    if expect == "chains" and orth == "mgs":
        want = (1, -(-nseg // 32))             # Modified Gram-Schmidt beyond 8 segments per CU: the resident-w form, ceil(nseg / CUs) segments per workgroup
    assert nseg <= 32 if expect == "G1" else True
    assert (f_before["single"], f_before["G"]) == want == (f1["single"], f1["G"]), (nseg, f_before, f1)
    assert f1["timeouts"] == 0 and f1["xl"] == 0              # nothing timed out; the XCD-local form is not planned on one XCD
    assert np.array_equal(h0, h1) and np.array_equal(x0, x1) and it0.mv_products == it1.mv_products
    if orth != "dgks":
        xo, ho = orc.gmres(A, b, restart=9, orth_meth=orth, maxiter=24, reltol=1e-10, mode="tree", shape=(W, L))
        assert np.array_equal(h1, np.asarray(ho["resnorm"])[:h1.size]) and h1.size == len(ho["resnorm"]) and np.array_equal(x1, xo)


def test_xcd_local_form_only_where_the_dispatch_is_the_one_it_was_written_for(pkg, orc, ctx):
    A, b = orc.advdiff(12, 1000.0)
    db = pkg.HipVector.from_numpy(b)
    dA = upload(pkg, A)
    it = pkg.gmres_iterable_(pkg.zerox(dA, db), dA, db, restart=10, maxiter=20, initially_zero=True)
    h = it.iterate_many(0, 20)
    f = form(pkg, it)
    assert f["single"] == 1 and f["timeouts"] == 0
    if ctx.info()["xcd_maps"]:
        assert f["xl"] == 1                                  # 8 XCDs: small Modified Gram-Schmidt columns run on one XCD
    ctx.set_tuning(KN.MACHINE, 256 | (4 << 16))              # four XCDs: the blockIdx % 8 rule does not hold
    try:
        it2 = pkg.gmres_iterable_(pkg.zerox(dA, db), dA, db, restart=10, maxiter=20, initially_zero=True)
        h2 = it2.iterate_many(0, 20)
        f2 = form(pkg, it2)
    finally:
        ctx.set_tuning(KN.MACHINE, 0)
    assert f2["single"] == 1 and f2["xl"] == 0 and f2["timeouts"] == 0 and np.array_equal(h, h2) and np.array_equal(it.x.to_numpy(), it2.x.to_numpy())


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_cg_and_spmv_without_xcd_maps_bit_exact(pkg, orc, ctx, dtype):
    """the strip map of a banded operator is a permutation of workgroups: planned for one XCD it is the identity; SpMV in every layout and the
    cg! history keep the oracle's bits"""
    A = orc.laplace(40, 3).astype(dtype)
    b = orc.hashed_rhs(A.n).astype(dtype)
    x = np.random.default_rng(3).standard_normal(A.n).astype(dtype)
    ctx.set_tuning(KN.MACHINE, CPX)
    try:
        dA = upload(pkg, A)
        for layout in ("auto", "csr"):
            dA.set_layout(layout)
            assert np.array_equal((dA @ pkg.HipVector.from_numpy(x)).to_numpy(), orc.spmv(A, x)), layout
            xs, ch = pkg.cg(dA, pkg.HipVector.from_numpy(b), log=True, maxiter=30, reltol=0.0)
            xo, ho = orc.cg(A, b, maxiter=30, reltol=0.0, mode="tree", shape=ctx.cg_shape(dtype))
            assert np.array_equal(ch["resnorm"], np.asarray(ho["resnorm"], dtype=np.float64)) and np.array_equal(xs.to_numpy(), xo), layout
    finally:
        ctx.set_tuning(KN.MACHINE, 0)
