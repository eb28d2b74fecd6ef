"""The Julia shim (iterativesolvers.jl_amd/julia/MIK.jl) cannot be executed here (no Julia toolchain), so it is kept in
step with the C ABI and with the tested Python mirror STATICALLY: every `ccall((:sym, libmik), ...)` must name a symbol
of include/mik.h with the right number of arguments, and the call sequences of the functions that replace reference
methods must equal the sequences api.py issues for the same methods."""
import os
import re

from conftest import ROOT

JL = open(os.path.join(ROOT, "iterativesolvers.jl_amd", "julia", "MIK.jl")).read()
CCALL = re.compile(r"ccall\(\(:(\w+), libmik\),\s*(\w+),\s*\(")


def ccalls(text):
    """[(symbol, return type, number of argument types)] in source order"""
    out = []
    for m in CCALL.finditer(text):
        i, depth, start = m.end(), 1, m.end()
        while depth:
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        types = text[start:i - 1]
        # split on top-level commas (Ref{Ptr{Cvoid}} contains none, NTuple would)
        n, d = (1 if types.strip().rstrip(",").strip() else 0), 0
        for ch in types.strip().rstrip(","):
            d += {"{": 1, "(": 1, "}": -1, ")": -1}.get(ch, 0)
            n += ch == "," and d == 0
        out.append((m.group(1), m.group(2), n))
    return out


def julia_function(name):
    """source text of `function <name>(...) ... end` (first definition)"""
    i = JL.index("function " + name)
    j = JL.index("\nend\n", i)
    return JL[i:j]


def test_every_ccall_matches_the_c_abi(pkg):
    sig = pkg._lib.SIGNATURES
    calls = ccalls(JL)
    assert len(calls) >= 40
    for sym, ret, nargs in calls:
        assert sym in sig, f"MIK.jl calls {sym}, which include/mik.h / _lib.py do not declare"
        assert nargs == len(sig[sym][1]), f"{sym}: MIK.jl passes {nargs} arguments, the C ABI takes {len(sig[sym][1])}"
        assert ret in ("Cint", "Cstring"), (sym, ret)
        assert (ret == "Cstring") == (sym == "mik_last_error")
    # the reference-facing surface is bound
    bound = {c[0] for c in calls}
    for must in ("mik_csr_create", "mik_spmv", "mik_dot", "mik_nrm2", "mik_axpy", "mik_xpby", "mik_sub", "mik_scal", "mik_divide", "mik_fill",
                 "mik_copy", "mik_cg_create", "mik_cg_create_op", "mik_cg_iterate", "mik_cg_iterate_many", "mik_cg_state", "mik_gmres_create",
                 "mik_gmres_create_op", "mik_gmres_iterate", "mik_gmres_state", "mik_gmres_create_partitioned", "mik_comm_create",
                 "mik_cgd_create", "mik_cgd_set_halo_plan", "mik_cgd_set_comm", "mik_cgd_init", "mik_cgd_iterate_many"):
        assert must in bound, must


def test_call_sequences_equal_the_python_mirror():
    """the C calls behind cg_iterator! / iterate / gmres_iterable! in MIK.jl, in order, are the ones api.py makes"""
    api = open(os.path.join(ROOT, "iterativesolvers.jl_amd", "api.py")).read() + "\n" + open(os.path.join(ROOT, "iterativesolvers.jl_amd", "extras.py")).read()

    def py_calls(cls, method):
        i = api.index(f"class {cls}")
        j = api.index(f"    def {method}(", i)
        k = api.index("\n    def ", j + 10)
        return re.findall(r"lib\(\)\.(mik_\w+)\(", api[j:k])
    jl = lambda f: [c[0] for c in ccalls(julia_function(f)) if not c[0].endswith("_destroy")]      # finalizers aside
    assert jl("IterativeSolvers.cg_iterator!") == ["mik_cg_create", "mik_cg_create_op"] == py_calls("CGIterable", "__init__")
    assert jl("Base.iterate(it::HipCGIterable") == ["mik_cg_iterate"] == py_calls("CGIterable", "iterate")
    assert jl("refresh!(it::HipCGIterable") == ["mik_cg_state"] == py_calls("CGIterable", "_refresh")
    assert jl("IterativeSolvers.gmres_iterable!") == ["mik_gmres_create", "mik_gmres_create_op"] == py_calls("GMRESIterable", "__init__")
    assert jl("Base.iterate(g::HipGMRESIterable") == ["mik_gmres_iterate"] == py_calls("GMRESIterable", "iterate")
    assert jl("refresh!(g::HipGMRESIterable") == ["mik_gmres_state"] == py_calls("GMRESIterable", "_refresh")
    # the widened solvers: one C call per iteration on both sides (the Python mirror keeps the statement-by-statement path next to it)
    assert jl("IterativeSolvers.bicgstabl_iterator!") == ["mik_bicgstab_create"] == py_calls("BiCGStabIterable", "__init__")
    assert jl("Base.iterate(it::HipBiCGStabIterable") == ["mik_bicgstab_step"] == py_calls("BiCGStabIterable", "iterate")[:1]
    assert jl("IterativeSolvers.minres_iterable!") == ["mik_minres_create"] == py_calls("MINRESIterable", "__init__")
    assert jl("Base.iterate(m::HipMINRESIterable") == ["mik_minres_step"] == py_calls("MINRESIterable", "iterate")[:1]
    assert jl("IterativeSolvers.idrs_iterable!") == ["mik_idrs_create"] == py_calls("IDRSIterable", "__init__")
    assert jl("Base.iterate(it::HipIDRSIterable") == ["mik_idrs_step"] == py_calls("IDRSIterable", "iterate")[:1]
    # adjoint(A) for lsqr! / lsmr! / qmr!: a second upload of the same arrays with is_csc = 0 on both sides
    wa = julia_function("with_adjoint")
    assert "pointer(A.colptr), pointer(A.rowval), pointer(A.nzval), 1, 0, h)" in wa and "size(A, 2), size(A, 1), nnz(A)" in wa
    i = api.index("def with_adjoint")
    assert "is_csc=False" in api[i:api.index("return A", i)]


def test_shim_has_no_silent_scalar_fallback_and_reference_defaults():
    assert "Array(v)[i]" in JL and "scalar_indexing_allowed[] || error(" in JL          # getindex fails loudly unless allowed
    assert "LinearAlgebra.ldiv!(y::HipVector{T}, P::HipJacobi{T}, x::HipVector{T})" in JL
    assert "Base.BroadcastStyle(::Type{<:HipVector}) = HipStyle()" in JL
    part = julia_function("gmres_iterable_partitioned!")
    assert "restart::Int = min(20, n_global)" in part and "maxiter::Int = n_global" in part and "orth_meth::OrthogonalizationMethod = ModifiedGramSchmidt()" in part
    # every finalizer that hands a context to the library checks that the context is still alive
    for m in re.finditer(r"finalizer\((\w+) -> ([^\n]+)", JL):
        assert "alive(" in m.group(2), m.group(0)
