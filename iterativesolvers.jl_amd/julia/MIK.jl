# MIK.jl -- Julia-side shim: binds libmik.so (include/mik.h) behind IterativeSolvers.jl's own
# dispatch points so that the UNMODIFIED `cg!` / `gmres!` drivers run the HIP path.
#
# STATUS: written against include/mik.h and IterativeSolvers v0.9.4; NOT EXECUTED -- no Julia
# toolchain exists in the build environment (SURVEY.md F3).  The Python mirror
# (iterativesolvers.jl_amd/api.py) issues the same C calls in the same order and IS tested on
# MI355X; keep the two in step.
#
# Plug points (file:line in the reference):
#   cg!(x, A, b; ...)        src/cg.jl:209  calls cg_iterator!(x, A, b, Pl; ...) at :224
#       -> method cg_iterator!(x::HipVector, A::HipCSR, b::HipVector, Pl; ...) below returns a
#          HipCGIterable; the driver's loop (:229-235) then uses Base.iterate, `iterable.residual`,
#          `iterable.mv_products`, `iterable.x` and IterativeSolvers.converged -- all defined here.
#   gmres!(x, A, b; ...)     src/gmres.jl:184 calls gmres_iterable!(x, A, b; ...) at :200
#       -> method gmres_iterable!(x::HipVector, A::HipCSR, b::HipVector; ...) returns a
#          HipGMRESIterable (needed because ArnoldiDecomp pins V to a host Matrix, src/gmres.jl:7,13).
#   Generic code paths (any other solver of the package) see HipVector/HipCSR through
#   mul!, dot, norm, axpy!, axpby!, rmul!, copyto!, fill!, similar, zero.
module MIK

using LinearAlgebra
using SparseArrays
import IterativeSolvers
import IterativeSolvers: Identity, OrthogonalizationMethod, ModifiedGramSchmidt, ClassicalGramSchmidt, DGKS

const libmik = get(ENV, "LIBMIK", joinpath(@__DIR__, "..", "libmik.so"))

const MIK_F64 = Cint(0); const MIK_F32 = Cint(1)
dtype_code(::Type{Float64}) = MIK_F64
dtype_code(::Type{Float32}) = MIK_F32
const MikFloat = Union{Float32, Float64}

struct MikError <: Exception
    code::Cint
    where::String
    detail::String
end

function check(code::Cint, where::AbstractString, ctx::Ptr{Cvoid} = C_NULL)
    code == 0 && return
    msg = ccall((:mik_last_error, libmik), Cstring, (Ptr{Cvoid},), ctx)
    throw(MikError(code, where, msg == C_NULL ? "" : unsafe_string(msg)))
end

# ---- context ------------------------------------------------------------------------------------
mutable struct Context
    handle::Ptr{Cvoid}
    function Context(device::Integer = 0)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:mik_ctx_create, libmik), Cint, (Cint, Ref{Ptr{Cvoid}}), device, h), "mik_ctx_create")
        ctx = new(h[])
        finalizer(c -> ccall((:mik_ctx_destroy, libmik), Cint, (Ptr{Cvoid},), c.handle), ctx)
        ctx
    end
end
const default_ctx = Ref{Union{Nothing, Context}}(nothing)
context() = (default_ctx[] === nothing && (default_ctx[] = Context(0)); default_ctx[]::Context)

# ---- device vector ------------------------------------------------------------------------------
mutable struct HipVector{T<:MikFloat} <: AbstractVector{T}
    ptr::Ptr{Cvoid}
    n::Int
    ctx::Context
    function HipVector{T}(::UndefInitializer, n::Integer, ctx::Context = context()) where {T<:MikFloat}
        p = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:mik_malloc, libmik), Cint, (Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}), ctx.handle, n * sizeof(T), p), "mik_malloc", ctx.handle)
        v = new{T}(p[], n, ctx)
        finalizer(x -> ccall((:mik_free, libmik), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), x.ctx.handle, x.ptr), v)
        v
    end
end
function HipVector(a::Vector{T}, ctx::Context = context()) where {T<:MikFloat}
    v = HipVector{T}(undef, length(a), ctx)
    GC.@preserve a check(ccall((:mik_memcpy_h2d, libmik), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t),
                               ctx.handle, v.ptr, pointer(a), sizeof(a)), "mik_memcpy_h2d", ctx.handle)
    v
end
Base.size(v::HipVector) = (v.n,)
Base.similar(v::HipVector{T}) where {T} = HipVector{T}(undef, v.n, v.ctx)
Base.similar(v::HipVector, ::Type{T}, n::Integer) where {T<:MikFloat} = HipVector{T}(undef, n, v.ctx)
Base.similar(v::HipVector, ::Type{T}, dims::Tuple{Int}) where {T<:MikFloat} = HipVector{T}(undef, dims[1], v.ctx)
Base.zero(v::HipVector{T}) where {T} = fill!(similar(v), zero(T))
function Base.Array(v::HipVector{T}) where {T}
    a = Vector{T}(undef, v.n)
    GC.@preserve a check(ccall((:mik_memcpy_d2h, libmik), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t),
                               v.ctx.handle, pointer(a), v.ptr, sizeof(a)), "mik_memcpy_d2h", v.ctx.handle)
    a
end
Base.getindex(v::HipVector, i::Int) = Array(v)[i]        # debugging / show only: one D2H copy per call
function Base.fill!(v::HipVector{T}, val) where {T}
    check(ccall((:mik_fill, libmik), Cint, (Ptr{Cvoid}, Cint, Int64, Ref{T}, Ptr{Cvoid}), v.ctx.handle, dtype_code(T), v.n, T(val), v.ptr), "mik_fill", v.ctx.handle)
    v
end
function Base.copyto!(y::HipVector{T}, x::HipVector{T}) where {T}
    check(ccall((:mik_copy, libmik), Cint, (Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Ptr{Cvoid}), y.ctx.handle, dtype_code(T), y.n, x.ptr, y.ptr), "mik_copy", y.ctx.handle)
    y
end
function LinearAlgebra.dot(x::HipVector{T}, y::HipVector{T}) where {T}
    out = Ref{T}()
    check(ccall((:mik_dot, libmik), Cint, (Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ref{T}), x.ctx.handle, dtype_code(T), x.n, x.ptr, y.ptr, out), "mik_dot", x.ctx.handle)
    out[]
end
function LinearAlgebra.norm(x::HipVector{T}) where {T}
    out = Ref{T}()
    check(ccall((:mik_nrm2, libmik), Cint, (Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Ref{T}), x.ctx.handle, dtype_code(T), x.n, x.ptr, out), "mik_nrm2", x.ctx.handle)
    out[]
end
function LinearAlgebra.axpy!(a, x::HipVector{T}, y::HipVector{T}) where {T}           # y .+= a .* x
    check(ccall((:mik_axpy, libmik), Cint, (Ptr{Cvoid}, Cint, Int64, Ref{T}, Ptr{Cvoid}, Ptr{Cvoid}), y.ctx.handle, dtype_code(T), y.n, T(a), x.ptr, y.ptr), "mik_axpy", y.ctx.handle)
    y
end
function xpby!(x::HipVector{T}, b, y::HipVector{T}) where {T}                          # y .= x .+ b .* y
    check(ccall((:mik_xpby, libmik), Cint, (Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Ref{T}, Ptr{Cvoid}), y.ctx.handle, dtype_code(T), y.n, x.ptr, T(b), y.ptr), "mik_xpby", y.ctx.handle)
    y
end
function LinearAlgebra.rmul!(x::HipVector{T}, a::Number) where {T}                    # x .*= a
    check(ccall((:mik_scal, libmik), Cint, (Ptr{Cvoid}, Cint, Int64, Ref{T}, Ptr{Cvoid}), x.ctx.handle, dtype_code(T), x.n, T(a), x.ptr), "mik_scal", x.ctx.handle)
    x
end

# ---- operator -----------------------------------------------------------------------------------
mutable struct HipCSR{T<:MikFloat}
    handle::Ptr{Cvoid}
    m::Int
    n::Int
    ctx::Context
end
"Upload a SparseMatrixCSC{T,Int} (colptr/rowval/nzval, 1-based Int64: test/laplace_matrix.jl:12)."
function HipCSR(A::SparseMatrixCSC{T, Int64}, ctx::Context = context()) where {T<:MikFloat}
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve A check(ccall((:mik_csr_create, libmik), Cint,
        (Ptr{Cvoid}, Cint, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Cvoid}, Cint, Cint, Ref{Ptr{Cvoid}}),
        ctx.handle, dtype_code(T), size(A, 1), size(A, 2), nnz(A), pointer(A.colptr), pointer(A.rowval), pointer(A.nzval), 1, 1, h),
        "mik_csr_create", ctx.handle)
    op = HipCSR{T}(h[], size(A, 1), size(A, 2), ctx)
    finalizer(o -> ccall((:mik_csr_destroy, libmik), Cint, (Ptr{Cvoid},), o.handle), op)
    op
end
Base.eltype(::HipCSR{T}) where {T} = T
Base.size(A::HipCSR) = (A.m, A.n)
Base.size(A::HipCSR, d::Integer) = d == 1 ? A.m : d == 2 ? A.n : 1
function LinearAlgebra.mul!(y::HipVector{T}, A::HipCSR{T}, x::HipVector{T}) where {T}
    check(ccall((:mik_spmv, libmik), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), A.ctx.handle, A.handle, x.ptr, y.ptr), "mik_spmv", A.ctx.handle)
    y
end
Base.:*(A::HipCSR{T}, x::HipVector{T}) where {T} = mul!(HipVector{T}(undef, A.m, A.ctx), A, x)

"Diagonal (Jacobi) left preconditioner on the device: ldiv!(y, P, x) = y .= x ./ P.diagonal (test/cg.jl:14-18)."
struct HipJacobi{T}
    diagonal::HipVector{T}
end

# ---- CGIterable ---------------------------------------------------------------------------------
mutable struct HipCGIterable{T, Tx<:HipVector{T}}
    handle::Ptr{Cvoid}
    A::HipCSR{T}
    x::Tx
    r::Tx; c::Tx; u::Tx; b::Tx       # keep the vectors alive while the handle uses their pointers
    Pl
    tol::T
    residual::T
    prev_residual::T
    maxiter::Int
    mv_products::Int
end

function refresh!(it::HipCGIterable{T}) where {T}
    res = Ref{Cdouble}(); prev = Ref{Cdouble}(); tol = Ref{Cdouble}(); mx = Ref{Int64}(); mv = Ref{Int64}(); cv = Ref{Cint}()
    check(ccall((:mik_cg_state, libmik), Cint, (Ptr{Cvoid}, Ref{Cdouble}, Ref{Cdouble}, Ref{Cdouble}, Ref{Int64}, Ref{Int64}, Ref{Cint}),
                it.handle, res, prev, tol, mx, mv, cv), "mik_cg_state", it.A.ctx.handle)
    it.residual = T(res[]); it.prev_residual = T(prev[]); it.tol = T(tol[]); it.mv_products = mv[]
    it
end

# cg_iterator!(x, A, b, Pl; ...)  -- src/cg.jl:120-155, specialised on the device types
function IterativeSolvers.cg_iterator!(x::HipVector{T}, A::HipCSR{T}, b::HipVector{T}, Pl = Identity();
        abstol::Real = zero(T), reltol::Real = sqrt(eps(T)), maxiter::Int = size(A, 2),
        statevars::IterativeSolvers.CGStateVariables = IterativeSolvers.CGStateVariables(zero(x), similar(x), similar(x)),
        initially_zero::Bool = false) where {T}
    Pl isa Identity || Pl isa HipJacobi || throw(MikError(Cint(5), "cg_iterator!", "Pl must be Identity() or HipJacobi on the device path"))
    diag = Pl isa HipJacobi ? Pl.diagonal.ptr : C_NULL
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mik_cg_create, libmik), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Int64, Cint, Ref{Ptr{Cvoid}}),
        A.ctx.handle, A.handle, x.ptr, b.ptr, statevars.u.ptr, statevars.r.ptr, statevars.c.ptr, diag,
        Float64(abstol), Float64(reltol), maxiter, initially_zero ? 1 : 0, h), "mik_cg_create", A.ctx.handle)
    it = HipCGIterable{T, typeof(x)}(h[], A, x, statevars.r, statevars.c, statevars.u, b, Pl, zero(T), zero(T), one(T), maxiter, 0)
    finalizer(i -> ccall((:mik_cg_destroy, libmik), Cint, (Ptr{Cvoid},), i.handle), it)
    refresh!(it)
end

IterativeSolvers.converged(it::HipCGIterable) = it.residual ≤ it.tol                      # src/cg.jl:32
IterativeSolvers.start(::HipCGIterable) = 0                                                # src/cg.jl:34
IterativeSolvers.done(it::HipCGIterable, iteration::Int) = iteration ≥ it.maxiter || IterativeSolvers.converged(it)   # :36

# iterate(it, iteration) -- src/cg.jl:43-66 (fused device step; one host-visible residual per call)
function Base.iterate(it::HipCGIterable{T}, iteration::Int = IterativeSolvers.start(it)) where {T}
    res = Ref{Cdouble}(); done = Ref{Cint}()
    check(ccall((:mik_cg_iterate, libmik), Cint, (Ptr{Cvoid}, Int64, Ref{Cdouble}, Ref{Cint}), it.handle, iteration, res, done), "mik_cg_iterate", it.A.ctx.handle)
    done[] != 0 && return nothing
    it.prev_residual = it.residual
    it.residual = T(res[])
    it.mv_products += 1
    it.residual, iteration + 1
end

# ---- GMRESIterable ------------------------------------------------------------------------------
orth_code(::ModifiedGramSchmidt) = Cint(0)
orth_code(::ClassicalGramSchmidt) = Cint(1)
orth_code(::DGKS) = Cint(2)

mutable struct HipResidual{T}      # stands in for g.residual.current read by the driver (src/gmres.jl:51)
    current::T
end
mutable struct HipGMRESIterable{T, Tx<:HipVector{T}}
    handle::Ptr{Cvoid}
    A::HipCSR{T}
    x::Tx
    b::Tx
    residual::HipResidual{T}
    mv_products::Int
    restart::Int
    k::Int
    maxiter::Int
    tol::T
    β::T
end

function refresh!(g::HipGMRESIterable{T}) where {T}
    res = Ref{Cdouble}(); tol = Ref{Cdouble}(); beta = Ref{Cdouble}(); k = Ref{Cint}(); mv = Ref{Int64}(); cv = Ref{Cint}()
    check(ccall((:mik_gmres_state, libmik), Cint, (Ptr{Cvoid}, Ref{Cdouble}, Ref{Cdouble}, Ref{Cdouble}, Ref{Cint}, Ref{Int64}, Ref{Cint}),
                g.handle, res, tol, beta, k, mv, cv), "mik_gmres_state", g.A.ctx.handle)
    g.residual.current = T(res[]); g.tol = T(tol[]); g.β = T(beta[]); g.k = k[]; g.mv_products = mv[]
    g
end

# gmres_iterable!(x, A, b; ...) -- src/gmres.jl:108-136
function IterativeSolvers.gmres_iterable!(x::HipVector{T}, A::HipCSR{T}, b::HipVector{T};
        Pl = Identity(), Pr = Identity(), abstol::Real = zero(T), reltol::Real = sqrt(eps(T)),
        restart::Int = min(20, size(A, 2)), maxiter::Int = size(A, 2), initially_zero::Bool = false,
        orth_meth::OrthogonalizationMethod = ModifiedGramSchmidt()) where {T}
    all(P -> P isa Identity || P isa HipJacobi, (Pl, Pr)) || throw(MikError(Cint(5), "gmres_iterable!", "Pl / Pr must be Identity() or HipJacobi on the device path"))
    pl = Pl isa HipJacobi ? Pl.diagonal.ptr : C_NULL
    pr = Pr isa HipJacobi ? Pr.diagonal.ptr : C_NULL
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mik_gmres_create, libmik), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Cint, Int64, Cint, Cint, Ref{Ptr{Cvoid}}),
        A.ctx.handle, A.handle, x.ptr, b.ptr, pl, pr, Float64(abstol), Float64(reltol), restart, maxiter, initially_zero ? 1 : 0, orth_code(orth_meth), h),
        "mik_gmres_create", A.ctx.handle)
    g = HipGMRESIterable{T, typeof(x)}(h[], A, x, b, HipResidual{T}(one(T)), 0, restart, 1, maxiter, zero(T), one(T))
    finalizer(i -> ccall((:mik_gmres_destroy, libmik), Cint, (Ptr{Cvoid},), i.handle), g)
    refresh!(g)
end

IterativeSolvers.converged(g::HipGMRESIterable) = g.residual.current ≤ g.tol                  # src/gmres.jl:51
IterativeSolvers.start(::HipGMRESIterable) = 0
IterativeSolvers.done(g::HipGMRESIterable, iteration::Int) = iteration ≥ g.maxiter || IterativeSolvers.converged(g)

# iterate(g, iteration) -- src/gmres.jl:57-106
function Base.iterate(g::HipGMRESIterable{T}, iteration::Int = IterativeSolvers.start(g)) where {T}
    res = Ref{Cdouble}(); done = Ref{Cint}()
    check(ccall((:mik_gmres_iterate, libmik), Cint, (Ptr{Cvoid}, Int64, Ref{Cdouble}, Ref{Cint}), g.handle, iteration, res, done), "mik_gmres_iterate", g.A.ctx.handle)
    done[] != 0 && return nothing
    refresh!(g)
    g.residual.current, iteration + 1
end

# zerox(A, b) (src/common.jl:18-23) already works: similar(b, T, size(A, 2)) + fill! are defined above.

# ------------------------------------------------------------------------------------------------
# gmres! over a row partition (one process per GPU): include/mik.h `mik_partition`
# ------------------------------------------------------------------------------------------------
# The handle calls back for the two couplings between ranks; everything else is the iterable above.
#   halo(user)::Cint                      -- send_buf is packed (stream-ordered); fill x_ext[n_loc+1:n_ext]
#   reduce(user, dtype, count, values)    -- partial sums in, ((p0 + p1) + p2) + ... in rank order out
struct Partition                     # same field order and types as the C struct
    rank::Cint
    nranks::Cint
    n_ext::Int64
    x_ext::Ptr{Cvoid}
    send_idx::Ptr{Int32}
    n_send::Int64
    send_buf::Ptr{Cvoid}
    halo::Ptr{Cvoid}                 # @cfunction(halo_cb, Cint, (Ptr{Cvoid},))
    reduce::Ptr{Cvoid}               # @cfunction(reduce_cb, Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}))
    user::Ptr{Cvoid}
end

"""
    gmres_iterable_partitioned!(x, A_loc, b, part; kwargs...)

`A_loc` is this rank's `n_loc x n_ext` block (halo columns behind the owned ones), `x`, `b` its rows.  Returns the
same `HipGMRESIterable`; drive it with `iterate` / `gmres!`-style loops on every rank in lockstep.
"""
function gmres_iterable_partitioned!(x::HipVector{T}, A::HipCSR{T}, b::HipVector{T}, part::Partition; Pl = Identity(), Pr = Identity(),
        abstol::Real = zero(real(T)), reltol::Real = sqrt(eps(real(T))), restart::Int = 20, maxiter::Int,
        initially_zero::Bool = false, orth_meth::OrthogonalizationMethod = ClassicalGramSchmidt()) where {T}
    pl = Pl isa HipJacobi ? Pl.diagonal.ptr : C_NULL
    pr = Pr isa HipJacobi ? Pr.diagonal.ptr : C_NULL
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mik_gmres_create_partitioned, libmik), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Cint, Int64, Cint, Cint, Ref{Partition}, Ref{Ptr{Cvoid}}),
        A.ctx.handle, A.handle, x.ptr, b.ptr, pl, pr, Float64(abstol), Float64(reltol), restart, maxiter, initially_zero ? 1 : 0,
        orth_code(orth_meth), Ref(part), h), "mik_gmres_create_partitioned", A.ctx.handle)
    g = HipGMRESIterable{T, typeof(x)}(h[], A, x, b, HipResidual{T}(one(T)), 0, restart, 1, maxiter, zero(T), one(T))
    finalizer(i -> ccall((:mik_gmres_destroy, libmik), Cint, (Ptr{Cvoid},), i.handle), g)
    refresh!(g)
end

end # module
