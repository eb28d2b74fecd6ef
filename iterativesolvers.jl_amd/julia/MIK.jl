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
#   bicgstabl!(x, A, b, l)   src/bicgstabl.jl:181 calls bicgstabl_iterator!(x, A, b, l; ...) at :197 -> HipBiCGStabIterable
#   minres!(x, A, b)         src/minres.jl:197 calls minres_iterable!(x, A, b; ...) at :208 -> HipMINRESIterable
#       (one C call per iteration each: mik_bicgstab_step, mik_minres_step)
#   lsqr! / lsmr! / qmr!     run UNMODIFIED on HipVector / HipCSR: adjoint(A) is the operator uploaded by MIK.with_adjoint (the CSC arrays read
#       as CSR), their vector statements lower through the broadcast style and axpy! / rmul! / dot / norm below (one L1 call per statement)
#   idrs!(x, A, b)           src/idrs.jl:49 -> idrs_method! (:150) calls idrs_iterable!(log, X, A, C, s, Pl, ...) at :156 -> HipIDRSIterable
#       (one C call per step: mik_idrs_step)
#   Generic code paths (any other solver of the package) see HipVector/HipCSR through
#   mul!, dot, norm, axpy!, rmul!, ldiv!, copyto!, fill!, similar, zero and the broadcast style below (the four
#   vector-update shapes of src/cg.jl), so the unmodified iterate(::CGIterable) runs on device vectors as well.
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
        # Julia gives no finalizer order at exit: the context marks itself dead (handle = C_NULL) and every child's
        # finalizer checks `alive(ctx)` first -- a child outliving its context leaks device memory of a dying process
        # instead of handing mik_free a context that no longer exists.
        finalizer(ctx) do c
            c.handle == C_NULL || ccall((:mik_ctx_destroy, libmik), Cint, (Ptr{Cvoid},), c.handle)
            c.handle = C_NULL
        end
        ctx
    end
end
alive(c::Context) = c.handle != C_NULL
struct DeviceInfo                    # same field order and types as the C struct mik_device_info
    device::Cint
    compute_units::Cint
    xcds::Cint
    wavefront_size::Cint
    lds_bytes_per_cu::Int64
    l2_bytes::Int64
    hbm_bytes::Int64
    arch::NTuple{64, UInt8}
    planned_compute_units::Cint
    planned_xcds::Cint
    xcd_maps::Cint
    resident_workgroup_cap::Cint
    gs_single_launch_max_segments::Cint
    gs_xcd_local_max_workgroups::Cint
    sweep_grid_cap::Cint
    mgs_resident_max_segments::Cint
    reserved::NTuple{7, Cint}
end
"The machine behind a context as the library queried it, and the launch caps it derived (mik_ctx_info)."
function device_info(ctx::Context = context())
    r = Ref{DeviceInfo}()
    check(ccall((:mik_ctx_info, libmik), Cint, (Ptr{Cvoid}, Ref{DeviceInfo}), ctx.handle, r), "mik_ctx_info", ctx.handle)
    r[]
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
        finalizer(x -> alive(x.ctx) && ccall((:mik_free, libmik), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), x.ctx.handle, x.ptr), v)
        v
    end
    # a view of device memory owned by somebody else (no finalizer): what the operator / preconditioner callbacks receive
    HipVector{T}(ptr::Ptr{Cvoid}, n::Integer, ctx::Context, ::Val{:unowned}) where {T<:MikFloat} = new{T}(ptr, n, ctx)
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
# Scalar indexing would copy the WHOLE vector device-to-host per element (134 MB at 256^3): any generic AbstractVector
# fallback that reaches it -- show, a missed broadcast, a generic dot -- must fail loudly instead of crawling.
const scalar_indexing_allowed = Ref(false)
"allowscalar(true) permits v[i] (one full device-to-host copy per call) for debugging at the REPL."
allowscalar(flag::Bool) = (scalar_indexing_allowed[] = flag)
function Base.getindex(v::HipVector, i::Int)
    scalar_indexing_allowed[] || error("scalar indexing of a HipVector is disabled (it copies the whole vector per element); ",
                                       "use Array(v), or MIK.allowscalar(true) at the REPL")
    Array(v)[i]
end
Base.setindex!(v::HipVector, x, i::Int) = error("setindex! on a HipVector is not supported; build a Vector and upload it with HipVector(a)")
Base.show(io::IO, v::HipVector{T}) where {T} = print(io, "HipVector{", T, "}(n = ", v.n, ") on device memory")
Base.show(io::IO, ::MIME"text/plain", v::HipVector) = show(io, v)
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

function sub!(x::HipVector{T}, y::HipVector{T}) where {T}                               # y .-= x
    check(ccall((:mik_sub, libmik), Cint, (Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Ptr{Cvoid}), y.ctx.handle, dtype_code(T), y.n, x.ptr, y.ptr), "mik_sub", y.ctx.handle)
    y
end

# ---- broadcast: SURVEY.md section 8b plug point (1) -----------------------------------------------------------------------
# The UNMODIFIED iterate(::CGIterable) writes its vector updates as fused broadcasts.  With this style they lower to
# the L1 entry points instead of falling into scalar indexing; the four shapes of src/cg.jl are
#     u .= r .+ beta .* u      (:51, :86)   -> mik_xpby
#     x .+= alpha .* u         (:58, :93)   -> mik_axpy(alpha)
#     r .-= alpha .* c         (:59, :94)   -> mik_axpy(-alpha)        [same rounded product, then a rounded subtract]
#     r .-= c                  (:138)       -> mik_sub
# and the scaled-sum shapes of src/lsqr.jl / src/lsmr.jl (`dest .= a .* v .+ w`, `dest .= v .+ w .* a`, `dest .= v .* a`, scale on either side)
#                                              -> mik_copy (when dest is not the scaled vector) + mik_xpby / mik_scal
# plus  x .= value (fill!) and  y .= x (copyto!).  Anything else throws -- by design: a shape that is not listed here
# would otherwise run as n scalar device-to-host copies.
struct HipStyle <: Base.Broadcast.AbstractArrayStyle{1} end
HipStyle(::Val{1}) = HipStyle()
HipStyle(::Val{N}) where {N} = Base.Broadcast.DefaultArrayStyle{N}()
Base.BroadcastStyle(::Type{<:HipVector}) = HipStyle()
const Bc = Base.Broadcast.Broadcasted

is_scaled(b) = b isa Bc && b.f === (*) && length(b.args) == 2 && b.args[1] isa Number && b.args[2] isa HipVector
# (scale, vector) of a term `v`, `a .* v` or `v .* a` (scale === nothing: unscaled); nothing for anything else.  The second form is how
# src/lsqr.jl:151,159 and src/lsmr.jl:161,167,199,201 write their updates; a rounded product is commutative, so both lower to the same call.
term(t::HipVector) = (nothing, t)
function term(t)
    (t isa Bc && t.f === (*) && length(t.args) == 2) || return nothing
    p, q = t.args
    p isa Number && q isa HipVector && return (p, q)
    q isa Number && p isa HipVector && return (q, p)
    nothing
end
function Base.copyto!(dest::HipVector{T}, bc::Bc{HipStyle}) where {T}
    f, a = bc.f, bc.args
    if f === identity && length(a) == 1
        return a[1] isa Number ? fill!(dest, a[1]) : copyto!(dest, a[1]::HipVector{T})
    elseif f === (+) && length(a) == 2 && is_scaled(a[2]) && (a[1] === dest || (a[2].args[2] === dest && a[1] isa HipVector))
        s, v = a[2].args
        a[1] === dest && return LinearAlgebra.axpy!(s, v, dest)                    # x .+= alpha .* u
        return xpby!(a[1], s, dest)                                                  # u .= r .+ beta .* u
    elseif f === (-) && length(a) == 2 && a[1] === dest
        is_scaled(a[2]) && return LinearAlgebra.axpy!(-a[2].args[1], a[2].args[2], dest)   # r .-= alpha .* c
        a[2] isa HipVector && return sub!(a[2], dest)                                # r .-= c
    elseif f === (+) && length(a) == 2 && term(a[1]) !== nothing && term(a[2]) !== nothing
        (s1, v1), (s2, v2) = term(a[1]), term(a[2])
        if s1 !== nothing && s2 === nothing                                          # dest .= s1 .* v1 .+ v2
            v2 === dest && v1 !== dest && return LinearAlgebra.axpy!(s1, v1, dest)
            v1 === dest || copyto!(dest, v1)                                         #   (u .= -alpha .* u .+ tmpm; w = t2 .* w .+ v; hbar .= hbar .* c .+ h)
            return xpby!(v2, s1, dest)
        elseif s1 === nothing && s2 !== nothing                                      # dest .= v1 .+ s2 .* v2   (u .= tmp_u .+ u .* -α)
            v2 === dest || (v1 === dest ? (return LinearAlgebra.axpy!(s2, v2, dest)) : copyto!(dest, v2))
            return xpby!(v1, s2, dest)
        elseif s1 === nothing && s2 === nothing && v1 === dest                       # x .+= (t1 * w)  with the product formed first (src/lsqr.jl:189)
            return LinearAlgebra.axpy!(one(T), v2, dest)
        end
    elseif f === (*) && length(a) == 2 && term(bc) !== nothing                       # u .*= inv(beta); wrho .= w .* inv(rho)
        s1, v1 = term(bc)
        v1 === dest || copyto!(dest, v1)
        return LinearAlgebra.rmul!(dest, s1)
    end
    error("broadcast shape not lowered for HipVector (MIK.jl lists the supported ones); expression: ", f, " over ", map(typeof, a))
end
Base.similar(bc::Bc{HipStyle}, ::Type{T}) where {T<:MikFloat} = HipVector{T}(undef, length(bc), first(x for x in Base.Broadcast.flatten(bc).args if x isa HipVector).ctx)

# ---- operator -----------------------------------------------------------------------------------
mutable struct HipCSR{T<:MikFloat}
    handle::Ptr{Cvoid}
    m::Int
    n::Int
    ctx::Context
    adj::Union{HipCSR{T}, Nothing}   # partner uploaded by with_adjoint: a reference cycle the GC collects as a whole (no global table keeps it alive)
    HipCSR{T}(handle, m, n, ctx) where {T<:MikFloat} = new{T}(handle, m, n, ctx, nothing)
end
"Upload a SparseMatrixCSC{T,Int} (colptr/rowval/nzval, 1-based Int64: test/laplace_matrix.jl:12)."
function HipCSR(A::SparseMatrixCSC{T, Int64}, ctx::Context = context()) where {T<:MikFloat}
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve A check(ccall((:mik_csr_create, libmik), Cint,
        (Ptr{Cvoid}, Cint, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Cvoid}, Cint, Cint, Ref{Ptr{Cvoid}}),
        ctx.handle, dtype_code(T), size(A, 1), size(A, 2), nnz(A), pointer(A.colptr), pointer(A.rowval), pointer(A.nzval), 1, 1, h),
        "mik_csr_create", ctx.handle)
    op = HipCSR{T}(h[], size(A, 1), size(A, 2), ctx)
    finalizer(o -> alive(o.ctx) && ccall((:mik_csr_destroy, libmik), Cint, (Ptr{Cvoid},), o.handle), op)
    op
end
"SparseMatrixCSC{T,Int32} -- the reference's tests run `Ti in (Int64, Int32)` (test/gmres.jl:38): 32-bit colptr / rowval."
function HipCSR(A::SparseMatrixCSC{T, Int32}, ctx::Context = context()) where {T<:MikFloat}
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve A check(ccall((:mik_csr_create_i32, libmik), Cint,
        (Ptr{Cvoid}, Cint, Int64, Int64, Int64, Ptr{Int32}, Ptr{Int32}, Ptr{Cvoid}, Cint, Cint, Ref{Ptr{Cvoid}}),
        ctx.handle, dtype_code(T), size(A, 1), size(A, 2), nnz(A), pointer(A.colptr), pointer(A.rowval), pointer(A.nzval), 1, 1, h),
        "mik_csr_create_i32", ctx.handle)
    op = HipCSR{T}(h[], size(A, 1), size(A, 2), ctx)
    finalizer(o -> alive(o.ctx) && ccall((:mik_csr_destroy, libmik), Cint, (Ptr{Cvoid},), o.handle), op)
    op
end
"""
The same upload from arrays that already live in DEVICE memory -- the colPtr / rowVal / nzVal buffers of an AMDGPU.jl
ROCSparseMatrixCSC, passed as raw device pointers: mik_csr_create detects the placement and starts its device-side pipeline
from them (no host copy).  The arrays are only read during the call.
"""
function HipCSR(::Type{T}, m::Integer, n::Integer, nz::Integer, colptr::Ptr{Int64}, rowval::Ptr{Int64}, nzval::Ptr{Cvoid},
                ctx::Context = context()) where {T<:MikFloat}
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mik_csr_create, libmik), Cint,
        (Ptr{Cvoid}, Cint, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Cvoid}, Cint, Cint, Ref{Ptr{Cvoid}}),
        ctx.handle, dtype_code(T), m, n, nz, colptr, rowval, nzval, 1, 1, h), "mik_csr_create", ctx.handle)
    op = HipCSR{T}(h[], m, n, ctx)
    finalizer(o -> alive(o.ctx) && ccall((:mik_csr_destroy, libmik), Cint, (Ptr{Cvoid},), o.handle), op)
    op
end
# adjoint(A) for lsqr! / lsmr! / qmr! (src/lsqr.jl:120, src/lsmr.jl:113, src/qmr.jl:51): the SAME colptr / rowval / nzval read as a CSR matrix
# are A' (row j of A' = column j of A, entries in storage order = the order mul!(y, adjoint(A), x) of SparseArrays sums them in), so the
# adjoint is a second upload with is_csc = 0 -- no transpose is formed for it.  `with_adjoint(A)` returns the operator; `adjoint(op)` / `op'`
# its partner.  (Real element types.)
function with_adjoint(A::SparseMatrixCSC{T, Int64}, ctx::Context = context()) where {T<:MikFloat}
    op = HipCSR(A, ctx)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve A check(ccall((:mik_csr_create, libmik), Cint,
        (Ptr{Cvoid}, Cint, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Cvoid}, Cint, Cint, Ref{Ptr{Cvoid}}),
        ctx.handle, dtype_code(T), size(A, 2), size(A, 1), nnz(A), pointer(A.colptr), pointer(A.rowval), pointer(A.nzval), 1, 0, h),
        "mik_csr_create", ctx.handle)
    adj = HipCSR{T}(h[], size(A, 2), size(A, 1), ctx)
    finalizer(o -> alive(o.ctx) && ccall((:mik_csr_destroy, libmik), Cint, (Ptr{Cvoid},), o.handle), adj)
    op.adj = adj; adj.adj = op
    op
end
function LinearAlgebra.adjoint(A::HipCSR)
    A.adj === nothing && throw(MikError(Cint(5), "adjoint", "this operator was uploaded without its adjoint: use MIK.with_adjoint(A)"))
    A.adj
end
Base.:*(a::Number, x::HipVector{T}) where {T} = LinearAlgebra.rmul!(copyto!(similar(x), x), T(a))       # t1*w (src/lsqr.jl:189)

"Release the CSR arrays of an operator that runs on one of the sliced layouts (mik_csr_compact); false if it needs them."
function compact!(A::HipCSR)
    rc = ccall((:mik_csr_compact, libmik), Cint, (Ptr{Cvoid},), A.handle)
    rc == 5 && return false
    check(rc, "mik_csr_compact", A.ctx.handle)
    true
end
"Name of the kernel mul!(y, A, x) launches for this operator (mik_spmv_kernel; for profiles)."
function spmv_kernel(A::HipCSR)
    buf = zeros(UInt8, 64)
    check(ccall((:mik_spmv_kernel, libmik), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint), A.handle, buf, 64), "mik_spmv_kernel", A.ctx.handle)
    unsafe_string(pointer(buf))
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
# ldiv!(y, P, x) / ldiv!(P, x): the contract of docs/src/preconditioning.md:5-14, so the GENERIC iterables (PCGIterable of
# src/cg.jl:72-100 through the broadcast style above, or any other solver of the package) accept it too
function LinearAlgebra.ldiv!(y::HipVector{T}, P::HipJacobi{T}, x::HipVector{T}) where {T}
    check(ccall((:mik_divide, libmik), Cint, (Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), y.ctx.handle, dtype_code(T), y.n, x.ptr, P.diagonal.ptr, y.ptr),
          "mik_divide", y.ctx.handle)
    y
end
LinearAlgebra.ldiv!(P::HipJacobi{T}, x::HipVector{T}) where {T} = LinearAlgebra.ldiv!(x, P, x)
Base.:\(P::HipJacobi{T}, x::HipVector{T}) where {T} = LinearAlgebra.ldiv!(similar(x), P, x)

# ---- CGIterable ---------------------------------------------------------------------------------
mutable struct HipCGIterable{T, Tx<:HipVector{T}}
    handle::Ptr{Cvoid}
    A                                # HipCSR or any operator with mul!
    ctx::Context
    keep::Vector{Any}                # callback boxes: must outlive the handle
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
                it.handle, res, prev, tol, mx, mv, cv), "mik_cg_state", it.ctx.handle)
    it.residual = T(res[]); it.prev_residual = T(prev[]); it.tol = T(tol[]); it.mv_products = mv[]
    it
end

# ---- any operator / any preconditioner: C callbacks (include/mik.h mik_mul_fn, mik_ldiv_fn) ------------------------------
# `user` carries a pointer to a Julia object that holds the operator and a template vector; the trampolines wrap the raw
# device pointers in non-owning HipVectors and call the ordinary mul! / ldiv! methods, so LinearMaps.jl maps over
# HipVectors, or any struct with mul! / ldiv! methods for HipVector, work unchanged (test/gmres.jl:59-66, :28-35).
struct MikOperator          # same field order and types as the C struct mik_operator
    dtype::Cint
    n::Int64
    csr::Ptr{Cvoid}
    mul::Ptr{Cvoid}
    user::Ptr{Cvoid}
end
struct MikPrecond           # mik_precond
    diag::Ptr{Cvoid}
    ldiv::Ptr{Cvoid}
    user::Ptr{Cvoid}
end
mutable struct CallbackBox{T}
    obj::Any
    n::Int
    ctx::Context
end
unowned(::Type{T}, p::Ptr{Cvoid}, n::Int, ctx::Context) where {T} = HipVector{T}(p, n, ctx, Val(:unowned))
function mul_trampoline(user::Ptr{Cvoid}, x::Ptr{Cvoid}, y::Ptr{Cvoid})::Cint
    box = unsafe_pointer_to_objref(user)::CallbackBox
    T = typeof(box).parameters[1]
    try
        LinearAlgebra.mul!(unowned(T, y, box.n, box.ctx), box.obj, unowned(T, x, box.n, box.ctx))
        return Cint(0)
    catch
        return Cint(1)          # surfaces as MIK_ERR_CALLBACK; never unwind through the C frame
    end
end
function ldiv_trampoline(user::Ptr{Cvoid}, y::Ptr{Cvoid}, x::Ptr{Cvoid})::Cint
    box = unsafe_pointer_to_objref(user)::CallbackBox
    T = typeof(box).parameters[1]
    try
        LinearAlgebra.ldiv!(unowned(T, y, box.n, box.ctx), box.obj, unowned(T, x, box.n, box.ctx))
        return Cint(0)
    catch
        return Cint(1)
    end
end
function operator_struct(A, ::Type{T}, n::Int, ctx::Context, keep::Vector{Any}) where {T}
    A isa HipCSR && return MikOperator(dtype_code(T), n, A.handle, C_NULL, C_NULL)
    box = CallbackBox{T}(A, n, ctx); push!(keep, box)
    MikOperator(dtype_code(T), n, C_NULL, @cfunction(mul_trampoline, Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid})), pointer_from_objref(box))
end
function precond_struct(P, ::Type{T}, n::Int, ctx::Context, keep::Vector{Any}) where {T}
    P isa Identity && return nothing
    P isa HipJacobi && return MikPrecond(P.diagonal.ptr, C_NULL, C_NULL)
    box = CallbackBox{T}(P, n, ctx); push!(keep, box)
    MikPrecond(C_NULL, @cfunction(ldiv_trampoline, Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid})), pointer_from_objref(box))
end
refptr(::Nothing) = C_NULL
refptr(r::Ref) = Base.unsafe_convert(Ptr{Cvoid}, r)

# cg_iterator!(x, A, b, Pl; ...)  -- src/cg.jl:120-155, specialised on the device vector type: A may be a HipCSR (fully fused
# step) or ANY operator with mul!(::HipVector, A, ::HipVector); Pl Identity(), HipJacobi (fused) or anything with ldiv!.
function IterativeSolvers.cg_iterator!(x::HipVector{T}, A, b::HipVector{T}, Pl = Identity();
        abstol::Real = zero(T), reltol::Real = sqrt(eps(T)), maxiter::Int = size(A, 2),
        statevars::IterativeSolvers.CGStateVariables = IterativeSolvers.CGStateVariables(zero(x), similar(x), similar(x)),
        initially_zero::Bool = false) where {T}
    ctx = x.ctx
    keep = Any[]
    h = Ref{Ptr{Cvoid}}(C_NULL)
    if A isa HipCSR && (Pl isa Identity || Pl isa HipJacobi)
        diag = Pl isa HipJacobi ? Pl.diagonal.ptr : C_NULL
        check(ccall((:mik_cg_create, libmik), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Int64, Cint, Ref{Ptr{Cvoid}}),
            ctx.handle, A.handle, x.ptr, b.ptr, statevars.u.ptr, statevars.r.ptr, statevars.c.ptr, diag,
            Float64(abstol), Float64(reltol), maxiter, initially_zero ? 1 : 0, h), "mik_cg_create", ctx.handle)
    else
        op = Ref(operator_struct(A, T, x.n, ctx, keep))
        pl = precond_struct(Pl, T, x.n, ctx, keep)
        plref = pl === nothing ? nothing : Ref(pl)
        GC.@preserve op plref keep check(ccall((:mik_cg_create_op, libmik), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Int64, Cint, Ref{Ptr{Cvoid}}),
            ctx.handle, refptr(op), refptr(plref), x.ptr, b.ptr, statevars.u.ptr, statevars.r.ptr, statevars.c.ptr,
            Float64(abstol), Float64(reltol), maxiter, initially_zero ? 1 : 0, h), "mik_cg_create_op", ctx.handle)
    end
    it = HipCGIterable{T, typeof(x)}(h[], A, ctx, keep, x, statevars.r, statevars.c, statevars.u, b, Pl, zero(T), zero(T), one(T), maxiter, 0)
    finalizer(i -> alive(i.ctx) && ccall((:mik_cg_destroy, libmik), Cint, (Ptr{Cvoid},), i.handle), it)
    refresh!(it)
end

IterativeSolvers.converged(it::HipCGIterable) = it.residual ≤ it.tol                      # src/cg.jl:32
IterativeSolvers.start(::HipCGIterable) = 0                                                # src/cg.jl:34
IterativeSolvers.done(it::HipCGIterable, iteration::Int) = iteration ≥ it.maxiter || IterativeSolvers.converged(it)   # :36

# iterate(it, iteration) -- src/cg.jl:43-66 (fused device step; one host-visible residual per call)
function Base.iterate(it::HipCGIterable{T}, iteration::Int = IterativeSolvers.start(it)) where {T}
    res = Ref{Cdouble}(); done = Ref{Cint}()
    check(ccall((:mik_cg_iterate, libmik), Cint, (Ptr{Cvoid}, Int64, Ref{Cdouble}, Ref{Cint}), it.handle, iteration, res, done), "mik_cg_iterate", it.ctx.handle)
    done[] != 0 && return nothing
    it.prev_residual = it.residual
    it.residual = T(res[])
    it.mv_products += 1
    it.residual, iteration + 1
end

"""
    iterate_many!(it, iteration, max_steps) -> Vector{Float64}

Up to `max_steps` `iterate` calls with ONE host synchronisation (`mik_cg_iterate_many`): the stopping test of src/cg.jl:36
runs on the device after every step.  Returns the residuals of the executed steps.
"""
function iterate_many!(it::HipCGIterable{T}, iteration::Int, max_steps::Int) where {T}
    res = Vector{Cdouble}(undef, max(max_steps, 1)); nd = Ref{Int64}(0)
    check(ccall((:mik_cg_iterate_many, libmik), Cint, (Ptr{Cvoid}, Int64, Int64, Ptr{Cdouble}, Ref{Int64}), it.handle, iteration, max_steps, res, nd),
          "mik_cg_iterate_many", it.ctx.handle)
    refresh!(it)
    resize!(res, nd[])
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
    A
    ctx::Context
    keep::Vector{Any}
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
                g.handle, res, tol, beta, k, mv, cv), "mik_gmres_state", g.ctx.handle)
    g.residual.current = T(res[]); g.tol = T(tol[]); g.β = T(beta[]); g.k = k[]; g.mv_products = mv[]
    g
end

# gmres_iterable!(x, A, b; ...) -- src/gmres.jl:108-136; A: HipCSR or any operator with mul!, Pl / Pr: Identity(), HipJacobi or
# anything with ldiv! (all three expand! methods of src/gmres.jl:285-304 run on the device side of the handle)
function IterativeSolvers.gmres_iterable!(x::HipVector{T}, A, b::HipVector{T};
        Pl = Identity(), Pr = Identity(), abstol::Real = zero(T), reltol::Real = sqrt(eps(T)),
        restart::Int = min(20, size(A, 2)), maxiter::Int = size(A, 2), initially_zero::Bool = false,
        orth_meth::OrthogonalizationMethod = ModifiedGramSchmidt()) where {T}
    ctx = x.ctx
    keep = Any[]
    simple(P) = P isa Identity || P isa HipJacobi
    h = Ref{Ptr{Cvoid}}(C_NULL)
    if A isa HipCSR && simple(Pl) && simple(Pr)
        pl = Pl isa HipJacobi ? Pl.diagonal.ptr : C_NULL
        pr = Pr isa HipJacobi ? Pr.diagonal.ptr : C_NULL
        check(ccall((:mik_gmres_create, libmik), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Cint, Int64, Cint, Cint, Ref{Ptr{Cvoid}}),
            ctx.handle, A.handle, x.ptr, b.ptr, pl, pr, Float64(abstol), Float64(reltol), restart, maxiter, initially_zero ? 1 : 0, orth_code(orth_meth), h),
            "mik_gmres_create", ctx.handle)
    else
        op = Ref(operator_struct(A, T, x.n, ctx, keep))
        pl = precond_struct(Pl, T, x.n, ctx, keep); plref = pl === nothing ? nothing : Ref(pl)
        pr = precond_struct(Pr, T, x.n, ctx, keep); prref = pr === nothing ? nothing : Ref(pr)
        GC.@preserve op plref prref keep check(ccall((:mik_gmres_create_op, libmik), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Cint, Int64, Cint, Cint, Ref{Ptr{Cvoid}}),
            ctx.handle, refptr(op), refptr(plref), refptr(prref), x.ptr, b.ptr, Float64(abstol), Float64(reltol), restart, maxiter,
            initially_zero ? 1 : 0, orth_code(orth_meth), h), "mik_gmres_create_op", ctx.handle)
    end
    g = HipGMRESIterable{T, typeof(x)}(h[], A, ctx, keep, x, b, HipResidual{T}(one(T)), 0, restart, 1, maxiter, zero(T), one(T))
    finalizer(i -> alive(i.ctx) && ccall((:mik_gmres_destroy, libmik), Cint, (Ptr{Cvoid},), i.handle), g)
    refresh!(g)
end

IterativeSolvers.converged(g::HipGMRESIterable) = g.residual.current ≤ g.tol                  # src/gmres.jl:51
IterativeSolvers.start(::HipGMRESIterable) = 0
IterativeSolvers.done(g::HipGMRESIterable, iteration::Int) = iteration ≥ g.maxiter || IterativeSolvers.converged(g)

# iterate(g, iteration) -- src/gmres.jl:57-106
function Base.iterate(g::HipGMRESIterable{T}, iteration::Int = IterativeSolvers.start(g)) where {T}
    res = Ref{Cdouble}(); done = Ref{Cint}()
    check(ccall((:mik_gmres_iterate, libmik), Cint, (Ptr{Cvoid}, Int64, Ref{Cdouble}, Ref{Cint}), g.handle, iteration, res, done), "mik_gmres_iterate", g.ctx.handle)
    done[] != 0 && return nothing
    refresh!(g)
    g.residual.current, iteration + 1
end

# ---- BiCGStabIterable ---------------------------------------------------------------------------
# bicgstabl!(x, A, b, l; ...) (src/bicgstabl.jl:181-219) calls bicgstabl_iterator!(x, A, b, l; ...) at :197; the reference's
# iterable keeps rs / us as host Matrices (:39-40), so the device path returns its own iterable with the fields the driver
# reads (`mv_products`, `residual`, `x`) and one C call per outer iteration: rho, beta, sigma, alpha, M, gamma, omega stay
# on the device (include/mik.h, mik_bicgstab_step).  `r_shadow` replaces rand(T, n) (:38) when reproducibility is wanted.
mutable struct HipBiCGStabIterable{T, Tx<:HipVector{T}}
    handle::Ptr{Cvoid}
    A::HipCSR{T}
    l::Int
    x::Tx
    r_shadow::Tx
    rs::Tx                 # n x (l + 1), column-major, leading dimension ld
    us::Tx
    ld::Int
    max_mv_products::Int
    mv_products::Int
    tol::T
    residual::T
    Pl
end
# column j (1-based) of a device block: a view, no finalizer
column(block::HipVector{T}, ld::Int, n::Int, j::Int) where {T} = HipVector{T}(block.ptr + (j - 1) * ld * sizeof(T), n, block.ctx, Val(:unowned))

function IterativeSolvers.bicgstabl_iterator!(x::HipVector{T}, A::HipCSR{T}, b::HipVector{T}, l::Int = 2;
        Pl = Identity(), max_mv_products = size(A, 2), abstol::Real = zero(T), reltol::Real = sqrt(eps(T)), initial_zero = false,
        r_shadow::HipVector{T} = HipVector(rand(T, x.n), x.ctx)) where {T}
    (Pl isa Identity || Pl isa HipJacobi) || throw(MikError(Cint(5), "bicgstabl_iterator!", "Pl must be Identity() or HipJacobi on the device path"))
    1 <= l <= 4 || throw(MikError(Cint(5), "bicgstabl_iterator!", "l must be 1 ... 4 on the device path"))
    n = x.n
    ld = cld(n, 64) * 64
    rs = HipVector{T}(undef, ld * (l + 1), x.ctx)                            # :39
    us = fill!(HipVector{T}(undef, ld * (l + 1), x.ctx), zero(T))            # :40
    residual = column(rs, ld, n, 1)
    mv_products = 0
    if initial_zero
        copyto!(residual, b)                                                 # :46
    else
        mul!(residual, A, x)                                                 # :48
        xpby!(b, -one(T), residual)                                          # residual .= b .- residual  :49
        mv_products += 1
    end
    Pl isa HipJacobi && ldiv!(residual, Pl, residual)                        # :55
    nrm = norm(residual)                                                     # :61
    tolerance = max(T(reltol) * nrm, T(abstol))                              # :69
    h = Ref{Ptr{Cvoid}}(C_NULL)
    pl = Pl isa HipJacobi ? Pl.diagonal.ptr : C_NULL
    check(ccall((:mik_bicgstab_create, libmik), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}),
        x.ctx.handle, A.handle, l, x.ptr, rs.ptr, ld, us.ptr, ld, r_shadow.ptr, pl, h), "mik_bicgstab_create", x.ctx.handle)
    it = HipBiCGStabIterable{T, typeof(x)}(h[], A, l, x, r_shadow, rs, us, ld, Int(max_mv_products), mv_products, tolerance, nrm, Pl)
    finalizer(i -> alive(i.x.ctx) && ccall((:mik_bicgstab_destroy, libmik), Cint, (Ptr{Cvoid},), i.handle), it)
    it
end

IterativeSolvers.converged(it::HipBiCGStabIterable) = it.residual ≤ it.tol                    # src/bicgstabl.jl:75
IterativeSolvers.start(::HipBiCGStabIterable) = 0
IterativeSolvers.done(it::HipBiCGStabIterable, iteration::Int) = it.mv_products ≥ it.max_mv_products || IterativeSolvers.converged(it)   # :77

# iterate(it, iteration) -- src/bicgstabl.jl:79-134, one call
function Base.iterate(it::HipBiCGStabIterable{T}, iteration::Int = IterativeSolvers.start(it)) where {T}
    IterativeSolvers.done(it, iteration) && return nothing
    res = Ref{T}()
    code = ccall((:mik_bicgstab_step, libmik), Cint, (Ptr{Cvoid}, Ref{T}), it.handle, res)
    code == 8 && throw(LinearAlgebra.SingularException(0))                   # MIK_ERR_SINGULAR: lu! at :124
    check(code, "mik_bicgstab_step", it.x.ctx.handle)
    it.mv_products += 2 * it.l                                               # :115
    it.residual = res[]                                                      # :132
    it.residual, iteration + 1
end

# ---- MINRESIterable -----------------------------------------------------------------------------
# minres!(x, A, b; ...) (src/minres.jl:197-230) calls minres_iterable!(x, A, b; ...) at :208; MINRESIterable wants DenseVector
# work vectors (:6), so the device path returns its own iterable with the fields the driver reads (`mv_products`, `x`) and one
# C call per iteration (include/mik.h, mik_minres_step): H, rhs and the rotations stay on the device.
mutable struct HipMINRESIterable{T, Tx<:HipVector{T}}
    handle::Ptr{Cvoid}
    A::HipCSR{T}
    skew_hermitian::Bool
    x::Tx
    v_prev::Tx
    v_curr::Tx
    v_next::Tx
    w_prev::Tx
    w_curr::Tx
    w_next::Tx
    mv_products::Int
    maxiter::Int
    tol::T
    resnorm::T
end

function IterativeSolvers.minres_iterable!(x::HipVector{T}, A::HipCSR{T}, b::HipVector{T};
        initially_zero::Bool = false, skew_hermitian::Bool = false, abstol::Real = zero(T), reltol::Real = sqrt(eps(T)), maxiter = size(A, 2)) where {T}
    v_prev = similar(x)
    v_curr = copyto!(similar(x), b)                                          # :47-48
    v_next = similar(x)
    w_prev = zero(x); w_curr = zero(x); w_next = zero(x)                     # :51-53 (similar there; zeros here: read only once written)
    mv_products = 0
    if !initially_zero
        mul!(v_next, A, x)                                                   # :60
        axpy!(-one(T), v_next, v_curr)                                       # :61
        mv_products = 1
    end
    resnorm = norm(v_curr)                                                   # :65
    tolerance = max(T(reltol) * resnorm, T(abstol))                          # :66
    rmul!(v_curr, inv(resnorm))                                              # :74
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mik_minres_create, libmik), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cint, Ref{Ptr{Cvoid}}),
        x.ctx.handle, A.handle, x.ptr, v_prev.ptr, v_curr.ptr, v_next.ptr, w_prev.ptr, w_curr.ptr, w_next.ptr, Float64(resnorm), skew_hermitian ? 1 : 0, h),
        "mik_minres_create", x.ctx.handle)
    m = HipMINRESIterable{T, typeof(x)}(h[], A, skew_hermitian, x, v_prev, v_curr, v_next, w_prev, w_curr, w_next, mv_products, Int(maxiter), tolerance, resnorm)
    finalizer(i -> alive(i.x.ctx) && ccall((:mik_minres_destroy, libmik), Cint, (Ptr{Cvoid},), i.handle), m)
    m
end

IterativeSolvers.converged(m::HipMINRESIterable) = m.resnorm ≤ m.tol                          # src/minres.jl:89
IterativeSolvers.start(::HipMINRESIterable) = 1                                                # :91
IterativeSolvers.done(m::HipMINRESIterable, iteration::Int) = iteration > m.maxiter || IterativeSolvers.converged(m)   # :93

# iterate(m, iteration) -- src/minres.jl:95-159, one call
function Base.iterate(m::HipMINRESIterable{T}, iteration::Int = IterativeSolvers.start(m)) where {T}
    IterativeSolvers.done(m, iteration) && return nothing
    res = Ref{T}()
    check(ccall((:mik_minres_step, libmik), Cint, (Ptr{Cvoid}, Int64, Ref{T}), m.handle, iteration, res), "mik_minres_step", m.x.ctx.handle)
    m.v_prev, m.v_curr, m.v_next = m.v_curr, m.v_next, m.v_prev              # :145 (the handle rotates its pointers alike)
    m.w_prev, m.w_curr, m.w_next = m.w_curr, m.w_next, m.w_prev              # :146
    m.resnorm = res[]                                                        # :154
    m.mv_products += 1
    m.resnorm, iteration + 1
end

# ---- IDRSIterable -------------------------------------------------------------------------------
# idrs!(x, A, b; ...) (src/idrs.jl:49-64) calls idrs_method! (:150-162), which calls idrs_iterable!(log, X, A, C, s, Pl, abstol, reltol,
# maxiter; smoothing, verbose) at :156, reduces over it and returns `iterable.X`.  The reference's IDRSIterable keeps P, U, G as Vectors of
# host vectors and its broadcasts would run statement by statement; the device path returns its own iterable with the fields the driver
# reads (`X`, `normR`) and ONE C call per step (include/mik.h, mik_idrs_step): M, f and omega live in the handle.  `P` (a device n x s
# block, leading dimension cld(n, 64) * 64) replaces rand! (:136) when reproducibility is wanted.
mutable struct HipIDRSIterable{T, Tx<:HipVector{T}}
    handle::Ptr{Cvoid}
    A::HipCSR{T}
    s::Int
    X::Tx
    R::Tx
    X_s::Union{Tx, Nothing}
    R_s::Union{Tx, Nothing}
    P::Tx                  # n x s, column-major, leading dimension ld
    U::Tx
    G::Tx
    ld::Int
    maxiter::Int
    smoothing::Bool
    verbose::Bool
    tol::T
    normR::T
    log
    Pl
end

function IterativeSolvers.idrs_iterable!(log, X::HipVector{T}, A::HipCSR{T}, C::HipVector{T}, s::Number, Pl, abstol::Real, reltol::Real,
        maxiter::Number; smoothing::Bool = false, verbose::Bool = false, P::Union{HipVector{T}, Nothing} = nothing) where {T}
    (Pl isa Identity || Pl isa HipJacobi) || throw(MikError(Cint(5), "idrs_iterable!", "Pl must be Identity() or HipJacobi on the device path"))
    1 <= s <= 32 || throw(MikError(Cint(5), "idrs_iterable!", "s must be 1 ... 32 on the device path"))
    n = X.n
    ld = cld(n, 64) * 64
    R = similar(X)
    mul!(R, A, X)                                                            # R = C - A*X  :119
    xpby!(C, -one(T), R)
    normR = norm(R)                                                          # :120
    tolerance = max(T(reltol) * normR, T(abstol))                            # :121
    X_s = smoothing ? copyto!(similar(X), X) : nothing                       # :123-126
    R_s = smoothing ? copyto!(similar(X), R) : nothing
    if P === nothing                                                         # :136
        Ph = zeros(T, ld * Int(s))
        for k in 1:Int(s)
            Ph[(k - 1) * ld + 1:(k - 1) * ld + n] = rand(T, n)
        end
        P = HipVector(Ph, X.ctx)
    end
    (P.n >= ld * Int(s) && P.ctx === X.ctx) || throw(MikError(Cint(3), "idrs_iterable!", "P must hold ld * s = $(ld * Int(s)) entries (column-major, leading dimension $ld) on X's context"))
    U = fill!(HipVector{T}(undef, ld * Int(s), X.ctx), zero(T))              # :137
    G = fill!(HipVector{T}(undef, ld * Int(s), X.ctx), zero(T))              # :138
    h = Ref{Ptr{Cvoid}}(C_NULL)
    pl = Pl isa HipJacobi ? Pl.diagonal.ptr : C_NULL
    check(ccall((:mik_idrs_create, libmik), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
         Cdouble, Ref{Ptr{Cvoid}}),
        X.ctx.handle, A.handle, s, X.ptr, R.ptr, P.ptr, ld, U.ptr, ld, G.ptr, ld, pl, smoothing ? X_s.ptr : C_NULL, smoothing ? R_s.ptr : C_NULL,
        Float64(normR), h), "mik_idrs_create", X.ctx.handle)
    it = HipIDRSIterable{T, typeof(X)}(h[], A, Int(s), X, R, X_s, R_s, P, U, G, ld, Int(maxiter), smoothing, verbose, tolerance, normR, log, Pl)
    finalizer(i -> alive(i.X.ctx) && ccall((:mik_idrs_destroy, libmik), Cint, (Ptr{Cvoid},), i.handle), it)
    it
end

# iterate(it, (iter, step)) -- src/idrs.jl:164-272, one call per step
function Base.iterate(it::HipIDRSIterable{T}, (iter, step) = (1, 1)) where {T}
    if it.normR < it.tol || iter > it.maxiter                                # :168
        it.log !== nothing && IterativeSolvers.setconv(it.log, 0 <= it.normR < it.tol)
        it.smoothing && copyto!(it.X, it.X_s)                                # :171-173
        return nothing
    end
    res = Ref{T}()
    check(ccall((:mik_idrs_step, libmik), Cint, (Ptr{Cvoid}, Cint, Ref{T}), it.handle, step, res), "mik_idrs_step", it.X.ctx.handle)
    it.normR = res[]
    nextstep = step <= it.s ? step + 1 : 1                                   # :240, :267
    if it.log !== nothing
        IterativeSolvers.nextiter!(it.log, mvps = 1)                         # :268-269
        push!(it.log, :resnorm, it.normR)
    end
    it.normR, (iter + 1, nextstep)
end

# zerox(A, b) (src/common.jl:18-23) already works: similar(b, T, size(A, 2)) + fill! are defined above.

# ------------------------------------------------------------------------------------------------
# gmres! over a row partition (one process per GPU): include/mik.h `mik_partition`
# ------------------------------------------------------------------------------------------------
# The handle calls back for the two couplings between ranks -- or, with `link` set, exchanges by itself on the device (Transport 3: no
# host round trip inside an Arnoldi column); everything else is the iterable above.
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
    link::Ptr{Cvoid}                 # C_NULL: the callbacks couple the ranks; a connected PartitionLink's handle: the library does, on the device
end

"""
    gmres_iterable_partitioned!(x, A_loc, b, part; n_global, kwargs...)

`A_loc` is this rank's `n_loc x n_ext` block (halo columns behind the owned ones), `x`, `b` its rows.  Returns the same
`HipGMRESIterable`; drive it with `iterate` / `gmres!`-style loops on every rank in lockstep.  Defaults are those of
`gmres_iterable!` (src/gmres.jl:108-117) evaluated on the GLOBAL size: `restart = min(20, n_global)`,
`maxiter = n_global`, `orth_meth = ModifiedGramSchmidt()` -- switching a solve to the partitioned call does not change
its result.  (ClassicalGramSchmidt() needs one reduction per step instead of k: pass it explicitly at scale.)
"""
function gmres_iterable_partitioned!(x::HipVector{T}, A::HipCSR{T}, b::HipVector{T}, part::Partition; n_global::Int, Pl = Identity(), Pr = Identity(),
        abstol::Real = zero(real(T)), reltol::Real = sqrt(eps(real(T))), restart::Int = min(20, n_global), maxiter::Int = n_global,
        initially_zero::Bool = false, orth_meth::OrthogonalizationMethod = ModifiedGramSchmidt()) where {T}
    all(P -> P isa Identity || P isa HipJacobi, (Pl, Pr)) || throw(MikError(Cint(5), "gmres_iterable_partitioned!", "Pl / Pr must be Identity() or HipJacobi"))
    pl = Pl isa HipJacobi ? Pl.diagonal.ptr : C_NULL
    pr = Pr isa HipJacobi ? Pr.diagonal.ptr : C_NULL
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mik_gmres_create_partitioned, libmik), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Cint, Int64, Cint, Cint, Ref{Partition}, Ref{Ptr{Cvoid}}),
        A.ctx.handle, A.handle, x.ptr, b.ptr, pl, pr, Float64(abstol), Float64(reltol), restart, maxiter, initially_zero ? 1 : 0,
        orth_code(orth_meth), Ref(part), h), "mik_gmres_create_partitioned", A.ctx.handle)
    g = HipGMRESIterable{T, typeof(x)}(h[], A, A.ctx, Any[], x, b, HipResidual{T}(one(T)), 0, restart, 1, maxiter, zero(T), one(T))
    finalizer(i -> alive(i.ctx) && ccall((:mik_gmres_destroy, libmik), Cint, (Ptr{Cvoid},), i.handle), g)
    refresh!(g)
end

# ------------------------------------------------------------------------------------------------
# cg! over a row partition, exchanges inside libmik.so (include/mik.h "Transport 1": RCCL over xGMI, one process per GPU)
# ------------------------------------------------------------------------------------------------
"RCCL communicator owned by libmik.so.  `id` = the 128 bytes of `unique_id()` made on rank 0 and broadcast by the host (MPI.jl)."
mutable struct Comm
    handle::Ptr{Cvoid}
    ctx::Context
end
function unique_id()
    id = zeros(UInt8, 128)
    check(ccall((:mik_comm_unique_id, libmik), Cint, (Ptr{UInt8},), id), "mik_comm_unique_id")
    id
end
function Comm(ctx::Context, id::Union{Nothing, Vector{UInt8}}, rank::Integer, nranks::Integer)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve id check(ccall((:mik_comm_create, libmik), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint, Ref{Ptr{Cvoid}}),
                                ctx.handle, id === nothing ? C_NULL : pointer(id), rank, nranks, h), "mik_comm_create", ctx.handle)
    c = Comm(h[], ctx)
    finalizer(x -> alive(x.ctx) && ccall((:mik_comm_destroy, libmik), Cint, (Ptr{Cvoid},), x.handle), c)
    c
end

# Transport 3 (include/mik.h): peer-mapped mailboxes -- no collective launch in the step.  `allgather` is whatever the host has for
# small byte vectors (MPI.Allgather!): it carries the 64-byte HIP IPC handles once.
"64-byte HIP IPC handle of this communicator's mailbox (mik_comm_mailbox_export)"
function mailbox_handle(c::Comm)
    h = zeros(UInt8, 64)
    check(ccall((:mik_comm_mailbox_export, libmik), Cint, (Ptr{Cvoid}, Ptr{UInt8}), c.handle, h), "mik_comm_mailbox_export", c.ctx.handle)
    h
end
"every rank's mailbox handle, rank order, 64 bytes each (mik_comm_mailbox_connect; collective)"
function connect_mailboxes!(c::Comm, handles::Vector{UInt8})
    GC.@preserve handles check(ccall((:mik_comm_mailbox_connect, libmik), Cint, (Ptr{Cvoid}, Ptr{UInt8}), c.handle, handles), "mik_comm_mailbox_connect", c.ctx.handle)
    c
end
"""
The device-driven links of one row partition on a communicator whose mailboxes are connected (include/mik.h `mik_plink`): halo plan,
landing buffer, peer mappings.  `recv` / `send`: (peer, offset, count) segments of the ghost region / of the packed send buffer.
`allgather(x)` is whatever the host has for small host objects (MPI.Allgather of the 64-byte handle, the ghost count and the receive
segments of every rank); `PartitionLink(...)` is collective.  Its `handle` goes into `Partition.link`.
"""
mutable struct PartitionLink
    handle::Ptr{Cvoid}
    comm::Comm
end
function PartitionLink(comm::Comm, ::Type{T}, n_ghost::Integer, recv::Vector{NTuple{3, Int}}, send::Vector{NTuple{3, Int}}, rank::Integer, allgather) where {T<:MikFloat}
    rp = Cint[p for (p, _, _) in recv]; ro = Int64[o for (_, o, _) in recv]; rc = Int64[k for (_, _, k) in recv]
    sp = Cint[p for (p, _, _) in send]; so = Int64[o for (_, o, _) in send]; sc = Int64[k for (_, _, k) in send]
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mik_plink_create, libmik), Cint, (Ptr{Cvoid}, Cint, Int64, Cint, Ptr{Cint}, Ptr{Int64}, Ptr{Int64}, Cint, Ptr{Cint}, Ptr{Int64}, Ptr{Int64}, Ref{Ptr{Cvoid}}),
                comm.handle, dtype_code(T), n_ghost, length(rp), rp, ro, rc, length(sp), sp, so, sc, h), "mik_plink_create", comm.ctx.handle)
    mine = zeros(UInt8, 64)
    check(ccall((:mik_plink_export, libmik), Cint, (Ptr{Cvoid}, Ptr{UInt8}), h[], mine), "mik_plink_export", comm.ctx.handle)
    info = allgather((mine, Int64(n_ghost), recv))                       # per rank: (handle, ghost count, receive segments)
    handles = reduce(vcat, [i[1] for i in info]); counts = Int64[i[2] for i in info]
    dst = landing_targets(send, info, rank)
    GC.@preserve handles counts dst check(ccall((:mik_plink_connect, libmik), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Ptr{Int64}, Ptr{Int64}), h[], handles, counts, dst),
                                          "mik_plink_connect", comm.ctx.handle)
    l = PartitionLink(h[], comm)
    finalizer(x -> alive(x.comm.ctx) && ccall((:mik_plink_destroy, libmik), Cint, (Ptr{Cvoid},), x.handle), l)
    l
end
"per SEND segment: the offset of the matching receive segment in the receiver's ghost region (info[q][3] = rank q's receive segments)"
function landing_targets(send, info, rank)
    taken = Dict{Int, Int}(); dst = Int64[]
    for (peer, _, cnt) in send
        cands = [sg for sg in info[peer + 1][3] if sg[1] == rank]
        k = get(taken, peer, 0) + 1; taken[peer] = k
        (k <= length(cands) && cands[k][3] == cnt) || throw(MikError(Cint(3), "landing_targets", "halo plans of ranks $rank and $peer disagree"))
        push!(dst, cands[k][2])
    end
    isempty(dst) ? Int64[0] : dst
end

"Row-partitioned CGIterable: this rank's block, the halo plan and the communicator; `iterate_many!` is ONE ccall per batch."
mutable struct HipDistCG{T}
    handle::Ptr{Cvoid}
    ctx::Context
    keep::Vector{Any}
    residual::Float64
    tol::Float64
    maxiter::Int
end
function dist_cg_iterator!(x::HipVector{T}, A_loc::HipCSR{T}, b::HipVector{T}, comm::Comm, rank::Integer, nranks::Integer;
        send_idx::Vector{Int32}, recv::Vector{NTuple{3, Int}}, send::Vector{NTuple{3, Int}},        # (peer, offset, count)
        abstol::Real = 0.0, reltol::Real = sqrt(eps(T)), maxiter::Int, initially_zero::Bool = true,
        u_ext::Union{Nothing, HipVector{T}} = nothing, ghosts = nothing) where {T}
    ctx = x.ctx
    n_loc, n_ext = size(A_loc)
    u_ext = u_ext === nothing ? fill!(HipVector{T}(undef, n_ext, ctx), 0) : u_ext
    r = similar(x); c = similar(x)
    sbuf = HipVector{T}(undef, max(length(send_idx), 1), ctx)
    dot_all = fill!(HipVector{T}(undef, nranks, ctx), 0); rr_all = fill!(HipVector{T}(undef, nranks, ctx), 0)
    sidx = HipVector{Float32}(undef, max(length(send_idx), 1), ctx)      # 4-byte slots: holds the Int32 indices
    isempty(send_idx) || GC.@preserve send_idx check(ccall((:mik_memcpy_h2d, libmik), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t),
                                                           ctx.handle, sidx.ptr, pointer(send_idx), sizeof(send_idx)), "mik_memcpy_h2d", ctx.handle)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mik_cgd_create, libmik), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint,
         Cdouble, Cdouble, Int64, Cint, Ref{Ptr{Cvoid}}),
        ctx.handle, A_loc.handle, x.ptr, b.ptr, u_ext.ptr, r.ptr, c.ptr, sidx.ptr, length(send_idx), sbuf.ptr, dot_all.ptr, rr_all.ptr, rank, nranks,
        Float64(abstol), Float64(reltol), maxiter, initially_zero ? 1 : 0, h), "mik_cgd_create", ctx.handle)
    rp = Cint[p for (p, _, _) in recv]; ro = Int64[o for (_, o, _) in recv]; rc = Int64[k for (_, _, k) in recv]
    sp = Cint[p for (p, _, _) in send]; so = Int64[o for (_, o, _) in send]; sc = Int64[k for (_, _, k) in send]
    check(ccall((:mik_cgd_set_halo_plan, libmik), Cint, (Ptr{Cvoid}, Cint, Ptr{Cint}, Ptr{Int64}, Ptr{Int64}, Cint, Ptr{Cint}, Ptr{Int64}, Ptr{Int64}),
                h[], length(rp), rp, ro, rc, length(sp), sp, so, sc), "mik_cgd_set_halo_plan", ctx.handle)
    check(ccall((:mik_cgd_set_comm, libmik), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), h[], comm.handle), "mik_cgd_set_comm", ctx.handle)
    if ghosts !== nothing
        # transport 3, halo pushed into the neighbours' landing buffers: ghosts = allgather (as for PartitionLink).  Every rank exports the
        # landing buffer the library allocated for its plan, the host gathers (handle, ghost count, receive segments) of every rank
        mine = zeros(UInt8, 64)
        check(ccall((:mik_cgd_ghost_export, libmik), Cint, (Ptr{Cvoid}, Ptr{UInt8}), h[], mine), "mik_cgd_ghost_export", ctx.handle)
        info = ghosts((mine, Int64(n_ext - n_loc), recv))
        gh = reduce(vcat, [i[1] for i in info]); gcnt = Int64[i[2] for i in info]; gdst = landing_targets(send, info, rank)
        GC.@preserve gh gcnt gdst check(ccall((:mik_cgd_connect_ghosts, libmik), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Ptr{Int64}, Ptr{Int64}), h[], gh, gcnt, gdst),
                                        "mik_cgd_connect_ghosts", ctx.handle)
    end
    res = Ref{Cdouble}(); tol = Ref{Cdouble}()
    check(ccall((:mik_cgd_init, libmik), Cint, (Ptr{Cvoid}, Ref{Cdouble}, Ref{Cdouble}), h[], res, tol), "mik_cgd_init", ctx.handle)
    it = HipDistCG{T}(h[], ctx, Any[x, b, u_ext, r, c, sbuf, dot_all, rr_all, sidx, A_loc, comm], res[], tol[], maxiter)
    finalizer(i -> alive(i.ctx) && ccall((:mik_cgd_destroy, libmik), Cint, (Ptr{Cvoid},), i.handle), it)
    it
end
"(runs, rows, merged) -- the contiguous row runs a rank updates, packs and puts on the wire ahead of the sweep over u (0 runs: the halo follows the sweep)"
function halo_early(it::HipDistCG)
    runs = Ref{Cint}(0); rows = Ref{Int64}(0); merged = Ref{Cint}(0)
    check(ccall((:mik_cgd_halo_early, libmik), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Int64}, Ref{Cint}), it.handle, runs, rows, merged), "mik_cgd_halo_early", it.ctx.handle)
    (Int(runs[]), Int(rows[]), merged[] != 0)
end
function iterate_many!(it::HipDistCG, iteration::Int, max_steps::Int)
    res = Vector{Cdouble}(undef, max(max_steps, 1)); nd = Ref{Int64}(0)
    check(ccall((:mik_cgd_iterate_many, libmik), Cint, (Ptr{Cvoid}, Int64, Int64, Ptr{Cdouble}, Ref{Int64}), it.handle, iteration, max_steps, res, nd),
          "mik_cgd_iterate_many", it.ctx.handle)
    resize!(res, nd[])
    isempty(res) || (it.residual = res[end])
    res
end

"""
orthogonalize_and_normalize!(V::Vector{Vector}, w, h, ModifiedGramSchmidt()) -- src/orthogonalize.jl:53-65 -- for a basis kept as
separate device vectors: one C call, same arithmetic and bits as the matrix method (mik_orthogonalize_vectors).
"""
function IterativeSolvers.orthogonalize_and_normalize!(V::Vector{HipVector{T}}, w::HipVector{T}, h::AbstractVector{T},
                                                       ::IterativeSolvers.ModifiedGramSchmidt) where {T<:MikFloat}
    k = length(V)
    ptrs = Ptr{Cvoid}[v.ptr for v in V]
    hh = Vector{T}(undef, max(k, 1))
    nrm = Ref{T}(zero(T))
    GC.@preserve V ptrs hh check(ccall((:mik_orthogonalize_vectors, libmik), Cint,
        (Ptr{Cvoid}, Cint, Int64, Cint, Ptr{Ptr{Cvoid}}, Ptr{Cvoid}, Ptr{T}, Ref{T}),
        w.ctx.handle, dtype_code(T), w.n, k, ptrs, w.ptr, hh, nrm), "mik_orthogonalize_vectors", w.ctx.handle)
    copyto!(h, 1, hh, 1, k)
    nrm[]
end

end # module
