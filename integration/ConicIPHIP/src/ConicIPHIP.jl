"""
    ConicIPHIP

MI355X back end for ConicIP's `kktsolver` plugin hook: the per-iteration Newton step (NT-scaled KKT assembly, dense LDLᵀ,
triangular solves) runs in `libcipkkt.so` (hand-written HIP for gfx950, C ABI in `include/cipkkt.h`); ConicIP keeps its
`conicIP` / MathOptInterface surface and its Mehrotra loop (`src/ConicIP.jl:730-934`).

    using ConicIP, ConicIPHIP
    sol = conicIP(Q, c, A, b, cone_dims, G, d; kktsolver = kktsolver_hip)              # block elimination on the device
    sol = conicIP(Q, c, A, b, cone_dims, G, d; kktsolver = pivot(kktsolver_2x2_hip))   # the reference's own `pivot` around the 2×2 form
    # the whole interior-point loop on the device (cip_conicip): same signature, keywords and Solution as conicIP
    sol = conicIP_hip(Q, c, A, b, cone_dims, G, d; optTol = 1e-6)
    sols = conicIP_hip_batch([(Q1, c1, A1, b1, K1), (Q2, c2, A2, b2, K2, G2, d2)]; optTol = 1e-6)   # cip_conicip_mixed
    # JuMP, with integration/moi_kktsolver.patch applied to ConicIP's src/MOI_wrapper.jl:
    model = Model(() -> ConicIP.Optimizer(kktsolver = kktsolver_hip))                 # plugin levels, ConicIP's loop
    model = Model(() -> ConicIP.Optimizer(solve = preprocess_conicIP_hip))           # pre-solve on the host, loop on the device

The library is looked up once, at module initialisation: `ENV["CONICIP_LIBCIPKKT"]` (a full path) if set, else
`Libdl.find_library` over `libcipkkt` in `LD_LIBRARY_PATH` and in `<this package>/../../conicip.jl_amd/cipkkt` (the in-tree
build of this repository).  This file cannot be executed in the repository's build image (no Julia): INTEGRATION.md walks
through every `ccall` and the C-ABI entry point it binds; `tests/test_integration_files.py` checks the symbol names and
argument counts used here against `include/cipkkt.h`.
"""
module ConicIPHIP

using ConicIP
using ConicIP: Block
using Libdl
using LinearAlgebra
using SparseArrays

export kktsolver_hip, kktsolver_hip_full3x3, kktsolver_2x2_hip, CIP_ROUTE_SCHUR, CIP_ROUTE_FULL3X3
export conicIP_hip, conicIP_hip_batch, preprocess_conicIP_hip, CipOptions, CipResult

const _libpath = Ref{String}("")
const _lib = Ref{Ptr{Cvoid}}(C_NULL)
# function pointers are looked up in the handle opened by __init__ (a `ccall` through a pointer needs no constant library name)
_sym(name::Symbol) = Libdl.dlsym(_lib[], name)

function __init__()
    path = get(ENV, "CONICIP_LIBCIPKKT", "")
    if isempty(path)
        here = normpath(joinpath(@__DIR__, "..", "..", "..", "conicip.jl_amd", "cipkkt"))
        path = Libdl.find_library(["libcipkkt"], [here])
    end
    isempty(path) && error("ConicIPHIP: libcipkkt.so not found; build it (python conicip.jl_amd/build.py) and set " *
                           "ENV[\"CONICIP_LIBCIPKKT\"] or LD_LIBRARY_PATH")
    _lib[] = Libdl.dlopen(path)     # fails here, loudly, rather than in the first ccall
    _libpath[] = path
    return
end

const CIP_ROUTE_SCHUR, CIP_ROUTE_FULL3X3 = Cint(0), Cint(1)
const _CONE_CODE = Dict("R" => Cint(0), "Q" => Cint(1), "S" => Cint(2))

_cipcheck(rc) = rc == 0 ? nothing :
    error("libcipkkt: " * unsafe_string(ccall(_sym(:cip_last_error), Cstring, ())))

mutable struct CipHandle
    ptr::Ptr{Cvoid}
    function CipHandle(p)
        h = new(p)
        finalizer(x -> ccall(_sym(:cip_destroy), Cint, (Ptr{Cvoid},), x.ptr), h)
        h
    end
end

# packed scaling, in cone order (layout documented in include/cipkkt.h).  conicIP computes its initial point with
# F = F⁻ᵀ = Block([Diagonal(ones(k)) for every cone]) (ConicIP.jl:704-706), so a (uniform) Diagonal element must be
# accepted for "Q" and "S" cones as well as the NT elements built inside the loop.
_uniform(D::Diagonal) = (d = D.diag[1]; (d > 0 && all(==(d), D.diag)) ? d :
    error("kktsolver_hip: a Diagonal scaling element of a Q/S cone must be a positive multiple of the identity"))
function _pack_scaling(F::Block, F⁻ᵀ::Block, cone_dims)
    out = Float64[]
    for (i, (ctype, k)) in enumerate(cone_dims)
        Fi = F[i]
        if ctype == "R"
            append!(out, Fi.diag)                       # Diagonal(sqrt.(s./v))       ConicIP.jl:598
        elseif ctype == "Q"
            if Fi isa Diagonal                          # d·I = diag(-β, β, …) + w wᵀ with β = d, w = √(2d)·e₁
                d = _uniform(Fi)
                push!(out, d); push!(out, sqrt(2d)); append!(out, zeros(k - 1))
            else
                push!(out, -Fi.A.diag[1])               # β   (J = Diagonal([-β; β…])) ConicIP.jl:189-192
                append!(out, vec(Fi.B) .* sqrt(Fi.D[1]))  # w
            end
        else
            if Fi isa Diagonal                          # vecm(RᵀXR) = d·vecm(X)  ⇔  R = √d·I
                d = _uniform(Fi); r = ConicIP.ord(zeros(k))
                append!(out, vec(Matrix(sqrt(d) * I, r, r))); append!(out, vec(Matrix(I / sqrt(d), r, r)))
            else
                append!(out, vec(Fi.R))                 # VecCongurance(R)            ConicIP.jl:208
                append!(out, vec(Matrix(F⁻ᵀ[i].R')))    # inv(R)  (F⁻ᵀ[i] = VecCongurance(inv(R)'))
            end
        end
    end
    out
end

"""
    kktsolver_hip(Q, A, G, cone_dims; route = CIP_ROUTE_SCHUR)

Drop-in `kktsolver` for `conicIP` running the Newton step on an AMD MI355X.  A `SparseMatrixCSC` A goes to the
device as CSR whatever the cones (O(nnz) Schur assembly for "R" / "Q" rows: the README box-QP, A = I; the rows of "S"
cones are expanded into a dense block on the device); a dense A is uploaded dense.
"""
function kktsolver_hip(Q, A, G, cone_dims; route = CIP_ROUTE_SCHUR)
    n, m, p = size(Q, 1), size(A, 1), size(G, 1)
    h = _cip_create(Q, A, G, cone_dims, route)                                # level 1

    function solve3x3gen(F, F⁻ᵀ)                                              # level 2
        packed = _pack_scaling(F, F⁻ᵀ, cone_dims)
        _cipcheck(ccall(_sym(:cip_set_scaling_packed), Cint, (Ptr{Cvoid}, Ptr{Float64}), h.ptr, packed))
        _cipcheck(ccall(_sym(:cip_factor), Cint, (Ptr{Cvoid},), h.ptr))   # asynchronous; cip_solve3x3 resolves it
        function solve3x3(x, y, z)                                            # level 3
            a, b, c = zeros(n), zeros(p), zeros(m)    # fresh, Julia-owned (they become fields of z / Δz, ConicIP.jl:690)
            _cipcheck(ccall(_sym(:cip_solve3x3), Cint,
                (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                h.ptr, Vector{Float64}(x), Vector{Float64}(y), Vector{Float64}(z), a, b, c))
            return (a, b, c)
        end
        return solve3x3
    end
    return solve3x3gen
end

# level 1 for whatever the user passed as A (the reference's tests use Matrix, SparseMatrixCSC and Diagonal / Id(n),
# test/runtests.jl:95,98,213,219; src/ConicIP.jl:18): anything sparse-structured goes over as CSR -- a Diagonal A at
# n = 8192 would otherwise be an n^2 upload of an identity
_as_csr_source(A::SparseMatrixCSC) = A
_as_csr_source(A::Diagonal) = sparse(A)
_as_csr_source(A::LinearAlgebra.Adjoint{<:Any, <:SparseMatrixCSC}) = sparse(A)
_as_csr_source(A::LinearAlgebra.Transpose{<:Any, <:SparseMatrixCSC}) = sparse(A)
_as_csr_source(A) = nothing
function _cip_create(Q, A, G, cone_dims, route)
    As = _as_csr_source(A)
    As === nothing ? _cip_create_dense(Q, A, G, cone_dims, route) : _cip_create_sparse(Q, As, G, cone_dims, route)
end

function _cip_create_dense(Q, A, G, cone_dims, route)
    n, m, p = size(Q, 1), size(A, 1), size(G, 1)
    Qd, Ad, Gd = Matrix{Float64}(Q), Matrix{Float64}(A), Matrix{Float64}(G)   # column-major, as the C ABI expects
    ctype = Cint[_CONE_CODE[c[1]] for c in cone_dims]
    cdim  = Cint[c[2] for c in cone_dims]
    href = Ref{Ptr{Cvoid}}(C_NULL)
    _cipcheck(ccall(_sym(:cip_create), Cint,
        (Cint, Cint, Cint, Cint, Ptr{Cint}, Ptr{Cint}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint, Ref{Ptr{Cvoid}}),
        n, m, p, length(cone_dims), ctype, cdim, Qd, Ad, Gd, route, href))
    CipHandle(href[])
end

# --- sparse A (e.g. the README box-QP, A = I): hand the CSR of A to cip_create_ex -------------------
# The SparseMatrixCSC of A' holds exactly the CSR arrays of A (colptr -> rowptr, rowval -> colind).
struct CipProblem                      # mirrors `cip_problem` of include/cipkkt.h field by field
    n::Cint; m::Cint; p::Cint
    ncones::Cint
    cone_type::Ptr{Cint}
    cone_dim::Ptr{Cint}
    Q::Ptr{Float64}; ldq::Cint
    A::Ptr{Float64}; lda::Cint
    A_rowptr::Ptr{Cint}; A_colind::Ptr{Cint}; A_val::Ptr{Float64}
    G::Ptr{Float64}; ldg::Cint
    route::Cint
    flags::Cint
end

# Everything a `cip_problem` points at, staged as Julia arrays that must stay alive (GC.@preserve) across the call that
# reads the struct.  A: CSR when sparse-structured (see _as_csr_source), else dense column-major.
struct _Staged
    n::Int; m::Int; p::Int
    ctype::Vector{Cint}; cdim::Vector{Cint}
    Qd::Matrix{Float64}; Ad::Matrix{Float64}; Gd::Matrix{Float64}
    rowptr::Vector{Cint}; colind::Vector{Cint}; val::Vector{Float64}
    sparseA::Bool
    route::Cint
end
function _stage(Q, A, G, cone_dims, route)
    n, m, p = size(Q, 1), size(A, 1), size(G, 1)
    ctype = Cint[_CONE_CODE[c[1]] for c in cone_dims]
    cdim  = Cint[c[2] for c in cone_dims]
    As = _as_csr_source(A)
    if As === nothing
        return _Staged(n, m, p, ctype, cdim, Matrix{Float64}(Q), Matrix{Float64}(A), Matrix{Float64}(G),
                       Cint[], Cint[], Float64[], false, route)
    end
    At = sparse(As')                                       # CSC of A' == CSR of A
    _Staged(n, m, p, ctype, cdim, Matrix{Float64}(Q), zeros(0, 0), Matrix{Float64}(G),
            Cint.(At.colptr .- 1), Cint.(At.rowval .- 1), Vector{Float64}(At.nzval), true, route)
end
# the struct itself: only valid while `st` is preserved
function _problem(st::_Staged)
    null = Ptr{Float64}(C_NULL)
    CipProblem(st.n, st.m, st.p, length(st.ctype), pointer(st.ctype), pointer(st.cdim),
               pointer(st.Qd), max(st.n, 1),
               (st.sparseA || st.m == 0) ? null : pointer(st.Ad), max(st.m, 1),
               st.sparseA ? pointer(st.rowptr) : Ptr{Cint}(C_NULL), st.sparseA ? pointer(st.colind) : Ptr{Cint}(C_NULL),
               st.sparseA ? pointer(st.val) : null,
               st.p > 0 ? pointer(st.Gd) : null, max(st.p, 1), st.route, 0)
end

function _cip_create_sparse(Q, A::SparseMatrixCSC, G, cone_dims, route)
    st = _stage(Q, A, G, cone_dims, route)
    href = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve st begin
        prob = Ref(_problem(st))
        _cipcheck(ccall(_sym(:cip_create_ex), Cint, (Ref{CipProblem}, Ref{Ptr{Cvoid}}), prob, href))
    end
    CipHandle(href[])
end

# --- the 2x2 form (src/ConicIP.jl:450-466; src/kktsolvers.jl:281-349) -------------------------------------------
# `kktsolver_2x2_hip` has the shape of ConicIP.kktsolver_2x2 and is meant to be wrapped by the reference's own
# `pivot`:   conicIP(...; kktsolver = pivot(kktsolver_2x2_hip))
# (cip_solve2x2 solves [Q + Aᵀ(FᵀF)⁻¹A  Gᵀ; G 0][Δy; Δw] = [y; w] on the factor of the Schur route).
function kktsolver_2x2_hip(Q, A, G, cone_dims)
    n, p = size(Q, 1), size(G, 1)
    h = _cip_create(Q, A, G, cone_dims, CIP_ROUTE_SCHUR)
    function solve2x2gen(F, F⁻ᵀ)
        packed = _pack_scaling(F, F⁻ᵀ, cone_dims)
        _cipcheck(ccall(_sym(:cip_set_scaling_packed), Cint, (Ptr{Cvoid}, Ptr{Float64}), h.ptr, packed))
        _cipcheck(ccall(_sym(:cip_factor), Cint, (Ptr{Cvoid},), h.ptr))
        function solve2x2(y, w)
            Δy, Δw = zeros(n), zeros(p)
            _cipcheck(ccall(_sym(:cip_solve2x2), Cint,
                (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                h.ptr, Vector{Float64}(y), Vector{Float64}(w), Δy, Δw))
            return (Δy, Δw)
        end
        return solve2x2
    end
    return solve2x2gen
end
# The 3x3 solver needs no wrapper (its Schur route performs the block elimination on the device):
#     conicIP(Q, c, A, b, cone_dims, G, d; kktsolver = kktsolver_hip)
# and for the literal 3x3 assembly of kktsolver_sparse (src/kktsolvers.jl:254-256):
#     conicIP(...; kktsolver = (Q, A, G, cd) -> kktsolver_hip(Q, A, G, cd; route = CIP_ROUTE_FULL3X3))

"`kktsolver` for the literal 3×3 assembly of `kktsolver_sparse` (src/kktsolvers.jl:254-256) on the device."
kktsolver_hip_full3x3(Q, A, G, cone_dims) = kktsolver_hip(Q, A, G, cone_dims; route = CIP_ROUTE_FULL3X3)

# --- the whole interior-point loop on the device (SURVEY 8 f1 from the reference side) --------------------------
# `cip_conicip` is src/ConicIP.jl:468-939 inside the library: every vector stays in HBM, the host sees scalars.  From
# Julia that removes what the plugin levels cannot: the reference's host loop does five dense mat-vecs per iteration
# and five more per refinement pass (src/ConicIP.jl:746-750, :912-915; 537 MB per Q*y at n = 8192).

struct CipOptions                      # mirrors `cip_options` of include/cipkkt.h field by field
    optTol::Cdouble; DTB::Cdouble; infeasTol::Cdouble; refinementThreshold::Cdouble
    maxRefinementSteps::Cint; maxIters::Cint; verbose::Cint
end
struct CipResult                       # mirrors `cip_result` of include/cipkkt.h field by field
    status::Cint; iter::Cint
    mu::Cdouble; prFeas::Cdouble; duFeas::Cdouble; muFeas::Cdouble; pobj::Cdouble; dobj::Cdouble
    n_factor::Cint; n_solve::Cint; trace_rows::Cint
    wall_s::Cdouble
end
CipResult() = CipResult(0, 0, 0.0, Inf, Inf, Inf, Inf, -Inf, 0, 0, 0, 0.0)
# CIP_STATUS_* -> the reference's status symbols (src/ConicIP.jl:786, :815-818, :847-850, :870-873, :936)
const _STATUS = (:None, :Optimal, :Infeasible, :Unbounded, :Abandoned, :Error)

_solution(y, w, v, r::CipResult) =
    ConicIP.Solution(y, w, v, _STATUS[r.status + 1], Int(r.iter), r.mu, r.prFeas, r.duFeas, r.muFeas, r.pobj, r.dobj)

# the reference's own input checks, same messages (src/ConicIP.jl:537-542)
function _check_dims(Q, c, A, b, G, d)
    n, m, p = length(c), size(A, 1), size(G, 1)
    size(Q, 1) != size(Q, 2) && error("Q is not square")
    size(b, 1) != m && error("Inconsistency in inequalities")
    size(Q, 1) != n && error("Inconsistency in inequalities/objective")
    size(A, 2) != n && m > 0 && error("Inconsistency in inequalities/objective")
    size(d, 1) != p && error("Inconsistency in equalities")
    size(G, 2) != n && error("Inconsistency in equalities/objective")
    (n, m, p)
end

"""
    conicIP_hip(Q, c, A, b, cone_dims, G = spzeros(0, length(c)), d = zeros(0); kwargs...) -> ConicIP.Solution

`conicIP` with the WHOLE Mehrotra loop on the MI355X (`cip_create(_ex)` + `cip_conicip`).  Positional arguments, keyword
names and defaults are `conicIP`'s (src/ConicIP.jl:468-509): `optTol = 1e-6`, `DTB = 0.01`, `verbose = true`,
`maxRefinementSteps = 3`, `maxIters = 100`, `infeasTol = optTol`, `refinementThreshold = optTol/1e7`; `cache_nestodd` is
accepted and ignored as in the reference; `kktsolver` is accepted and ignored (the device loop brings its own: the Schur
route by default, `route = CIP_ROUTE_FULL3X3` for the literal 3x3 assembly).  `stats = Ref{CipResult}()` receives the
raw result (factorisations, solves, wall-clock of the loop).
"""
function conicIP_hip(Q, c::AbstractVector, A, b::AbstractVector, cone_dims,
                     G = spzeros(0, length(c)), d = zeros(0);
                     kktsolver = nothing, optTol = 1e-6, DTB = 0.01, verbose = true, maxRefinementSteps = 3,
                     maxIters = 100, cache_nestodd = false, infeasTol = optTol, refinementThreshold = optTol / 1e7,
                     route = CIP_ROUTE_SCHUR, stats = nothing)
    n, m, p = _check_dims(Q, c, A, b, G, d)
    h = _cip_create(Q, A, G, cone_dims, route)
    opt = Ref(CipOptions(optTol, DTB, infeasTol, refinementThreshold, maxRefinementSteps, maxIters, verbose ? 1 : 0))
    res = Ref(CipResult())
    y, w, v = zeros(n), zeros(p), zeros(m)                  # fresh, Julia-owned: they become the Solution's fields
    cv, bv, dv = Vector{Float64}(c), Vector{Float64}(b), Vector{Float64}(d)
    _cipcheck(ccall(_sym(:cip_conicip), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{CipOptions}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
         Ref{CipResult}, Ptr{Float64}, Cint),
        h.ptr, cv, bv, dv, opt, y, w, v, res, C_NULL, 0))
    finalize(h)                                             # cip_destroy now: the handle holds GBs of HBM at n = 8192
    stats === nothing || (stats[] = res[])
    _solution(y, w, v, res[])
end

"""
    conicIP_hip_batch(problems; in_flight = 4, route = CIP_ROUTE_SCHUR, kwargs...) -> Vector{ConicIP.Solution}

Independent problems in, solutions out, on one GPU (`cip_conicip_mixed`): the problems that share a shape advance through
the loop in lock-step (one launch per step for all of them), the others through `in_flight` host threads inside the
library.  `problems[i]` is a tuple `(Q, c, A, b, cone_dims)` or `(Q, c, A, b, cone_dims, G, d)`; the keywords are
`conicIP`'s and apply to every problem.  (The reference solves one problem per `conicIP` call, src/ConicIP.jl:472-480: this
is N independent calls.  Sharding over several GPUs is one process per GPU, problem i on rank i mod N.)
"""
function conicIP_hip_batch(problems::AbstractVector; in_flight::Integer = 4, route = CIP_ROUTE_SCHUR,
                           optTol = 1e-6, DTB = 0.01, verbose = false, maxRefinementSteps = 3, maxIters = 100,
                           cache_nestodd = false, infeasTol = optTol, refinementThreshold = optTol / 1e7, stats = nothing)
    k = length(problems)
    k == 0 && return ConicIP.Solution[]
    staged = Vector{_Staged}(undef, k)
    cs = Vector{Vector{Float64}}(undef, k); bs = similar(cs); ds = similar(cs)
    ys = similar(cs); ws = similar(cs); vs = similar(cs)
    for (i, pr) in enumerate(problems)
        Q, c, A, b, K = pr[1], pr[2], pr[3], pr[4], pr[5]
        G = length(pr) >= 7 ? pr[6] : spzeros(0, length(c))
        d = length(pr) >= 7 ? pr[7] : zeros(0)
        n, m, p = _check_dims(Q, c, A, b, G, d)
        staged[i] = _stage(Q, A, G, K, route)
        cs[i], bs[i], ds[i] = Vector{Float64}(c), Vector{Float64}(b), Vector{Float64}(d)
        ys[i], ws[i], vs[i] = zeros(n), zeros(p), zeros(m)
    end
    opt = Ref(CipOptions(optTol, DTB, infeasTol, refinementThreshold, maxRefinementSteps, maxIters, verbose ? 1 : 0))
    res = fill(CipResult(), k)
    GC.@preserve staged cs bs ds ys ws vs begin
        probs = [_problem(st) for st in staged]
        ptrs(xs) = Ptr{Float64}[isempty(x) ? Ptr{Float64}(C_NULL) : pointer(x) for x in xs]
        # (a NULL entry is fine where the problem's m or p is 0: include/cipkkt.h, cip_conicip_mixed)
        _cipcheck(ccall(_sym(:cip_conicip_mixed), Cint,
            (Cint, Ptr{CipProblem}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ref{CipOptions},
             Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{CipResult}, Cint),
            k, probs, ptrs(cs), ptrs(bs), ptrs(ds), opt, ptrs(ys), ptrs(ws), ptrs(vs), res, in_flight))
    end
    stats === nothing || (stats[] = res)
    [_solution(ys[i], ws[i], vs[i], res[i]) for i in 1:k]
end

"""
    preprocess_conicIP_hip(Q, c, A, b, cone_dims, G, d; kwargs...)

`ConicIP.preprocess_conicIP` (src/preprocessor.jl:29-96) with `conicIP_hip` in place of `conicIP`: the rank-revealing
pre-solve stays on the host exactly as in the reference (`ConicIP.imcols`), the interior-point loop runs on the device.
This is what `ConicIP.Optimizer(solve = preprocess_conicIP_hip)` (integration/moi_kktsolver.patch) calls from JuMP.
"""
function preprocess_conicIP_hip(Q, c::AbstractVector, A, b::AbstractVector, cone_dims,
                                G = spzeros(0, length(c)), d = zeros(0); verbose = false, options...)
    n, m, p = length(c), size(A, 1), size(G, 1)
    (IP, pconsistent) = ConicIP.imcols(G, d)                                   # src/preprocessor.jl:55
    (ID, dconsistent) = ConicIP.imcols([Q A' G[IP, :]'], c)                    # :56
    if !(pconsistent && dconsistent)                                           # :58-61
        return ConicIP.Solution(zeros(n) / 0, zeros(p) / 0, zeros(m) / 0, :Infeasible, 0, NaN, NaN, NaN, NaN, NaN, NaN)
    end
    z = ones(n); z[ID] .= 0; Z = spdiagm(0 => z)                               # :75
    sol = conicIP_hip(Q + Z, c, A, b, cone_dims, G[IP, :], d[IP]; verbose = verbose, options...)   # :79-84
    w = zeros(size(G, 1)); w[IP] = sol.w; sol.w = w                            # :90
    return sol
end

end # module
