"""
    ConicIPHIP

MI355X back end for ConicIP's `kktsolver` plugin hook: the per-iteration Newton step (NT-scaled KKT assembly, dense LDLᵀ,
triangular solves) runs in `libcipkkt.so` (hand-written HIP for gfx950, C ABI in `include/cipkkt.h`); ConicIP keeps its
`conicIP` / MathOptInterface surface and its Mehrotra loop (`src/ConicIP.jl:730-934`).

    using ConicIP, ConicIPHIP
    sol = conicIP(Q, c, A, b, cone_dims, G, d; kktsolver = kktsolver_hip)              # block elimination on the device
    sol = conicIP(Q, c, A, b, cone_dims, G, d; kktsolver = pivot(kktsolver_2x2_hip))   # the reference's own `pivot` around the 2×2 form
    # JuMP, with integration/moi_kktsolver.patch applied to ConicIP's src/MOI_wrapper.jl:
    model = Model(() -> ConicIP.Optimizer(kktsolver = kktsolver_hip))

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
    h = if A isa SparseMatrixCSC
        _cip_create_sparse(Q, A, G, cone_dims, route)
    else
        _cip_create_dense(Q, A, G, cone_dims, route)
    end

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

function _cip_create_sparse(Q, A::SparseMatrixCSC, G, cone_dims, route)
    n, m, p = size(Q, 1), size(A, 1), size(G, 1)
    At = sparse(A')                                        # CSC of A' == CSR of A
    rowptr = Cint.(At.colptr .- 1); colind = Cint.(At.rowval .- 1); val = Vector{Float64}(At.nzval)
    Qd, Gd = Matrix{Float64}(Q), Matrix{Float64}(G)
    ctype = Cint[_CONE_CODE[c[1]] for c in cone_dims]
    cdim  = Cint[c[2] for c in cone_dims]
    href = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve Qd Gd rowptr colind val ctype cdim begin
        prob = Ref(CipProblem(n, m, p, length(cone_dims), pointer(ctype), pointer(cdim),
                              pointer(Qd), n, Ptr{Float64}(C_NULL), 0,
                              pointer(rowptr), pointer(colind), pointer(val),
                              p > 0 ? pointer(Gd) : Ptr{Float64}(C_NULL), max(p, 1), route, 0))
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
    h = (A isa SparseMatrixCSC) ?
        _cip_create_sparse(Q, A, G, cone_dims, CIP_ROUTE_SCHUR) : _cip_create_dense(Q, A, G, cone_dims, CIP_ROUTE_SCHUR)
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

end # module
