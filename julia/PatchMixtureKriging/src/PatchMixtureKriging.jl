# PatchMixtureKriging -- drop-in Julia front end of the MI355X implementation.
#
# Same module name, kernel types and fit/query API as RoyCCWang/PatchMixtureKriging, so that the
# reference's examples/mixGP.jl and examples/IBB1D.jl run unchanged; every numerical method is a
# `ccall` into libpmk_hip.so (C ABI: include/pmk.h).  Set ENV["PMK_LIB"] to the library path
# (default: ../../../patchmixturekriging_amd/csrc/libpmk_hip.so relative to this file).
#
# NOTE: Julia is not installed in the build container, so this file has never been executed there.
# It is kept mechanical and is mirrored 1:1 by the Python package patchmixturekriging_amd, which is
# what the test-suite drives (see INTEGRATION.md).  Indices cross the ABI 0-based; this file adds 1.
module PatchMixtureKriging

using LinearAlgebra
import Libdl

const libpmk = get(ENV, "PMK_LIB",
    normpath(joinpath(@__DIR__, "..", "..", "..", "patchmixturekriging_amd", "csrc", "libpmk_hip.so")))

export RKHSProblemType, fitRKHS!, query!, constructkernelmatrix, evalkernel, evalquery,
       setuppartition, getpartitionlines!, organizetrainingsets, fetchhyperplanes,
       MixtureGPType, MixtureGPDebugType, fitmixtureGP!

# ------------------------------------------------------------------------------------------ errors
struct PMKError <: Exception
    msg::String
end
lasterror() = unsafe_string(ccall((:pmk_last_error, libpmk), Cstring, ()))
function check(rc::Integer, what::String)
    rc < 0 && throw(PMKError("$what failed ($rc): $(lasterror())"))
    return rc
end

# ------------------------------------------------------------------------------------------ context
const CTX = Ref{Ptr{Cvoid}}(C_NULL)
function context()
    if CTX[] == C_NULL
        h = Ref{Ptr{Cvoid}}(C_NULL)
        dev = parse(Int, get(ENV, "PMK_DEVICE", "0"))
        check(ccall((:pmk_ctx_create, libpmk), Cint, (Cint, Ref{Ptr{Cvoid}}), dev, h), "pmk_ctx_create")
        CTX[] = h[]
    end
    return CTX[]
end

# ------------------------------------------------------------------------------------------ kernel types
# same names and fields as src/misc/declarations.jl:18-111 of the reference
abstract type StationaryKernelType end
abstract type BrownianBridgeKernelType end
struct Spline12KernelType{T} <: StationaryKernelType; a::T; end
struct Spline32KernelType{T} <: StationaryKernelType; a::T; end
struct Spline34KernelType{T} <: StationaryKernelType; a::T; end
struct RationalQuadraticKernelType{T} <: StationaryKernelType; a::T; end
struct TunableRationalQuadraticKernelType{T} <: StationaryKernelType; a::T; w::T; end
struct ModulatedSqExpKernelType{T} <: StationaryKernelType; ϵ_sq::T; ν::T; end
struct GaussianKernel1DType{T} <: StationaryKernelType; ϵ_sq::T; end
struct BrownianBridge10{T} <: BrownianBridgeKernelType; a::T; end
struct BrownianBridge20{T} <: BrownianBridgeKernelType; a::T; end
struct BrownianBridge1ϵ{T} <: BrownianBridgeKernelType; ϵ::T; end
struct BrownianBridge2ϵ{T} <: BrownianBridgeKernelType; ϵ::T; end
struct BrownianBridgeSemiInfDomain{BT <: BrownianBridgeKernelType}; θ_base::BT; end

# pmk_kernel_desc { int32 family; int32 flags; double p[4]; }
struct KernelDesc
    family::Int32
    flags::Int32
    p::NTuple{4,Float64}
end
desc(f, p1 = 0.0, p2 = 0.0; flags = 0) = KernelDesc(Int32(f), Int32(flags), (Float64(p1), Float64(p2), 0.0, 0.0))
desc(θ::Spline34KernelType) = desc(1, θ.a[1])
desc(θ::Spline12KernelType) = desc(2, θ.a[1])
desc(θ::Spline32KernelType) = desc(3, θ.a[1])
desc(θ::GaussianKernel1DType) = desc(4, θ.ϵ_sq[1])
desc(θ::RationalQuadraticKernelType) = desc(5, θ.a[1])
desc(θ::TunableRationalQuadraticKernelType) = desc(6, θ.a[1], θ.w[1])
desc(θ::ModulatedSqExpKernelType) = desc(7, θ.ϵ_sq[1], θ.ν[1])
desc(θ::BrownianBridge10) = desc(10, θ.a)
desc(θ::BrownianBridge20) = desc(11, θ.a)
desc(θ::BrownianBridge1ϵ) = desc(12, θ.ϵ)
desc(θ::BrownianBridge2ϵ) = desc(13, θ.ϵ)
function desc(θ::BrownianBridgeSemiInfDomain)
    d = desc(θ.θ_base)
    return KernelDesc(d.family, Int32(1), d.p)
end

# ------------------------------------------------------------------------------------------ closure-carrying kernels
# src/misc/declarations.jl:113-135, src/RKHS/kernel.jl:31-67,87-139.  A closure cannot cross the C ABI and need not: every
# one of these kernels is a stationary canonical kernel on τ² = |p - q|² + Σᵢ (gᵢ(p) - gᵢ(q))² with gᵢ a weighted warp
# function.  The warp functions are evaluated HERE, once per point (the table the reference's FastAdaptiveKernelType
# keeps as w_X), and appended to the coordinates; the device evaluates the canonical kernel on the augmented points
# (input dimension + number of warps ≤ 4).
struct AdaptiveKernelType{KT}
    canonical_params::KT
    warpfunc::Function
end
struct FastAdaptiveKernelType{KT,T,D}
    canonical_kernel::KT
    warpfuncs::Vector{Function}
    w_X::Array{T,D}      # pre-computed warp map evaluations at the training positions (refreshed by constructkernelmatrix)
    s::Vector{T}
end
struct AdaptiveKernelMultiWarpType{KT,T}
    canonical_params::KT
    warpfuncs::Vector{Function}
    a::Vector{T}
end
# the DPP variants (declarations.jl:115-118, 141-149; kernel.jl:70-89, 102-113): the same warped kernel where p != q, and a
# point-dependent term where p == q (1 + g(p)², resp. 1 + self_gain Σ aₘ |wₘ(p)|) -- on the device a per-point addend of
# the kernel's diagonal (pmk_model_set_diag / pmk_query_set_diag)
struct AdaptiveKernelDPPType{KT}
    canonical_params::KT
    warpfunc::Function
end
struct AdaptiveKernelMultiWarpDPPType{KT,T}
    canonical_params::KT
    warpfuncs::Vector{Function}
    a::Vector{T}
    self_gain::T
end
const DPPKernel = Union{AdaptiveKernelDPPType,AdaptiveKernelMultiWarpDPPType}
const WarpedKernel = Union{AdaptiveKernelType,FastAdaptiveKernelType,AdaptiveKernelMultiWarpType,AdaptiveKernelDPPType,
                           AdaptiveKernelMultiWarpDPPType}
canonical(θ::AdaptiveKernelType) = θ.canonical_params
canonical(θ::FastAdaptiveKernelType) = θ.canonical_kernel
canonical(θ::AdaptiveKernelMultiWarpType) = θ.canonical_params
canonical(θ::AdaptiveKernelDPPType) = θ.canonical_params
canonical(θ::AdaptiveKernelMultiWarpDPPType) = θ.canonical_params
desc(θ::WarpedKernel) = desc(canonical(θ))
features(θ::AdaptiveKernelType, x) = [Float64(θ.warpfunc(x))]
features(θ::AdaptiveKernelDPPType, x) = [Float64(θ.warpfunc(x))]
features(θ::FastAdaptiveKernelType, x) = [Float64(θ.s[i] * θ.warpfuncs[i](x)) for i = 1:length(θ.warpfuncs)]
features(θ::AdaptiveKernelMultiWarpType, x) = [Float64(sqrt(θ.a[m]) * θ.warpfuncs[m](x)) for m = 1:length(θ.warpfuncs)]
features(θ::AdaptiveKernelMultiWarpDPPType, x) = [Float64(sqrt(θ.a[m]) * θ.warpfuncs[m](x)) for m = 1:length(θ.warpfuncs)]
# the diagonal term of a kernel at the points X: nothing for every kernel but the DPP variants
diagaddend(θ, X) = nothing
diagaddend(θ::AdaptiveKernelDPPType, X) = Float64[θ.warpfunc(x)^2 for x in X]
diagaddend(θ::AdaptiveKernelMultiWarpDPPType, X) =
    Float64[θ.self_gain * sum(θ.a[m] * abs(θ.warpfuncs[m](x)) for m = 1:length(θ.warpfuncs)) for x in X]

# ------------------------------------------------------------------------------------------ helpers
"""array2matrix (src/misc/utilities.jl:25-36): Vector{Vector{T}} -> D x N matrix"""
function array2matrix(X::Vector{Vector{T}})::Matrix{T} where T
    N = length(X); D = length(X[1])
    out = Matrix{T}(undef, D, N)
    for n = 1:N
        out[:, n] = X[n]
    end
    return out
end
pack(X::Vector{Vector{T}}) where T = Matrix{Float64}(array2matrix(X))
pack(X::Vector{T}) where T <: Real = reshape(Vector{Float64}(X), 1, length(X))
# the points the device evaluates kernel θ on: X itself, or X with the warp features appended
kpack(θ, X) = pack(X)
function kpack(θ::WarpedKernel, X::Vector{Vector{T}}) where T
    Xa = [vcat(Vector{Float64}(x), features(θ, x)) for x in X]
    length(Xa[1]) <= 4 || throw(PMKError("input dimension + number of warp functions must be <= 4 on the device path"))
    return pack(Xa)
end

"""convert2itpindex (src/misc/utilities.jl:562-579)"""
function convert2itpindex(x::Vector{T}, a::Vector{T}, b::Vector{T}, M::Vector{Int})::Vector{T} where T <: Real
    @assert length(x) == length(a) == length(b)
    return collect((x[d] - a[d]) / (b[d] - a[d]) * (M[d] - 1) + 1 for d = 1:length(x))
end

# ------------------------------------------------------------------------------------------ kernel matrix
"""constructkernelmatrix(X, θ)::Matrix{Float64} (src/RKHS/RKHS.jl:4-34); host matrix, exactly symmetric"""
function constructkernelmatrix(X, θ)::Matrix{Float64}
    if θ isa FastAdaptiveKernelType                     # RKHS.jl:141-146: refresh the warp table
        for i = 1:length(θ.warpfuncs), n = 1:length(X)
            θ.w_X[n, i] = θ.warpfuncs[i](X[n])
        end
    end
    Xm = kpack(θ, X); D, n = size(Xm)
    K = Matrix{Float64}(undef, n, n)
    d = Ref(desc(θ))
    check(ccall((:pmk_kernel_matrix, libpmk), Cint,
        (Ptr{Cvoid}, Ref{KernelDesc}, Cint, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Int64),
        context(), d, D, n, Xm, n, C_NULL, K, n), "constructkernelmatrix")
    g = diagaddend(θ, X)
    if g !== nothing                                    # kernel.jl:74, 108: where p == q (by distance: duplicates too)
        for i = 1:n, j = 1:i
            if i == j || X[i] == X[j]
                K[i, j] += g[i]
                i == j || (K[j, i] += g[i])
            end
        end
    end
    return K
end
"""constructkernelmatrix(X, Z, θ) (src/RKHS/RKHS.jl:95-110)"""
function constructkernelmatrix(X::Vector{Vector{T}}, Z::Vector{Vector{T}}, θ)::Matrix{T} where T
    Xm = kpack(θ, X); Zm = kpack(θ, Z); D, n = size(Xm); m = size(Zm, 2)
    K = Matrix{Float64}(undef, n, m)
    d = Ref(desc(θ))
    check(ccall((:pmk_kernel_matrix, libpmk), Cint,
        (Ptr{Cvoid}, Ref{KernelDesc}, Cint, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Int64),
        context(), d, D, n, Xm, m, Zm, K, n), "constructkernelmatrix")
    g = diagaddend(θ, X)
    if g !== nothing
        for i = 1:n, j = 1:m
            X[i] == Z[j] && (K[i, j] += g[i])
        end
    end
    return K
end
evalkernel(p::Vector{T}, q::Vector{T}, θ) where T = constructkernelmatrix([p], [q], θ)[1, 1]
evalkernel(p::T, q::T, θ) where T <: Real = constructkernelmatrix([[p]], [[q]], θ)[1, 1]
evalkernel(τ::T, θ::StationaryKernelType) where T <: Real = evalkernel([τ], [zero(T)], θ)

# ------------------------------------------------------------------------------------------ BSP tree
mutable struct HyperplaneType{T}          # partition.jl:3-9
    v::Vector{T}
    c::T
    HyperplaneType{T}(v, c) where T = new{T}(v, c)
    HyperplaneType{T}() where T = new{T}()
end
mutable struct PartitionDataType{T}       # partition.jl:11-16
    hp::HyperplaneType{T}
    X::Vector{Vector{T}}
    global_X_indices::Vector{Int}
    index::Int
end
mutable struct BinaryNode{T}              # partition.jl:18-29
    data::T
    parent::BinaryNode{T}
    left::BinaryNode{T}
    right::BinaryNode{T}
    BinaryNode{T}(data) where T = new{T}(data)
    BinaryNode{T}(data, parent::BinaryNode{T}) where T = new{T}(data, parent)
end
BinaryNode(data) = BinaryNode{typeof(data)}(data)

# the native tree (pmk_bsp*) of every root returned by setuppartition.  Weak keys: the table must not keep a root
# alive, or its finalizer (which frees the native tree) would never run.
const NATIVE = WeakKeyDict{Any,Ptr{Cvoid}}()
native(root) = get(NATIVE, root) do
    throw(PMKError("this node is not a root returned by setuppartition"))
end

function buildnodes(h::Ptr{Cvoid}, D::Int, levels::Int, X)
    P = Int(ccall((:pmk_bsp_num_leaves, libpmk), Int64, (Ptr{Cvoid},), h))
    N = Int(ccall((:pmk_bsp_num_points, libpmk), Int64, (Ptr{Cvoid},), h))
    hv = Matrix{Float64}(undef, D, P - 1); hc = Vector{Float64}(undef, P - 1)
    off = Vector{Int64}(undef, P + 1); inds = Vector{Int64}(undef, max(N, 1))
    check(ccall((:pmk_bsp_arrays, libpmk), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}, Ptr{Int64}),
        h, hv, hc, off, inds), "pmk_bsp_arrays")
    T = Float64
    k = Ref(0); leaf = Ref(0)
    function make(parent, depth)
        if depth == levels - 1
            leaf[] += 1
            gi = N > 0 ? Vector{Int}(inds[off[leaf[]]+1:off[leaf[]+1]] .+ 1) : Int[]
            data = PartitionDataType(HyperplaneType{T}(), Vector{Vector{T}}(undef, 0), gi, leaf[])
            return parent === nothing ? BinaryNode(data) : BinaryNode{typeof(data)}(data, parent)
        end
        k[] += 1
        data = PartitionDataType(HyperplaneType{T}(hv[:, k[]], hc[k[]]), Vector{Vector{T}}(undef, 0), Int[], 0)
        node = parent === nothing ? BinaryNode(data) : BinaryNode{typeof(data)}(data, parent)
        node.left = make(node, depth + 1)       # pre-order: left subtree first
        node.right = make(node, depth + 1)
        return node
    end
    return make(nothing, 0), off, inds
end

"""setuppartition(X, level) -> root, X_parts, X_parts_inds (src/patchwork/partition.jl:106-129);
`device = true` builds the tree on the GPU (pmk_bsp_build_device, same result bit for bit).  `sign_mode` / `dot_mode`
select between the two readings of two Julia-stdlib behaviours the reference's text does not fix (include/pmk.h)."""
function setuppartition(X::Vector{Vector{T}}, level; sign_mode::Int = 1, dot_mode::Int = 0, device::Bool = false) where T
    Xm = pack(X); D, N = size(Xm)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    if device
        check(ccall((:pmk_bsp_build_device, libpmk), Cint,
            (Ptr{Cvoid}, Cint, Int64, Ptr{Float64}, Cint, Cint, Cint, Ref{Ptr{Cvoid}}),
            context(), D, N, Xm, level, sign_mode, dot_mode, h), "setuppartition")
    else
        check(ccall((:pmk_bsp_build, libpmk), Cint, (Cint, Int64, Ptr{Float64}, Cint, Cint, Cint, Ref{Ptr{Cvoid}}),
            D, N, Xm, level, sign_mode, dot_mode, h), "setuppartition")
    end
    root, off, inds = buildnodes(h[], D, Int(level), X)
    NATIVE[root] = h[]
    let hh = h[]
        finalizer(r -> (ccall((:pmk_bsp_destroy, libpmk), Cvoid, (Ptr{Cvoid},), hh); nothing), root)
    end
    P = length(off) - 1
    X_parts_inds = [Vector{Int}(inds[off[l]+1:off[l+1]] .+ 1) for l = 1:P]
    X_parts = [Vector{Vector{Float64}}(X[ix]) for ix in X_parts_inds]      # labelleafnodes, partition.jl:131-159
    return root, X_parts, X_parts_inds
end

"""findpartition(x, root, levels) (partition.jl:248-262), 1-based leaf index"""
findpartition(x::Vector{T}, root, levels::Int) where T =
    Int(ccall((:pmk_bsp_findpartition, libpmk), Int64, (Ptr{Cvoid}, Ptr{Float64}), native(root), Vector{Float64}(x))) + 1

"""organizetrainingsets(root, levels, X0, ε) (partition.jl:301-357); `device = true`: pmk_bsp_assign_device"""
function organizetrainingsets(root, levels::Int, X0::Vector{Vector{T}}, ε::T; device::Bool = false) where T
    Xm = pack(X0); D, N = size(Xm)
    h = native(root)
    P = Int(ccall((:pmk_bsp_num_leaves, libpmk), Int64, (Ptr{Cvoid},), h))
    off = Vector{Int64}(undef, P + 1)
    # (ccall wants its argument types as a literal tuple at every call site)
    assign(o, i, lo, li) = device ?
        ccall((:pmk_bsp_assign_device, libpmk), Cint,
              (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Float64}, Float64, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}),
              context(), h, N, Xm, ε, o, i, lo, li) :
        ccall((:pmk_bsp_assign, libpmk), Cint,
              (Ptr{Cvoid}, Int64, Ptr{Float64}, Float64, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}),
              h, N, Xm, ε, o, i, lo, li)
    check(assign(off, C_NULL, C_NULL, C_NULL), "organizetrainingsets")
    inds = Vector{Int64}(undef, max(off[end], 1)); loff = Vector{Int64}(undef, N + 1); lists = similar(inds)
    check(assign(off, inds, loff, lists), "organizetrainingsets")
    X_set_inds = [Vector{Int}(inds[off[r]+1:off[r+1]] .+ 1) for r = 1:P]
    X_set = [X0[ix] for ix in X_set_inds]
    regions_list_set = [Vector{Int}(lists[loff[n]+1:loff[n+1]] .+ 1) for n = 1:N]
    problematic_inds = Vector{Vector{Int}}(undef, 0)
    return X_set, X_set_inds, regions_list_set, problematic_inds
end

"""fetchhyperplanes(root): pre-order (src/RKHS/mixtureGP.jl:322-334)"""
function fetchhyperplanes(root::BinaryNode{PartitionDataType{T}}) where T
    hps = Vector{HyperplaneType{T}}(undef, 0)
    function visit(node)
        isdefined(node.data.hp, :v) && push!(hps, node.data.hp)
        isdefined(node, :left) && visit(node.left)
        isdefined(node, :right) && visit(node.right)
    end
    visit(root)
    return hps
end

"""findneighbourpartitions(p, radius, root, levels, hps, home; δ) (mixtureGP.jl:339-405)"""
function findneighbourpartitions(p::Vector{T}, radius::T, root, levels, hps, p_region_ind::Int; δ::T = 1e-10) where T
    h = native(root); D = length(p); M = length(hps)
    reg = Vector{Int64}(undef, max(M, 1)); ts = Vector{Float64}(undef, M); zs = Matrix{Float64}(undef, D, M)
    keep = Vector{UInt8}(undef, M)
    k = ccall((:pmk_bsp_neighbours, libpmk), Int64,
        (Ptr{Cvoid}, Ptr{Float64}, Float64, Float64, Int64, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Ptr{UInt8}),
        h, Vector{Float64}(p), radius, δ, p_region_ind - 1, reg, ts, zs, keep)
    check(k, "findneighbourpartitions")
    return Vector{Int}(reg[1:k] .+ 1), ts, [zs[:, i] for i = 1:M], BitVector(keep .!= 0)
end

# plotting helpers of src/patchwork/visualize_2D.jl (host only)
function get2Dline(u::Vector{T}, c::T) where T
    return -u[1] / u[2], c / u[2]
end
function prunepartitionline(node, y::Vector{T}, t::Vector{T}) where T
    @assert length(y) == length(t)
    isdefined(node, :parent) || return y, t
    c = node.parent.data.hp.c; v = node.parent.data.hp.v
    isright = node.parent.right === node
    keep = [xor(dot(v, [t[n]; y[n]]) < c, isright) for n = 1:length(t)]
    return prunepartitionline(node.parent, y[keep], t[keep])
end
function getpartitionlines!(y_set::Vector{Vector{T}}, t_set, node::BinaryNode{PartitionDataType{T}}, level::Int,
                            min_t, max_t, max_N_t::Int, centroid::Vector{T}, max_dist::T) where T
    m, b = get2Dline(node.data.hp.v, node.data.hp.c)
    t = collect(LinRange(min_t, max_t, max_N_t)); y = m .* t .+ b
    near = [norm([t[n]; y[n]] - centroid) < max_dist for n = 1:length(t)]
    y_pruned, t_pruned = prunepartitionline(node, y[near], t[near])
    push!(y_set, y_pruned); push!(t_set, t_pruned)
    if level != 2
        getpartitionlines!(y_set, t_set, node.left, level - 1, min_t, max_t, max_N_t, centroid, max_dist)
        getpartitionlines!(y_set, t_set, node.right, level - 1, min_t, max_t, max_N_t, centroid, max_dist)
    end
    return nothing
end

# ------------------------------------------------------------------------------------------ mixture GP
mutable struct MixtureGPDebugType{T}      # mixtureGP.jl:5-35
    w_tilde_set::Vector{Vector{T}}; u_set::Vector{Vector{T}}; v_set::Vector{Vector{T}}
    region_inds_set::Vector{Vector{Int}}; p_region_ind_set::Vector{Int}
    hps_keep_flags_set::Vector{BitVector}; zs_set::Vector{Vector{Vector{T}}}; ts_set::Vector{Vector{T}}
end
MixtureGPDebugType(dummy_val::T) where T = MixtureGPDebugType{T}([], [], [], [], [], [], [], [])

"""U_set / L_set of a fitted model (mixtureGP.jl:43-46): the factors stay on the device (2 x 8 n² bytes per patch, 16.8 GB
at 256 x 2000) and `η.L_set[r]` / `η.U_set[r]` pull ONE patch to the host on first use and keep it, as the Python mirror
does (mixture.py, _LazyFactors).  Indexes and iterates like the reference's Vector."""
mutable struct LazyFactors{T,M} <: AbstractVector{M}
    model::Ref{Ptr{Cvoid}}      # shared with the MixtureGPType that owns the device model
    what::Int                   # pmk_model_get selector: 1 = L (chol_U.L), 2 = K without noise (U_set)
    n::Vector{Int}
    cache::Dict{Int,M}
end
Base.size(f::LazyFactors) = (length(f.n),)
Base.IndexStyle(::Type{<:LazyFactors}) = IndexLinear()
function Base.getindex(f::LazyFactors{T,M}, r::Int) where {T,M}
    haskey(f.cache, r) && return f.cache[r]
    f.model[] == C_NULL && throw(PMKError("fitmixtureGP! must run before U_set / L_set are read"))
    A = Matrix{T}(model_get(f.model[], r, f.what, f.n[r]))
    f.cache[r] = f.what == 1 ? LowerTriangular(A) : A          # exactly M in either case
    return f.cache[r]
end
Base.setindex!(f::LazyFactors, v, r::Int) = (f.cache[r] = v)

mutable struct MixtureGPType{T}           # mixtureGP.jl:38-52 (+ the device model handle)
    X_parts::Vector{Vector{Vector{T}}}
    c_set::Vector{Vector{T}}
    σ²_set::Vector{T}
    U_set::LazyFactors{T,Matrix{T}}
    L_set::LazyFactors{T,LowerTriangular{T,Matrix{T}}}
    hps::Vector{HyperplaneType{T}}
    model::Ptr{Cvoid}
    handle::Ref{Ptr{Cvoid}}     # the same handle, shared with U_set / L_set
end
function MixtureGPType(X_parts::Vector{Vector{Vector{T}}}, hps::Vector{HyperplaneType{T}}) where T
    N = length(X_parts)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    n = Int[length(X) for X in X_parts]
    η = MixtureGPType{T}(X_parts, Vector{Vector{T}}(undef, N), Vector{T}(undef, N),
                         LazyFactors{T,Matrix{T}}(h, 2, n, Dict{Int,Matrix{T}}()),
                         LazyFactors{T,LowerTriangular{T,Matrix{T}}}(h, 1, n, Dict{Int,LowerTriangular{T,Matrix{T}}}()),
                         hps, C_NULL, h)
    finalizer(e -> (e.model != C_NULL && ccall((:pmk_model_destroy, libpmk), Cvoid, (Ptr{Cvoid},), e.model); nothing), η)
    return η
end

function model_get(model::Ptr{Cvoid}, r::Int, what::Int, n::Int)
    out = what == 0 ? Vector{Float64}(undef, n) : Matrix{Float64}(undef, n, n)
    check(ccall((:pmk_model_get, libpmk), Cint, (Ptr{Cvoid}, Int64, Cint, Ptr{Float64}, Int64), model, r - 1, what, out, n),
          "pmk_model_get")
    return out
end

"""fitmixtureGP!(η, y_parts, θ, σ²) -> η (mixtureGP.jl:70-118).  c_set comes back to the host; U_set / L_set are read from
the device patch by patch when they are first indexed (LazyFactors); store_factors = true pulls all of them at once."""
function fitmixtureGP!(η::MixtureGPType{T}, y_parts::Vector{Vector{T}}, θ, σ²; store_factors::Bool = false) where T
    P = length(η.X_parts)
    # positions, or positions + warp values for a warp-feature kernel (the tree is built on the positions alone)
    Xm = [kpack(θ, X) for X in η.X_parts]; ys = [Vector{Float64}(y) for y in y_parts]
    n = Int64[size(x, 2) for x in Xm]; D = size(Xm[1], 1)
    for r = 1:P
        @assert length(ys[r]) == n[r]                      # mixtureGP.jl:298
    end
    if η.model != C_NULL
        ccall((:pmk_model_destroy, libpmk), Cvoid, (Ptr{Cvoid},), η.model)
        η.model = C_NULL; η.handle[] = C_NULL
    end
    h = Ref{Ptr{Cvoid}}(C_NULL); info = Vector{Int32}(undef, P); d = Ref(desc(θ))
    gs = [diagaddend(θ, X) for X in η.X_parts]          # the DPP kernels' own diagonal term, else nothing
    if gs[1] === nothing
        GC.@preserve Xm ys begin
            rc = ccall((:pmk_fit_batched, libpmk), Cint,
                (Ptr{Cvoid}, Ref{KernelDesc}, Float64, Cint, Int64, Ptr{Int64}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}},
                 Ref{Ptr{Cvoid}}, Ptr{Ptr{Float64}}, Ptr{Int32}),
                context(), d, σ², D, P, n, [pointer(x) for x in Xm], [pointer(y) for y in ys], h, C_NULL, info)
        end
        check(rc, "fitmixtureGP!")
    else
        GC.@preserve Xm ys gs begin
            check(ccall((:pmk_model_create, libpmk), Cint,
                (Ptr{Cvoid}, Cint, Int64, Ptr{Int64}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ref{Ptr{Cvoid}}),
                context(), D, P, n, [pointer(x) for x in Xm], [pointer(y) for y in ys], h), "pmk_model_create")
            check(ccall((:pmk_model_set_diag, libpmk), Cint, (Ptr{Cvoid}, Ptr{Ptr{Float64}}), h[], [pointer(g) for g in gs]),
                  "pmk_model_set_diag")
            check(ccall((:pmk_model_fit, libpmk), Cint, (Ptr{Cvoid}, Ref{KernelDesc}, Float64), h[], d, σ²), "pmk_model_fit")
            check(ccall((:pmk_model_info, libpmk), Cint, (Ptr{Cvoid}, Ptr{Int32}), h[], info), "pmk_model_info")
        end
    end
    η.model = h[]
    η.handle[] = h[]
    empty!(η.U_set.cache); empty!(η.L_set.cache)
    η.U_set.n = Int.(n); η.L_set.n = Int.(n)
    bad = findfirst(!=(0), info)
    bad === nothing || throw(PosDefException(Int(info[bad])))       # cholesky(U) of mixtureGP.jl:109
    cs = [Vector{Float64}(undef, Int(n[r])) for r = 1:P]          # every c_r in one device-to-host transfer
    GC.@preserve cs check(ccall((:pmk_model_get_weights, libpmk), Cint, (Ptr{Cvoid}, Ptr{Ptr{Float64}}), η.model,
                                [pointer(c) for c in cs]), "pmk_model_get_weights")
    for r = 1:P
        η.c_set[r] = cs[r]
        η.σ²_set[r] = σ²
        if store_factors
            η.L_set[r]; η.U_set[r]                                  # pulled and cached
        end
    end
    return η
end

"""querymixtureGP!(Yq, Vq, Xq, η, root, levels, radius, δ, θ, σ², weight_θ, debug_vars; debug_flag)
(mixtureGP.jl:159-294)"""
function querymixtureGP!(Yq::Vector{T}, Vq::Vector{T}, Xq::Vector{Vector{T}}, η::MixtureGPType{T}, root, levels,
                         radius::T, δ::T, θ, σ², weight_θ, debug_vars::MixtureGPDebugType{T};
                         debug_flag = false)::Nothing where T
    η.model == C_NULL && throw(PMKError("fitmixtureGP! must run before querymixtureGP!"))
    Nq = length(Xq); Xm = kpack(θ, Xq)                         # positions (+ warp values): the tree uses the positions
    resize!(Yq, Nq); resize!(Vq, Nq)                          # mixtureGP.jl:179-180
    check(ccall((:pmk_model_set_bsp, libpmk), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), η.model, native(root), 0), "pmk_model_set_bsp")
    q = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:pmk_query_create, libpmk), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Ref{Ptr{Cvoid}}), η.model, Nq, Xm, q), "pmk_query_create")
    try
        gq = diagaddend(θ, Xq)
        gq === nothing || check(ccall((:pmk_query_set_diag, libpmk), Cint, (Ptr{Cvoid}, Ptr{Float64}), q[], gq), "pmk_query_set_diag")
        check(ccall((:pmk_query_plan, libpmk), Cint, (Ptr{Cvoid}, Float64, Float64), q[], radius, δ), "pmk_query_plan")
        check(ccall((:pmk_query_items, libpmk), Cint, (Ptr{Cvoid}, Ref{KernelDesc}), q[], Ref(desc(θ))), "pmk_query_items")
        check(ccall((:pmk_query_mix, libpmk), Cint, (Ptr{Cvoid}, Ref{KernelDesc}, Int64, Int64), q[], Ref(desc(weight_θ)), 0, Nq), "pmk_query_mix")
        check(ccall((:pmk_query_fetch, libpmk), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), q[], Yq, Vq), "pmk_query_fetch")
        if debug_flag
            tot = Ref{Int64}(0)
            check(ccall((:pmk_query_counts, libpmk), Cint, (Ptr{Cvoid}, Ref{Int64}, Ptr{Int64}, Ptr{Int64}), q[], tot, C_NULL, C_NULL), "pmk_query_counts")
            home = Vector{Int64}(undef, Nq); off = Vector{Int64}(undef, Nq + 1); reg = Vector{Int64}(undef, max(tot[], 1))
            t = Vector{Float64}(undef, max(tot[], 1)); w = similar(t); u = similar(t); v = similar(t)
            check(ccall((:pmk_query_debug, libpmk), Cint,
                (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                q[], home, off, reg, t, w, u, v), "pmk_query_debug")
            # the reference resize!s every field to Nq and assigns (mixtureGP.jl:185-195): a reused struct does not grow
            resize!(debug_vars.w_tilde_set, Nq); resize!(debug_vars.u_set, Nq); resize!(debug_vars.v_set, Nq)
            resize!(debug_vars.region_inds_set, Nq); resize!(debug_vars.p_region_ind_set, Nq)
            resize!(debug_vars.hps_keep_flags_set, Nq); resize!(debug_vars.zs_set, Nq); resize!(debug_vars.ts_set, Nq)
            for j = 1:Nq
                s = off[j]+1:off[j+1]
                debug_vars.w_tilde_set[j] = w[s]; debug_vars.u_set[j] = u[s]; debug_vars.v_set[j] = v[s]
                debug_vars.region_inds_set[j] = Vector{Int}(reg[s[1:end-1]] .+ 1)
                debug_vars.p_region_ind_set[j] = Int(home[j]) + 1
                _, ts, zs, keep = findneighbourpartitions(Xq[j], radius, root, levels, η.hps, Int(home[j]) + 1; δ = δ)
                debug_vars.hps_keep_flags_set[j] = keep; debug_vars.zs_set[j] = zs; debug_vars.ts_set[j] = ts
            end
        end
    finally
        ccall((:pmk_query_destroy, libpmk), Cvoid, (Ptr{Cvoid},), q[])
    end
    return nothing
end

# ------------------------------------------------------------------------------------------ multi-GPU (one process per GPU)
"""Comm(rank, world, id): the library's own RCCL communicator (pmk_comm_create; collective).  Rank 0 makes `id` with
`comm_unique_id()` and ships the 128 bytes to the other ranks by any host channel (MPI.jl's `MPI.Bcast!`, a file ...)."""
mutable struct Comm
    h::Ptr{Cvoid}
    rank::Int
    world::Int
end
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    check(ccall((:pmk_comm_unique_id, libpmk), Cint, (Ptr{UInt8},), id), "pmk_comm_unique_id")
    return id
end
function Comm(rank::Integer, world::Integer, id::Vector{UInt8})
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:pmk_comm_create, libpmk), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{UInt8}, Ref{Ptr{Cvoid}}),
                context(), rank, world, id, h), "pmk_comm_create")
    c = Comm(h[], rank, world)
    finalizer(x -> (ccall((:pmk_comm_destroy, libpmk), Cvoid, (Ptr{Cvoid},), x.h); nothing), c)
    return c
end

"""querymixtureGP!(Yq, Vq, Xq_local, η_local, comm, root, levels, radius, δ, θ, σ², weight_θ): the sharded form of
querymixtureGP! (mixtureGP.jl:159-294).  This rank's `η_local` holds the leaves `rank*P/world+1 : (rank+1)*P/world` of the
replicated tree `root` (fit them with fitmixtureGP! on `X_set[that range]`), `Xq_local` is this rank's share of the
queries; requests travel to the leaf owners and (u, v) back over RCCL inside the library (pmk_query_predict_sharded)."""
function querymixtureGP!(Yq::Vector{T}, Vq::Vector{T}, Xq::Vector{Vector{T}}, η::MixtureGPType{T}, comm::Comm, root, levels,
                         radius::T, δ::T, θ, σ², weight_θ)::Nothing where T
    η.model == C_NULL && throw(PMKError("fitmixtureGP! must run before querymixtureGP!"))
    Nq = length(Xq); Xm = pack(Xq)
    resize!(Yq, Nq); resize!(Vq, Nq)
    P = Int(ccall((:pmk_model_num_patches, libpmk), Int64, (Ptr{Cvoid},), η.model))
    check(ccall((:pmk_model_set_bsp, libpmk), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), η.model, native(root), comm.rank * P), "pmk_model_set_bsp")
    q = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:pmk_query_create, libpmk), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Ref{Ptr{Cvoid}}), η.model, Nq, Xm, q), "pmk_query_create")
    try
        check(ccall((:pmk_query_predict_sharded, libpmk), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Ref{KernelDesc}, Ref{KernelDesc}, Float64, Float64, Ptr{Int64}),
            q[], comm.h, Ref(desc(θ)), Ref(desc(weight_θ)), radius, δ, C_NULL), "pmk_query_predict_sharded")
        check(ccall((:pmk_query_fetch, libpmk), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), q[], Yq, Vq), "pmk_query_fetch")
    finally
        ccall((:pmk_query_destroy, libpmk), Cvoid, (Ptr{Cvoid},), q[])
    end
    return nothing
end

"""querymixtureGP_allgather!(Yq, Vq, Xq_all, η_local, comm, root, levels, radius, δ, θ, σ², weight_θ): the replicated-query form
of the sharded predict step (pmk_query_predict_allgather): every rank passes ALL queries, evaluates the items of its own
leaves, one RCCL all-gather of the (u, v) slices, and every rank ends with all of Yq, Vq."""
function querymixtureGP_allgather!(Yq::Vector{T}, Vq::Vector{T}, Xq::Vector{Vector{T}}, η::MixtureGPType{T}, comm::Comm, root, levels,
                                   radius::T, δ::T, θ, σ², weight_θ)::Nothing where T
    η.model == C_NULL && throw(PMKError("fitmixtureGP! must run before querymixtureGP_allgather!"))
    Nq = length(Xq); Xm = pack(Xq)
    resize!(Yq, Nq); resize!(Vq, Nq)
    P = Int(ccall((:pmk_model_num_patches, libpmk), Int64, (Ptr{Cvoid},), η.model))
    check(ccall((:pmk_model_set_bsp, libpmk), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), η.model, native(root), comm.rank * P), "pmk_model_set_bsp")
    q = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:pmk_query_create, libpmk), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Ref{Ptr{Cvoid}}), η.model, Nq, Xm, q), "pmk_query_create")
    try
        check(ccall((:pmk_query_predict_allgather, libpmk), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Ref{KernelDesc}, Ref{KernelDesc}, Float64, Float64, Ptr{Int64}),
            q[], comm.h, Ref(desc(θ)), Ref(desc(weight_θ)), radius, δ, C_NULL), "pmk_query_predict_allgather")
        check(ccall((:pmk_query_fetch, libpmk), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), q[], Yq, Vq), "pmk_query_fetch")
    finally
        ccall((:pmk_query_destroy, libpmk), Cvoid, (Ptr{Cvoid},), q[])
    end
    return nothing
end

function querymixtureGP(Xq::Vector{Vector{T}}, η::MixtureGPType{T}, root, levels, radius::T, δ::T, θ, σ², weight_θ;
                        debug_flag = false) where T
    Yq = Vector{T}(undef, 0); Vq = Vector{T}(undef, 0)
    debug_vars = MixtureGPDebugType(one(T))
    querymixtureGP!(Yq, Vq, Xq, η, root, levels, radius, δ, θ, σ², weight_θ, debug_vars; debug_flag = debug_flag)
    return Yq, Vq, debug_vars
end
querymixtureGP(xq::Vector{T}, η::MixtureGPType{T}, root, levels, radius::T, δ::T, θ, σ², weight_θ; debug_flag = false) where T <: Real =
    querymixtureGP([xq], η, root, levels, radius, δ, θ, σ², weight_θ; debug_flag = debug_flag)

"""queryinner(xq, X, θ, c, L) -> (μ, σ²) (mixtureGP.jl:296-320): host factors are uploaded with pmk_model_load and
one strip of the prediction kernel runs against them"""
# the last (X, c, L) that queryinner uploaded, by identity: a loop over query points against one patch (dev/debug.jl:46)
# moves the n x n factor to the device once, not once per point
mutable struct InnerCache
    key::Tuple{UInt,UInt,UInt}
    model::Ptr{Cvoid}
end
const INNER_CACHE = InnerCache((UInt(0), UInt(0), UInt(0)), C_NULL)

function queryinner(xq::Vector{T}, X, θ, c, L) where T
    key = (objectid(X), objectid(c), objectid(L))
    if INNER_CACHE.model == C_NULL || INNER_CACHE.key != key
        if INNER_CACHE.model != C_NULL
            ccall((:pmk_model_destroy, libpmk), Cvoid, (Ptr{Cvoid},), INNER_CACHE.model)
            INNER_CACHE.model = C_NULL
        end
        Xm = pack(X); D, n = size(Xm)
        cc = Vector{Float64}(c); Lm = Matrix{Float64}(L)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        GC.@preserve Xm cc Lm begin
            check(ccall((:pmk_model_load, libpmk), Cint,
                (Ptr{Cvoid}, Cint, Int64, Ptr{Int64}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}, Ptr{Int64}, Ref{Ptr{Cvoid}}),
                context(), D, 1, Int64[n], [pointer(Xm)], [pointer(cc)], [pointer(Lm)], Int64[n], h), "pmk_model_load")
        end
        INNER_CACHE.model = h[]; INNER_CACHE.key = key
    end
    μ = Ref{Float64}(0.0); v = Ref{Float64}(0.0)
    check(ccall((:pmk_model_queryinner, libpmk), Cint,
        (Ptr{Cvoid}, Int64, Ref{KernelDesc}, Int64, Ptr{Float64}, Ref{Float64}, Ref{Float64}),
        INNER_CACHE.model, 0, Ref(desc(θ)), 1, Vector{Float64}(xq), μ, v), "queryinner")
    return μ[], v[]
end
queryinner!(kq::Vector{T}, xq, X, θ, c, L; min_v = 1e-12) where T = queryinner(xq, X, θ, c, L)

# ------------------------------------------------------------------------------------------ single problem
struct RKHSProblemType{Kernel_Type,T,X_Type}      # src/misc/declarations.jl:226-231
    c::Vector{T}
    X::Vector{X_Type}
    θ::Kernel_Type
    σ²::T
end

"""fitRKHS!(η, y): η.c[:] = (K + σ²I) \\ y (src/RKHS/RKHS.jl:182-217), one patch through the batched fit"""
function fitRKHS!(η, y::Vector{T}) where T
    @assert !isempty(η.X)
    @assert !isempty(y)
    @assert length(η.X) == length(y)
    Xm = kpack(η.θ, η.X); yy = Vector{Float64}(y); D, n = size(Xm)
    h = Ref{Ptr{Cvoid}}(C_NULL); info = Vector{Int32}(undef, 1); c = Vector{Float64}(undef, n); d = Ref(desc(η.θ))
    GC.@preserve Xm yy c begin
        rc = ccall((:pmk_fit_batched, libpmk), Cint,
            (Ptr{Cvoid}, Ref{KernelDesc}, Float64, Cint, Int64, Ptr{Int64}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}},
             Ref{Ptr{Cvoid}}, Ptr{Ptr{Float64}}, Ptr{Int32}),
            context(), d, η.σ²[1], D, 1, Int64[n], [pointer(Xm)], [pointer(yy)], h, [pointer(c)], info)
    end
    check(rc, "fitRKHS!")
    ccall((:pmk_model_destroy, libpmk), Cvoid, (Ptr{Cvoid},), h[])
    info[1] == 0 || throw(PosDefException(Int(info[1])))
    η.c[:] = c
    return nothing
end

"""query!(Yq, Xq, η): mean only (src/RKHS/RKHS.jl:220-247)"""
function query!(Yq::Vector{T}, Xq, η::RKHSProblemType) where T
    @assert !isempty(Xq)
    @assert size(Yq) == size(Xq)
    Xm = kpack(η.θ, η.X); Qm = kpack(η.θ, Xq); D, n = size(Xm); Nq = size(Qm, 2)
    out = Vector{Float64}(undef, Nq)
    check(ccall((:pmk_query_mean, libpmk), Cint,
        (Ptr{Cvoid}, Ref{KernelDesc}, Cint, Int64, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}),
        context(), Ref(desc(η.θ)), D, n, Xm, Vector{Float64}(η.c), Nq, Qm, out), "query!")
    Yq[:] = out
    return nothing
end

"""query!(Yq, Xq, η::RKHSProblemType{Vector{KT}}) (src/RKHS/RKHS.jl:278-305): one kernel per centre"""
function query!(Yq::Vector{T}, Xq::Vector{Vector{T}}, η::RKHSProblemType{Vector{KT},T}) where {KT,T}
    @assert !isempty(Xq)
    @assert size(Yq) == size(Xq)
    Xm = pack(η.X); Qm = pack(Xq); D, n = size(Xm); Nq = size(Qm, 2)
    @assert length(η.θ) == n
    ds = [desc(θ) for θ in η.θ]
    out = Vector{Float64}(undef, Nq)
    check(ccall((:pmk_query_mean_multi, libpmk), Cint,
        (Ptr{Cvoid}, Ptr{KernelDesc}, Cint, Int64, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}),
        context(), ds, D, n, Xm, Vector{Float64}(η.c), Nq, Qm, out), "query!")
    Yq[:] = out
    return nothing
end

"""setupGPquery(c, X, θ, σ²) -> fq, fq(xq) = (mean, variance) (src/RKHS/querying.jl:43-79).  The reference's closure
solves `A \\ k` by LU on every call; here K + σ²I is factorised once on the device and a call is one strip of the
prediction kernel.  The variance is `k(xq,xq) - k'(A \\ k)` as the reference returns it: not clamped."""
function setupGPquery(c::Vector{T}, X, θ, σ²::T)::Function where T
    Xm = kpack(θ, X); D, n = size(Xm)
    @assert length(c) == n
    h = Ref{Ptr{Cvoid}}(C_NULL); info = Vector{Int32}(undef, 1); y0 = zeros(Float64, n); d = Ref(desc(θ))
    GC.@preserve Xm y0 begin
        rc = ccall((:pmk_fit_batched, libpmk), Cint,
            (Ptr{Cvoid}, Ref{KernelDesc}, Float64, Cint, Int64, Ptr{Int64}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}},
             Ref{Ptr{Cvoid}}, Ptr{Ptr{Float64}}, Ptr{Int32}),
            context(), d, σ², D, 1, Int64[n], [pointer(Xm)], [pointer(y0)], h, C_NULL, info)
    end
    check(rc, "setupGPquery")
    info[1] == 0 || throw(PosDefException(Int(info[1])))
    cc = Vector{Float64}(c)
    GC.@preserve cc check(ccall((:pmk_model_set_weights, libpmk), Cint, (Ptr{Cvoid}, Ptr{Ptr{Float64}}), h[], [pointer(cc)]),
                          "pmk_model_set_weights")
    model = Ref(h[])
    finalizer(m -> (ccall((:pmk_model_destroy, libpmk), Cvoid, (Ptr{Cvoid},), m[]); nothing), model)
    return xx -> evalqueryGP!(model, xx, θ)
end
function evalqueryGP!(model::Ref{Ptr{Cvoid}}, xq::Vector{T}, θ)::Tuple{T,T} where T
    Qm = kpack(θ, [xq]); μ = Ref(0.0); v = Ref(0.0)
    check(ccall((:pmk_model_queryinner_ex, libpmk), Cint,
        (Ptr{Cvoid}, Int64, Ref{KernelDesc}, Int64, Ptr{Float64}, Float64, Ref{Float64}, Ref{Float64}),
        model[], 0, Ref(desc(θ)), 1, Qm, -Inf, μ, v), "evalqueryGP!")
    return μ[], v[]
end

"""evalquery(x, c, X, θ) (src/RKHS/querying.jl:2-5)"""
function evalquery(x::Vector{T}, c::Vector{T}, X::Vector{Vector{T}}, θ)::T where T
    y = Vector{T}(undef, 1)
    query!(y, [x], RKHSProblemType(c, X, θ, zero(T)))
    return y[1]
end

end # module
