# Smoke script of the Julia front end (SURVEY section 7.1 step 2): the flows of examples/IBB1D.jl and examples/mixGP.jl
# of the reference, without plotting, against libpmk_hip.so.  Julia is not part of the build image, so this script has
# never been executed there; run it on a GPU box that has Julia:
#     PMK_LIB=/path/to/libpmk_hip.so julia --project=julia/PatchMixtureKriging julia/smoke.jl
import PatchMixtureKriging
using LinearAlgebra, Random
const PMK = PatchMixtureKriging

# ---- IBB1D.jl:19-62 -------------------------------------------------------------------------------------------------
θ1 = PMK.BrownianBridge10(1.0)
σ² = 1e-5
N = 15
X1 = [[x] for x in LinRange(1e-5, 1 - 1e-5, N)]
f1 = x -> sinc(4 * x) * x^3
y1 = [f1(x[1]) for x in X1]
K = PMK.constructkernelmatrix(X1, θ1)
@assert K == K' && isposdef(K) && rank(K) == N
η1 = PMK.RKHSProblemType(zeros(N), X1, θ1, σ²)
PMK.fitRKHS!(η1, y1)
@assert norm((K + σ² * I) * η1.c - y1) / norm(y1) < 1e-10
xq = [[x] for x in LinRange(0, 1, 100)]
yq = Vector{Float64}(undef, 100)
PMK.query!(yq, xq, η1)
@assert all(isfinite, yq)

# ---- mixGP.jl:100-176 -----------------------------------------------------------------------------------------------
Random.seed!(25)
Nt = 850
X = [[-5 + 10 * rand(), -10 + 20 * rand()] for _ = 1:Nt]
A = [1.0 0.4; 0.4 1.0] .* 0.1
f = x -> sinc((dot(x, A * x) / 3.2)^2) * (norm(x) / 4)^3
y = f.(X)
θ = PMK.Spline34KernelType(1 / 15)
levels = 3
root, X_parts, X_parts_inds = PMK.setuppartition(X, levels)
X_set, X_set_inds, _, problematic = PMK.organizetrainingsets(root, levels, X, 1.5)
@assert isempty(problematic)
hps = PMK.fetchhyperplanes(root)
@assert length(hps) == length(X_set) - 1                       # patchGP_partitioning.jl:198
η = PMK.MixtureGPType(X_set, hps)
Y_set = [y[ix] for ix in X_set_inds]
PMK.fitmixtureGP!(η, Y_set, θ, σ²)
for r in eachindex(X_set)
    U = η.U_set[r] + σ² * I
    @assert norm(η.L_set[r] * η.L_set[r]' - U) / norm(U) < 1e-12
    @assert norm(U * η.c_set[r] - Y_set[r]) / (norm(U) * norm(η.c_set[r]) + norm(Y_set[r])) < 1e-12
end
Xq = vec([[a, b] for a in LinRange(-5, 5, 100), b in LinRange(-10, 10, 200)])
radius = 0.3
Yq = Vector{Float64}(undef, 0); Vq = Vector{Float64}(undef, 0)
dbg = PMK.MixtureGPDebugType(1.0)
PMK.querymixtureGP!(Yq, Vq, Xq, η, root, levels, radius, 1e-5, θ, σ², PMK.Spline34KernelType(1 / radius), dbg;
                    debug_flag = true)
@assert length(Yq) == length(Xq) && all(isfinite, Yq) && all(v -> 1e-12 <= v <= 1 + 1e-9, Vq)
for j in 1:500:length(Xq)                                      # visualization.jl:163-172: home region last, weight 1
    @assert length(dbg.u_set[j]) == length(dbg.w_tilde_set[j]) == length(dbg.region_inds_set[j]) + 1
    @assert dbg.w_tilde_set[j][end] == 1.0
end
println("julia smoke ok: ", length(X_set), " patches, ", length(Xq), " queries, max |Yq| = ", maximum(abs, Yq))
