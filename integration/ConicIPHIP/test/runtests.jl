# Run on a box with Julia, ConicIP.jl, an MI355X and libcipkkt.so:   julia --project=integration/ConicIPHIP -e 'using Pkg; Pkg.test()'
# The checks mirror the reference's own end-to-end tests (test/runtests.jl:133-166: every kktsolver must give the same
# answer) with the HIP plugin in place of the shipped solvers.
using Test, LinearAlgebra, SparseArrays
using ConicIP, ConicIPHIP

@testset "kktsolver_hip against kktsolver_qr" begin
    n = 200
    M = [sin(0.37 * i * j + 0.11 * i) for i in 1:n, j in 1:n]        # deterministic, no RNG
    Q = M' * M / n + 0.1I
    c = [cos(0.9 * i) for i in 1:n]
    A = sparse(1.0I, n, n); b = zeros(n)
    K = [("R", n)]
    ref = conicIP(Q, c, A, b, K; optTol = 1e-7)
    for ks in (kktsolver_hip, kktsolver_hip_full3x3, pivot(kktsolver_2x2_hip))
        sol = conicIP(Q, c, A, b, K; optTol = 1e-7, kktsolver = ks)
        @test sol.status == ref.status == :Optimal
        @test sol.Iter == ref.Iter
        @test norm(sol.y - ref.y) <= 1e-6 * (1 + norm(ref.y))
    end
end

@testset "second-order and semidefinite cones, equalities" begin
    n = 30
    K = [("R", 6), ("Q", 5), ("S", 6), ("Q", 4)]
    m = sum(k for (_, k) in K)
    A = [sin(1.3 * i + 0.7 * j * j) for i in 1:m, j in 1:n] ./ sqrt(n)
    G = [cos(0.4 * i * j) for i in 1:3, j in 1:n]
    Q = Matrix(1.0I, n, n); c = [sin(2.0 * i) for i in 1:n]
    e = zeros(m); e[1:6] .= 1; e[7] = 1; e[12] = 1; e[15] = 1; e[17] = 1; e[18] = 1
    b = -e; d = zeros(3)
    ref = conicIP(Q, c, A, b, K, G, d; optTol = 1e-7)
    sol = conicIP(Q, c, A, b, K, G, d; optTol = 1e-7, kktsolver = kktsolver_hip)
    @test sol.status == ref.status
    @test sol.Iter == ref.Iter
    @test norm(sol.y - ref.y) <= 1e-6 * (1 + norm(ref.y))
end
