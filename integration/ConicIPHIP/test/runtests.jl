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

# ---- the whole loop on the device: conicIP_hip must walk conicIP's trajectory (same status and iteration count, same
# iterate to the solver's tolerance) on every cone type, with and without equalities, for each concrete type of A the
# reference's tests pass (test/runtests.jl:95,98,213,219)
@testset "conicIP_hip against conicIP" begin
    n = 120
    M = [sin(0.37 * i * j + 0.11 * i) for i in 1:n, j in 1:n]
    Q = M' * M / n + 0.1I
    c = [cos(0.9 * i) for i in 1:n]
    b = zeros(n); K = [("R", n)]
    ref = conicIP(Q, c, sparse(1.0I, n, n), b, K; optTol = 1e-7, verbose = false)
    for A in (sparse(1.0I, n, n), Matrix(1.0I, n, n), Id(n))          # SparseMatrixCSC, Matrix, Diagonal (src/ConicIP.jl:18)
        st = Ref{CipResult}()
        sol = conicIP_hip(Q, c, A, b, K; optTol = 1e-7, verbose = false, stats = st)
        @test sol isa ConicIP.Solution
        @test sol.status == ref.status == :Optimal
        @test sol.Iter == ref.Iter
        @test norm(sol.y - ref.y) <= 1e-6 * (1 + norm(ref.y))
        @test norm(sol.v - ref.v) <= 1e-6 * (1 + norm(ref.v))
        @test isapprox(sol.pobj, ref.pobj; rtol = 1e-6) && isapprox(sol.dobj, ref.dobj; rtol = 1e-6)
        @test st[].n_factor == sol.Iter + 1                        # one factorisation per iteration + the initial point (:706, :737)
    end
    # mixed cones + equalities, both elimination routes
    n = 30
    K = [("R", 6), ("Q", 5), ("S", 6), ("Q", 4)]
    m = sum(k for (_, k) in K)
    A = [sin(1.3 * i + 0.7 * j * j) for i in 1:m, j in 1:n] ./ sqrt(n)
    G = [cos(0.4 * i * j) for i in 1:3, j in 1:n]
    Q = Matrix(1.0I, n, n); c = [sin(2.0 * i) for i in 1:n]
    e = zeros(m); e[1:6] .= 1; e[7] = 1; e[12] = 1; e[15] = 1; e[17] = 1; e[18] = 1
    b = -e; d = zeros(3)
    ref = conicIP(Q, c, A, b, K, G, d; optTol = 1e-7, verbose = false)
    for route in (CIP_ROUTE_SCHUR, CIP_ROUTE_FULL3X3)
        sol = conicIP_hip(Q, c, A, b, K, G, d; optTol = 1e-7, verbose = false, route = route)
        @test sol.status == ref.status
        @test sol.Iter == ref.Iter
        @test norm(sol.y - ref.y) <= 1e-6 * (1 + norm(ref.y))
        @test norm(sol.w - ref.w) <= 1e-6 * (1 + norm(ref.w))
    end
    # statuses other than :Optimal come back as the reference's symbols (test/runtests.jl:441-505)
    inf = conicIP_hip(Matrix(1.0I, 2, 2), zeros(2), [1.0 0; -1.0 0], [1.0, 1.0], [("R", 2)]; verbose = false)   # y1 >= 1, -y1 >= 1
    @test inf.status == :Infeasible
    @test conicIP_hip(Q, c, A, b, K, G, d; maxIters = 2, verbose = false).status == :Abandoned
    @test_throws ErrorException conicIP_hip(ones(2, 3), zeros(2), zeros(0, 2), zeros(0), [])      # "Q is not square" (:538)
end

@testset "conicIP_hip_batch: lock-step groups + thread pool" begin
    mk(n, s) = begin
        M = [sin(0.37 * i * j + 0.11 * i + s) for i in 1:n, j in 1:n]
        (M' * M / n + 0.1I, [cos(0.9 * i + s) for i in 1:n], sparse(1.0I, n, n), zeros(n), [("R", n)])
    end
    probs = Any[mk(64, 0.1), mk(64, 0.2), mk(48, 0.3), mk(64, 0.4)]       # three of one shape (lock-step) + a lone one
    st = Ref{Vector{CipResult}}()
    sols = conicIP_hip_batch(probs; optTol = 1e-7, stats = st)
    @test length(sols) == 4 && length(st[]) == 4
    for (pr, sol) in zip(probs, sols)
        ref = conicIP(pr...; optTol = 1e-7, verbose = false)
        @test sol.status == ref.status == :Optimal
        @test sol.Iter == ref.Iter
        @test norm(sol.y - ref.y) <= 1e-6 * (1 + norm(ref.y))
    end
end

@testset "preprocess_conicIP_hip" begin
    # a redundant equality row: the pre-solve drops it on the host, the loop runs on the device (src/preprocessor.jl:55-90)
    n = 20
    Q = Matrix(1.0I, n, n); c = [sin(1.0 * i) for i in 1:n]
    A = sparse(1.0I, n, n); b = -ones(n); K = [("R", n)]
    g = [cos(0.3 * j) for j in 1:n]'
    G = [g; 2g]; d = [1.0, 2.0]
    ref = preprocess_conicIP(Q, c, A, b, K, G, d; optTol = 1e-7)
    sol = preprocess_conicIP_hip(Q, c, A, b, K, G, d; optTol = 1e-7)
    @test sol.status == ref.status == :Optimal
    @test length(sol.w) == 2
    @test norm(sol.y - ref.y) <= 1e-6 * (1 + norm(ref.y))
end
