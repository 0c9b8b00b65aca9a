# runtests.jl -- the reference's own router / CFMM tests, driven through AMDRouter on an MI355X, plus the check this
# repository cannot make anywhere else: the DEVICE against the REFERENCE ITSELF (CFMMRouter.jl's find_arb! / route! running in
# the same Julia process), pool by pool and bit for bit.
#
#     CFMM_AMD_LIB=$PWD/cfmmrouter.jl_amd/libcfmm_amd.so julia --project=julia -e 'using Pkg; Pkg.test()'
#
# STATUS: never executed (no Julia toolchain in the build image).  tests/test_julia_binding_static.py parses this file and
# checks that every AMDRouter verb it calls exists in src/CFMMRouterAMD.jl; the same scenarios run in Python against the C
# restatement of the reference (tests/test_host_cpu.py, tests/test_gpu_parity.py, tests/test_oracle_kat.py).
#
# What mirrors what (paths relative to the reference root):
#   "device sweep vs CFMMRouter.find_arb!"   the instances of test/cfmms.jl:64-107 (two-coin) and :117-199 (UniV3 fixture, 14 scenarios)
#   "arbitrage markets"                      test/arb.jl:41-89 with its feasibility predicates (:5-28), `all_flows .== netflows(r)` included
#   "swap markets"                           test/swap.jl:1-52
#   "README quick start"                     README.md:27-38
using CFMMRouter
using CFMMRouterAMD
using Test
using LinearAlgebra, Random, StatsBase

const CR = CFMMRouter
const FEAS_TOL = 1e-4                                   # test/arb.jl:3

# One pool, one price pair: the reference's per-pool method on the CPU (src/cfmms.jl:130-140, :185-196, :339-395).
function reference_trade(pool, v)
    Δ, Λ = zeros(2), zeros(2)
    CR.find_arb!(Δ, Λ, pool, v[pool.Ai])
    return Δ, Λ
end

# A vector of pools swept by the device at prices v (find_arb!(r::Router, v), src/router.jl:38-42), then compared row by
# row with the reference: `exact` families must agree in every bit, the others within `rtol` of the pool's reserve scale.
function device_matches_reference(pools, n, v; exact=true, rtol=1e-12)
    r = AMDRouter(LinearNonnegative(ones(n)), pools, n)
    find_arb!(r, v)
    for (i, pool) in enumerate(pools)
        Δ, Λ = reference_trade(pool, v)
        if exact
            @test r.Δs[i] == Δ
            @test r.Λs[i] == Λ
        else
            scale = max(1.0, maximum(pool.R))
            @test maximum(abs.(r.Δs[i] .- Δ)) <= rtol * scale
            @test maximum(abs.(r.Λs[i] .- Λ)) <= rtol * scale
        end
    end
    # Ψ: the device's reduction against the reference's serial loop over the reference's own rows
    ψ = zeros(n)
    for pool in pools
        Δ, Λ = reference_trade(pool, v)
        ψ[pool.Ai] .+= Λ .- Δ
    end
    @test maximum(abs.(netflows(r; exact=false) .- ψ)) <= 1e-12 * max(1.0, maximum(abs.(ψ)))
    exact && @test netflows(r) == ψ                       # exact = true (default): the reference's loop, bit for bit
    return r
end

# test/arb.jl:5-23 as predicates on an AMDRouter
function primal_feasible(r; arbitrage=true)
    flows = zero(r.v)
    for (Δ, Λ, pool) in zip(r.Δs, r.Λs, r.cfmms)
        @test all(Δ .>= -FEAS_TOL) && all(Λ .>= -FEAS_TOL)
        @test CR.ϕ(pool, R=pool.R + pool.γ * Δ - Λ) >= CR.ϕ(pool) - sqrt(eps())
        flows[pool.Ai] .+= Λ - Δ
    end
    @test all(flows .== netflows(r))                      # test/arb.jl:16, unedited semantics
    if arbitrage
        @test all(flows .>= -FEAS_TOL)
    else
        @test sum(flows .>= -FEAS_TOL) == 1
    end
end

# test/arb.jl:25-28
function dual_feasible(r)
    @test all(r.v .>= CR.lower_limit(r.objective) .- FEAS_TOL)
    @test all(r.v .<= CR.upper_limit(r.objective) .+ FEAS_TOL)
end

function random_product_market(npools, ncoins; seed=1234, fee=1.0)
    Random.seed!(seed)
    pools = Vector{CFMM{Float64}}(undef, npools)
    for i in 1:npools
        pools[i] = ProductTwoCoin(1000 * rand(2), fee, sample(1:ncoins, 2, replace=false))
    end
    return pools
end

@testset "CFMMRouterAMD" begin

@testset "device sweep vs CFMMRouter.find_arb!" begin
    Random.seed!(1234)
    fees = [rand() for _ in 1:3]
    reserves = [10 * rand(2) for _ in 1:3]
    prices = [rand(2) for _ in 1:3]

    @testset "ProductTwoCoin (bit for bit)" begin
        unit = ProductTwoCoin([1, 1], 1, [1, 2])
        for v in ([1.0, 1.0], [2.0, 2.0])                 # test/cfmms.jl:71-80: no arbitrage without a fee
            r = device_matches_reference(CFMM{Float64}[unit], 2, v)
            @test iszero(r.Δs[1]) && iszero(r.Λs[1])
        end
        r = device_matches_reference(CFMM{Float64}[unit], 2, [2.0, 1.0])          # test/cfmms.jl:82-86
        @test r.Δs[1][2] ≈ sqrt(2) - 1 && r.Λs[1][1] ≈ 1 - sqrt(1 / 2)
        pools = CFMM{Float64}[ProductTwoCoin(R, γ, [1, 2]) for R in reserves for γ in fees]
        for v in prices
            device_matches_reference(pools, 2, v)
        end
        @test_throws ArgumentError AMDRouter(LinearNonnegative(ones(2)), CFMM{Float64}[ProductTwoCoin([1, 1], 1, [1, 3])], 2)
    end

    @testset "GeometricMeanTwoCoin (log-space forms: 1e-12 of the reserve scale)" begin
        weights = [(w = rand(); [w, 1 - w]) for _ in 1:3]
        pools = CFMM{Float64}[GeometricMeanTwoCoin(R, w, γ, [1, 2]) for R in reserves for γ in fees for w in weights]
        for v in prices
            device_matches_reference(pools, 2, v; exact=false)
        end
        # option geomean_exact = 1: pow in the reference's operation order (device libm's pow: within 1e-12 as well)
        r = AMDRouter(LinearNonnegative(ones(2)), pools, 2)
        CFMMRouterAMD.set_option!(r, "geomean_exact", 1)
        find_arb!(r, prices[1])
        for (i, pool) in enumerate(pools)
            Δ, Λ = reference_trade(pool, prices[1])
            @test maximum(abs.(r.Δs[i] .- Δ)) <= 1e-12 * max(1.0, maximum(pool.R))
        end
    end

    @testset "UniV3 fixture (bit for bit)" begin
        # the reference's hand fixture (test/cfmms.jl:117-120) and its fourteen price scenarios (:127-199)
        for γ in (1.0, 0.997)
            pool = UniV3(15.0, [30.0, 20, 10, 5], [1.0, 2.0, 1.5, 0.0], γ, [1, 2])
            inside = γ == 1.0 ? 15.0 : 15.0 * (1 + γ) / 2
            for p in (inside, 16.0, 14.0, 25.0, 7.5, 4.0, 35.0)
                device_matches_reference(CFMM{Float64}[pool], 2, [p, 1.0])
            end
        end
    end

    @testset "mixed families in one router, shuffled" begin
        Random.seed!(7)
        n = 6
        pools = CFMM{Float64}[]
        for _ in 1:40
            Ai = sample(1:n, 2, replace=false)
            push!(pools, ProductTwoCoin(1000 * rand(2), rand((0.997, 1.0)), Ai))
            w = rand()
            push!(pools, GeometricMeanTwoCoin(1000 * rand(2), [w, 1 - w], 0.997, Ai))
            push!(pools, UniV3(15.0, [30.0, 20, 10, 5], [1.0, 2.0, 1.5, 0.0], 0.997, Ai))
        end
        shuffle!(pools)
        v = exp.(0.3 .* (2 .* rand(n) .- 1))
        r = AMDRouter(LinearNonnegative(ones(n)), pools, n)
        find_arb!(r, v)
        for (i, pool) in enumerate(pools)                   # router order is the caller's order, whatever the packing
            Δ, Λ = reference_trade(pool, v)
            if pool isa GeometricMeanTwoCoin
                @test maximum(abs.(r.Δs[i] .- Δ)) <= 1e-9
            else
                @test r.Δs[i] == Δ && r.Λs[i] == Λ
            end
        end
    end
end

@testset "arbitrage markets" begin
    @testset "two pools" begin
        r = AMDRouter(LinearNonnegative(ones(2)), CFMM{Float64}[ProductTwoCoin([100, 100], 1, [1, 2]), ProductTwoCoin([1, 2], 1, [1, 2])], 2)
        route!(r)
        primal_feasible(r)
        dual_feasible(r)
    end
    @testset "100 random pools, 10 coins, no fee" begin
        pools = random_product_market(100, 10)
        c = rand(10)
        r = AMDRouter(LinearNonnegative(c), pools, 10)
        route!(r)
        primal_feasible(r)
        dual_feasible(r)
        # the same market through the reference's Router: same solver (LBFGSB.jl), same callbacks up to summation order
        ref = Router(LinearNonnegative(c), pools, 10)
        route!(ref)
        @test maximum(abs.(netflows(r) .- netflows(ref))) <= 1e-6 * maximum(abs.(netflows(ref)))     # north_star's tolerance
        # ... and through the library's own L-BFGS-B (one ccall for the whole route!)
        rn = AMDRouter(LinearNonnegative(c), pools, 10)
        info = route_native!(rn)
        @test info.evaluations >= 2
        @test maximum(abs.(netflows(rn) .- netflows(ref))) <= 1e-6 * maximum(abs.(netflows(ref)))
        primal_feasible(rn)
        dual_feasible(rn)
    end
end

@testset "swap markets" begin
    @testset "two pools" begin
        r = AMDRouter(BasketLiquidation(1, [5.0, 0.0]), CFMM{Float64}[ProductTwoCoin([100, 100], 1, [1, 2]), ProductTwoCoin([1, 2], 1, [1, 2])], 2)
        route!(r)
        primal_feasible(r)
        dual_feasible(r)
    end
    @testset "100 random pools, 10 coins, no fee" begin
        pools = random_product_market(100, 10)
        Δin = vcat([0.0], 100 * rand(9))
        r = AMDRouter(BasketLiquidation(1, Δin), pools, 10)
        route!(r)
        primal_feasible(r; arbitrage=false)
        dual_feasible(r)
        ref = Router(BasketLiquidation(1, Δin), pools, 10)
        route!(ref)
        @test maximum(abs.(netflows(r) .- netflows(ref))) <= 1e-6 * maximum(abs.(netflows(ref)))
    end
end

@testset "README quick start" begin
    pools = CFMM{Float64}[ProductTwoCoin([1e6, 1e6], 1, [1, 2]), ProductTwoCoin([1e3, 2e3], 1, [1, 2])]
    r = AMDRouter(LinearNonnegative(ones(2)), pools, 2)
    route!(r)
    Ψ = netflows(r)
    @test abs(Ψ[1]) <= 1e-3 && isapprox(Ψ[2], 171.4; atol=0.1)
    rn = AMDRouter(LinearNonnegative(ones(2)), pools, 2)
    route_native!(rn)
    @test isapprox(netflows(rn)[2], Ψ[2]; rtol=1e-6)
end

@testset "update_reserves!: no arbitrage left at the same prices" begin
    pools = random_product_market(200, 8; seed=99, fee=0.997)
    r = AMDRouter(LinearNonnegative(rand(8)), pools, 8)
    route!(r)
    v = copy(r.v)
    before = [copy(p.R) for p in r.cfmms]
    traded = [copy(Δ) for Δ in r.Δs], [copy(Λ) for Λ in r.Λs]
    update_reserves!(r)
    for (i, p) in enumerate(r.cfmms)                       # R <- R + γΔ − Λ (src/cfmms.jl:26-31)
        @test p.R ≈ before[i] .+ p.γ .* traded[1][i] .- traded[2][i]
    end
    find_arb!(r, v)
    @test all(maximum(abs.(Δ)) <= 1e-9 for Δ in r.Δs)
end

end
