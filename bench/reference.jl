# bench/reference.jl -- runs the REFERENCE (CFMMRouter.jl, Julia CPU path) on a market file written
# by cfmmrouter.jl_amd/poolfile.py, so the reference and the MI355X path see identical pool bits.
#
#   python -c "import bench_market; ..."   (see bench/write_market.py)  ->  market.bin
#   julia -t auto bench/reference.jl market.bin
#
# NOT EXECUTED in the build image (no Julia toolchain): any number this script prints is the
# user's, not this repository's.  Mirrors benchmark/scaling.jl:36 (`@benchmark route!(router; v=v0)`)
# and additionally times find_arb!(router, v) alone (the sweep of src/router.jl:38-42).
using CFMMRouter, LinearAlgebra, Printf

function load_market(path)
    io = open(path, "r")
    String(read(io, 8)) == "CFMMAMD1" || error("not a CFMMAMD1 market file")
    rd(T, dims...) = read!(io, Array{T}(undef, dims...))
    n, nseg = rd(Int64, 2)
    kind, idx = rd(Int64, 2)
    vec = rd(Float64, n)
    objective = kind == 0 ? LinearNonnegative(vec) : BasketLiquidation(Int(idx), vec)
    v0 = rd(Int64, 1)[1] == 1 ? rd(Float64, n) : nothing
    cfmms = Vector{CFMM{Float64}}()
    for _ in 1:nseg
        k, m = rd(Int64, 2)
        if k == 0
            R = rd(Float64, 2, m); γ = rd(Float64, m); Ai = rd(Int64, 2, m)
            for i in 1:m
                push!(cfmms, ProductTwoCoin(R[:, i], γ[i], Ai[:, i]))
            end
        elseif k == 1
            R = rd(Float64, 2, m); w = rd(Float64, 2, m); γ = rd(Float64, m); Ai = rd(Int64, 2, m)
            for i in 1:m
                push!(cfmms, GeometricMeanTwoCoin(R[:, i], w[:, i], γ[i], Ai[:, i]))
            end
        else
            cp = rd(Float64, m); γ = rd(Float64, m); Ai = rd(Int64, 2, m); off = rd(Int64, m + 1)
            T = off[end]
            ticks = rd(Float64, T); liq = rd(Float64, T)
            for i in 1:m
                r = (off[i]+1):off[i+1]
                push!(cfmms, UniV3(cp[i], ticks[r], liq[r], γ[i], Ai[:, i]))
            end
        end
    end
    close(io)
    return objective, cfmms, Int(n), v0
end

objective, cfmms, n, v0 = load_market(ARGS[1])
router = Router(objective, cfmms, n)
v = isnothing(v0) ? ones(n) ./ n : v0

# sweep alone: find_arb!(r, v)  (src/router.jl:38-42)
find_arb!(router, v)
reps = max(3, round(Int, 2e7 / length(cfmms)))
t = @elapsed for _ in 1:reps
    find_arb!(router, v)
end
@printf("find_arb!: %d pools, %d threads, %.3e pool-evals/s\n", length(cfmms), Threads.nthreads(),
        length(cfmms) * reps / t)

# route!  (benchmark/scaling.jl:36)
route!(router; v=v0)
ts = [(@elapsed route!(router; v=v0)) for _ in 1:5]
Ψ = netflows(router)
@printf("route!: min %.3f ms, median %.3f ms; max|Ψ| = %.6e\n", 1e3 * minimum(ts), 1e3 * sort(ts)[3], maximum(abs, Ψ))
open(ARGS[1] * ".psi", "w") do io
    write(io, Ψ)            # float64 little-endian netflows, for a bitwise-level comparison with the GPU path
end
