# reference_fixtures.jl -- pins this repository's oracle to the REAL CMF.jl.   NOT RUN IN THIS REPOSITORY: the build image and
# the GPU boxes have no Julia (SURVEY.md section 8c), which is why DESIGN.md section 2 says "parity unpinned by the reference".
# Anyone who has Julia and a checkout of degleris1/CMF.jl closes that gap with one command:
#
#     julia tools/reference_fixtures.jl /path/to/CMF.jl
#     CMF_REQUIRE_REF=1 python -m pytest tests/test_reference_fixtures.py -q          # oracle (CPU) and, with -m gpu, the HIP path
#
# What it does: `include`s the reference's own module (src/CMF.jl of the checkout -- nothing of this repository is loaded),
# reads the inputs of the small committed fixtures from tests/golden/ref_inputs/<name>.h5 (written by
# tests/golden/export_reference_inputs.py in HDF5.jl's conventions: data N x T, W0 K x N x L, H0 K x T, max_itr, l1W, l2W, l1H,
# l2H as scalars, rule as a string), runs
#
#     fit_cnmf(data; L, K, alg=MultUpdate | HALSUpdate | PGDUpdate, max_itr, W_init=W0, H_init=H0,
#              check_convergence=false, l1W, l2W, l1H, l2H)                       (src/model.jl:58-85 -> src/algs/alternating.jl:16-71)
#
# and writes loss_hist, W, H to tests/golden/ref_<name>.h5.  The initial factors are GIVEN (W_init / H_init, model.jl:71-72), so
# Julia's RNG never enters; what the files pin is the arithmetic of mult.jl:23-58, hals.jl:31-154, pgd.jl:158-255 on this
# repository's own inputs -- exactly what oracle/cmf_oracle.{py,c} restate.
#
# Requirements of the checkout's environment (CMF.jl's Project.toml): HDF5, FFTW, JLD, PyPlot, Combinatorics.  `import PyPlot`
# at src/CMF.jl:10 needs a matplotlib; on a headless machine `ENV["MPLBACKEND"] = "Agg"` (set below) is enough.
# HEAD uses the five-argument mul! (src/common.jl:77,112): Julia >= 1.3.

using HDF5

ENV["MPLBACKEND"] = get(ENV, "MPLBACKEND", "Agg")

length(ARGS) >= 1 || error("usage: julia tools/reference_fixtures.jl /path/to/CMF.jl [fixture names...]")
const REF = abspath(ARGS[1])
isfile(joinpath(REF, "src", "CMF.jl")) || error("$REF does not look like a checkout of CMF.jl (src/CMF.jl is missing)")
include(joinpath(REF, "src", "CMF.jl"))   # defines module CMF from the reference's sources, as they lie in the checkout

const GOLDEN = joinpath(dirname(@__DIR__), "tests", "golden")
const ALL = ["mu_small", "mu_small_reg", "mu_k5", "hals_small", "pgd_small"]
const RULES = Dict("mult" => CMF.MultUpdate, "hals" => CMF.HALSUpdate, "pgd" => CMF.PGDUpdate)

scalar(x) = x isa AbstractArray ? first(x) : x   # (a scalar dataset reads as a number, a 1-element one as an array)

function run_fixture(name)
    inp = joinpath(GOLDEN, "ref_inputs", name * ".h5")
    isfile(inp) || error("$inp is missing: run `python tests/golden/export_reference_inputs.py` first")
    data = Matrix{Float64}(h5read(inp, "data"))
    W0 = Array{Float64,3}(h5read(inp, "W0"))
    H0 = Matrix{Float64}(h5read(inp, "H0"))
    max_itr = Int(scalar(h5read(inp, "max_itr")))
    reg = Dict(k => Float64(scalar(h5read(inp, String(k)))) for k in (:l1W, :l2W, :l1H, :l2H))
    rule = String(scalar(h5read(inp, "rule")))
    K, N, L = size(W0)
    size(data) == (N, size(H0, 2)) || error("$name: data is $(size(data)), W0 $(size(W0)), H0 $(size(H0))")

    # the reference's entry point with its own defaults (PGD: SquareLoss, NonnegConstraint, penaltiesW=[SquarePenalty(1)],
    # penaltiesH=[], pgd.jl:158-202 -- the l* keywords fall into kwargs... there and are ignored, as the oracle's fit_pgd assumes)
    results = CMF.fit_cnmf(data; L=L, K=K, alg=RULES[rule], max_itr=max_itr, W_init=W0, H_init=H0,
                           check_convergence=false, l1W=reg[:l1W], l2W=reg[:l2W], l1H=reg[:l1H], l2H=reg[:l2H])
    length(results.loss_hist) == max_itr + 1 || error("$name: loss_hist has $(length(results.loss_hist)) entries, expected $(max_itr + 1)")

    out = joinpath(GOLDEN, "ref_" * name * ".h5")
    h5open(out, "w") do f
        write(f, "W", results.W)                    # K x N x L
        write(f, "H", results.H)                    # K x T
        write(f, "loss_hist", Vector{Float64}(results.loss_hist))
        write(f, "rule", rule)
        write(f, "julia_version", string(VERSION))
        write(f, "reference", REF)
    end
    println("$name ($rule, $N x $(size(data, 2)), K=$K, L=$L, $max_itr iterations): loss ",
            results.loss_hist[1], " -> ", results.loss_hist[end], "  =>  ", out)
end

for name in (length(ARGS) >= 2 ? ARGS[2:end] : ALL)
    run_fixture(name)
end
