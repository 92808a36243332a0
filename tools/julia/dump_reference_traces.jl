# dump_reference_traces.jl -- pins oracle/mpc.py to the REFERENCE the day a Julia box exists (SURVEY.md 8c, "caveat to
# carry forward"; VERDICT r4 next #7).  NOT run in the build container or on the GPU box (no Julia there): run it once
# in an environment with Julia >= 1.10 and MadIPM's test dependencies,
#
#     julia --project=/path/to/MadIPM/test tools/julia/dump_reference_traces.jl /path/to/repo/tests/golden
#
# and commit the two files it writes; tests/test_reference_pin.py then compares oracle/mpc.py against them (it is skipped,
# saying why, while they do not exist).
#
# What it runs -- the two inputs of test/runtests.jl that matter for the hot path:
#   * simple_lp()                        (test/runtests.jl:24-55: fully specified in the reference, no RNG)
#   * MadNLPTests.DenseDummyQP(zeros(10); m=5)   (test/runtests.jl:9,60: Julia's RNG -- hence the MATRICES are dumped too)
# through MadIPM.MPCSolver with MadNLP's default K2 system (SparseKKTSystem) and LapackCPUSolver, default options
# (src/utils.jl:69-103).  Per iteration k the tuple MadNLP.print_iter shows (src/structure.jl:178-195) plus inf_compl and mu:
# obtained by solving with max_iter = k for k = 0, 1, .. -- mpc! (src/solver.jl:254-345) returns from the head of
# iteration k with every field of that iterate in place, so no reference source needs a hook.
using JSON, LinearAlgebra, SparseArrays
using MadNLP, MadIPM, MadNLPTests, QuadraticModels, NLPModels

function simple_lp()  # the data of test/runtests.jl:24-55
    return QuadraticModel(ones(2), Int[], Int[], Float64[]; Arows = [1, 1], Acols = [1, 2], Avals = [1.0; 1.0],
                          lcon = [1.0], ucon = [1.0], lvar = [0.0; 0.0], uvar = [Inf; Inf], c0 = 0.0, x0 = ones(2),
                          name = "simpleLP")
end

# dense data of an NLPModels QP: f(x) = c0 + q'x + x'Hx/2, lcon <= A x <= ucon, lvar <= x <= uvar
function dense_data(nlp)
    n, m = NLPModels.get_nvar(nlp), NLPModels.get_ncon(nlp)
    z = zeros(n)
    hr, hc = NLPModels.hess_structure(nlp)
    hv = NLPModels.hess_coord(nlp, z)
    H = zeros(n, n)
    for (i, j, v) in zip(hr, hc, hv)   # lower triangle -> both
        H[i, j] += v
        i != j && (H[j, i] += v)
    end
    A = zeros(m, n)
    if m > 0
        jr, jc = NLPModels.jac_structure(nlp)
        jv = NLPModels.jac_coord(nlp, z)
        for (i, j, v) in zip(jr, jc, jv)
            A[i, j] += v
        end
    end
    inf2s(v) = [isfinite(x) ? x : (x > 0 ? "inf" : "-inf") for x in v]
    return Dict("n" => n, "m" => m, "H" => [H[i, :] for i in 1:n], "A" => [A[i, :] for i in 1:m],
                "q" => NLPModels.grad(nlp, z), "c0" => NLPModels.obj(nlp, z),
                "lvar" => inf2s(NLPModels.get_lvar(nlp)), "uvar" => inf2s(NLPModels.get_uvar(nlp)),
                "lcon" => inf2s(NLPModels.get_lcon(nlp)), "ucon" => inf2s(NLPModels.get_ucon(nlp)),
                "x0" => NLPModels.get_x0(nlp))
end

function run_to(nlp, k; opts...)
    solver = MadIPM.MPCSolver(nlp; print_level = MadNLP.ERROR, linear_solver = MadNLP.LapackCPUSolver, max_iter = k, opts...)
    stats = MadIPM.solve!(solver)
    return solver, stats
end

function dump_case(name, nlp, path; opts...)
    full, stats = run_to(nlp, 3000; opts...)
    iters = full.cnt.k
    trace = []
    for k in 0:iters
        s, _ = run_to(nlp, k; opts...)
        @assert s.cnt.k == k
        push!(trace, Dict("k" => k, "obj" => s.obj_val / s.cb.obj_scale[], "inf_pr" => s.inf_pr, "inf_du" => s.inf_du,
                          "inf_compl" => s.inf_compl, "mu" => s.mu,
                          "dnorm" => k == 0 ? 0.0 : norm(MadNLP.primal(s.d), Inf), "del_w" => s.del_w,
                          "alpha_d" => s.alpha_d, "alpha_p" => s.alpha_p))
    end
    out = Dict("case" => name, "reference" => "MadIPM (MadQP.jl) snapshot 2025-06-14, MadNLP " * string(pkgversion(MadNLP)),
               "kkt_system" => "SparseKKTSystem (K2)", "linear_solver" => "LapackCPUSolver",
               "options" => Dict(string(k) => string(v) for (k, v) in opts),
               "data" => dense_data(nlp), "status" => Int(stats.status), "iter" => iters,
               "objective" => stats.objective, "solution" => stats.solution, "constraints" => stats.constraints,
               "multipliers" => stats.multipliers, "multipliers_L" => stats.multipliers_L,
               "multipliers_U" => stats.multipliers_U, "trace" => trace)
    open(path, "w") do io
        JSON.print(io, out)
    end
    println("wrote ", path, ": ", iters, " iterations, objective ", stats.objective)
end

outdir = length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "..", "..", "tests", "golden")
dump_case("simple_lp", simple_lp(), joinpath(outdir, "reference_simple_lp.json"))
dump_case("dense_dummy_qp_n10_m5", MadNLPTests.DenseDummyQP(zeros(10); m = 5), joinpath(outdir, "reference_dense_dummy_qp_n10_m5.json"))
dump_case("dense_dummy_qp_n10_m5_ncorr5", MadNLPTests.DenseDummyQP(zeros(10); m = 5),
          joinpath(outdir, "reference_dense_dummy_qp_n10_m5_ncorr5.json"); max_ncorr = 5)
