# MadQPHIP.jl -- Julia glue that drops libmadqp_hip.so into MadIPM's solver loop.
#
# STATUS: UNVERIFIED UNDER JULIA.  No Julia toolchain exists in the authoring or GPU image (SURVEY.md section 0), so
# this file has never been executed.  What IS executed: every ccall below binds one prototype of include/madqp.h,
# and tests/test_gpu_julia_replay.py replays -- through ctypes, entry point by entry point, in the order MadIPM
# issues them (src/solver.jl:6-125,127-182,254-345) -- exactly the call sequence of the methods of this file, each
# replay function named after the Julia method it stands for.  A change here needs the same change there.
#
# What it provides (SURVEY.md 8b)
#   HIPCondensedKKTSystem / HIPAugmentedKKTSystem / HIPNormalKKTSystem  <: MadNLP.AbstractKKTSystem
#       (modelled on NormalKKTSystem, src/KKT/normalkkt.jl; same generic fields, same methods)
#   HIPCholeskySolver <: MadNLP.AbstractLinearSolver   (ctor `Solver(aug_com; opt)`, factorize!, solve!(s, rhs))
#   HIPDistributedKKTSystem / HIPDistributedCholeskySolver: the condensed system of ONE QP on a P x Q grid of GPUs,
#       one process per GPU (madqp_dist_* / madqp_dkkt_*, RCCL issued by the library)
#   methods of MadIPM's per-variable kernels (src/kernels.jl) specialised on these KKT types, so that `mpc!`
#   (src/solver.jl:254-345) runs unchanged with every vector on the device.
#
# Usage
#   using MadNLP, MadIPM, AMDGPU; include("MadQPHIP.jl"); using .MadQPHIP
#   solver = MadIPM.MPCSolver(qp_on_rocarrays; kkt_system = MadQPHIP.HIPCondensedKKTSystem,
#                             linear_solver = MadQPHIP.HIPCholeskySolver,
#                             regularization = MadIPM.FixedRegularization(1e-8, -1e-8))
#   MadIPM.solve!(solver)
module MadQPHIP

import MadNLP
import MadIPM
using LinearAlgebra
import LinearAlgebra: mul!

const libmadqp = get(ENV, "MADQP_HIP_LIB", "libmadqp_hip.so")

# --------------------------------------------------------------------------- low level
mutable struct Context
    ptr::Ptr{Cvoid}
    function Context(device::Integer = 0, stream::Ptr{Cvoid} = C_NULL)
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:madqp_ctx_create, libmadqp), Int32, (Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, stream, ref)
        rc == 0 || error("madqp_ctx_create failed ($rc)")
        ctx = new(ref[])
        finalizer(c -> ccall((:madqp_ctx_destroy, libmadqp), Int32, (Ptr{Cvoid},), c.ptr), ctx)
        return ctx
    end
end

last_error(ctx::Context) = unsafe_string(ccall((:madqp_last_error, libmadqp), Cstring, (Ptr{Cvoid},), ctx.ptr))

function check(ctx::Context, rc::Int32)
    rc == 0 && return
    rc < 0 && error("libmadqp_hip: $(last_error(ctx)) ($rc)")     # usage / device fault
    throw(MadNLP.SolveException())        # rc > 0: numerical condition (src/linear_solver.jl:41-43)
end

# madqp_state of include/madqp.h: device-pointer view of MPCSolver + the KKT diagonals.
struct CState
    n::Int64; m::Int64; nlb::Int64; nub::Int64
    ind_lb::Ptr{Int64}; ind_ub::Ptr{Int64}
    x::Ptr{Float64}; xl::Ptr{Float64}; xu::Ptr{Float64}; zl::Ptr{Float64}; zu::Ptr{Float64}; f::Ptr{Float64}
    y::Ptr{Float64}; c::Ptr{Float64}; jacl::Ptr{Float64}
    d::Ptr{Float64}; p::Ptr{Float64}
    correction_lb::Ptr{Float64}; correction_ub::Ptr{Float64}
    reg::Ptr{Float64}; pr_diag::Ptr{Float64}; du_diag::Ptr{Float64}
    l_diag::Ptr{Float64}; l_lower::Ptr{Float64}; u_diag::Ptr{Float64}; u_lower::Ptr{Float64}
end

dptr(v) = Base.unsafe_convert(Ptr{eltype(v)}, v)     # ROCArray -> device pointer
const NULLF = Ptr{Float64}(C_NULL)

# --------------------------------------------------------------------------- linear solver
# `aug_com`: the matrix object the linear-solver constructor receives once and re-reads at every factorize!
# (src/KKT/normalkkt.jl:97-101).  Here the matrix (K, later its factor) lives in the library's KKT object; this is the
# handle to it.
mutable struct HIPDenseKKTMatrix{T}
    handle::Ptr{Cvoid}        # madqp_kkt*
    ctx::Context
    order::Int
    maps::Vector{Ptr{Cvoid}}  # madqp_coo_map handles that live as long as the KKT object
    function HIPDenseKKTMatrix{T}(handle, ctx, order, maps) where {T}
        A = new{T}(handle, ctx, order, maps)
        finalizer(A) do a       # ownership: the library frees what it allocated (SURVEY.md 8b)
            ccall((:madqp_kkt_destroy, libmadqp), Int32, (Ptr{Cvoid},), a.handle)
            foreach(mp -> ccall((:madqp_coo_map_destroy, libmadqp), Int32, (Ptr{Cvoid},), mp), a.maps)
        end
        return A
    end
end
Base.size(A::HIPDenseKKTMatrix) = (A.order, A.order)
Base.eltype(::HIPDenseKKTMatrix{T}) where {T} = T

MadNLP.@kwdef mutable struct HIPCholeskyOptions <: MadNLP.AbstractOptions end

mutable struct HIPCholeskySolver{T} <: MadNLP.AbstractLinearSolver{T}
    aug_com::HIPDenseKKTMatrix{T}
    chol::Ptr{Cvoid}          # madqp_chol* of the KKT object (borrowed)
    info::Int32
    opt::HIPCholeskyOptions
    logger::MadNLP.MadNLPLogger
end

# src/KKT/normalkkt.jl:99-101: `linear_solver(aug_com; opt = opt_linear_solver)`
function HIPCholeskySolver(aug_com::HIPDenseKKTMatrix{T}; opt = HIPCholeskyOptions(),
                           logger = MadNLP.MadNLPLogger()) where {T}
    chol = Ref{Ptr{Cvoid}}(C_NULL)
    order = Ref{Int64}(0)
    check(aug_com.ctx, ccall((:madqp_kkt_chol, libmadqp), Int32, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ref{Int64}),
                             aug_com.handle, chol, order))
    return HIPCholeskySolver{T}(aug_com, chol[], Int32(0), opt, logger)
end

MadNLP.default_options(::Type{HIPCholeskySolver}) = HIPCholeskyOptions()
MadNLP.introduce(::HIPCholeskySolver) = "madqp-hip blocked left-looking fp64 Cholesky (MFMA, gfx950)"
MadNLP.is_supported(::Type{HIPCholeskySolver}, ::Type{Float64}) = true
MadNLP.is_inertia(::HIPCholeskySolver) = true
MadNLP.inertia(s::HIPCholeskySolver) = s.info == 0 ? (s.aug_com.order, 0, 0) : (Int(s.info) - 1, 0, 1)
MadNLP.improve!(::HIPCholeskySolver) = false
MadIPM.is_factorized(s::HIPCholeskySolver) = s.info == 0        # src/utils.jl:54-62

function MadNLP.factorize!(s::HIPCholeskySolver)                 # via factorize_wrapper!, src/linear_solver.jl:10
    info = Ref{Int32}(0)
    check(s.aug_com.ctx, ccall((:madqp_kkt_factorize, libmadqp), Int32, (Ptr{Cvoid}, Ref{Int32}), s.aug_com.handle, info))
    s.info = info[]              # > 0: not positive definite -> x100 regularization retry (src/linear_solver.jl:11-15)
    return s
end

# src/KKT/normalkkt.jl:196: `MadNLP.solve!(kkt.linear_solver, r2)`, in place
function MadNLP.solve!(s::HIPCholeskySolver, rhs::AbstractVector)
    check(s.aug_com.ctx, ccall((:madqp_chol_solve, libmadqp), Int32, (Ptr{Cvoid}, Ptr{Float64}), s.chol, dptr(rhs)))
    return rhs
end

# --------------------------------------------------------------------------- KKT system
# F = :condensed  K = H + Sigma_x + A' Theta A            (madqp_kkt_create)
#     :augmented  [H + Sigma_x, A'; A, -D], L diag(I,-I) L' (madqp_kkt_create_augmented; MadNLP's default K2 form)
#     :scaled_augmented  the same matrix scaled symmetrically (madqp_kkt_create_scaled_augmented; MadNLP's K2.5,
#                 ScaledSparseKKTSystem: src/kernels.jl:149-165, test/runtests.jl:95-115)
#     :normal     A Sigma^-1 A', LP only                  (madqp_kkt_create_normal; the reference's NormalKKTSystem)
struct HIPKKTSystem{T, VT, MT, QN, VI, LS, F} <: MadNLP.AbstractKKTSystem{T, VT, MT, QN}
    aug_com::HIPDenseKKTMatrix{T}
    handle::Ptr{Cvoid}
    ctx::Context
    H::MT                 # nx x nx dense symmetric (device); 0 x 0 for an LP
    At::MT                # dense Jacobian operand (device): nx x m Julia matrix (= A with row k contiguous) for the
                          # condensed / augmented forms, m x nx Julia matrix (= A' with variable k contiguous) for :normal
    lda::Int              # its leading dimension as the library sees it (nx, or m for :normal)
    jac::VT               # nnzj callback buffer (get_jacobian), COO order of the model's pattern
    hess::VT              # nnzh callback buffer (get_hessian)
    jac_map::Ptr{Cvoid}   # madqp_coo_map: jac -> At
    hess_map::Ptr{Cvoid}  # madqp_coo_map: hess -> H (symmetric)
    # fields MadIPM reads generically (src/kernels.jl:135-144, src/solver.jl:16-18)
    reg::VT; pr_diag::VT; du_diag::VT
    l_diag::VT; u_diag::VT; l_lower::VT; u_lower::VT
    linear_solver::LS
    ind_ineq::VI; ind_lb::VI; ind_ub::VI      # 1-based, as MadNLP keeps them
    ind_lb0::VI; ind_ub0::VI                  # 0-based device copies handed to the library
    cstate::Base.RefValue{CState}             # view with the KKT's own fields only (build_kkt!, solve!, mul!)
    n::Int; m::Int; nx::Int
end
const HIPCondensedKKTSystem = HIPKKTSystem{T, VT, MT, QN, VI, LS, :condensed} where {T, VT, MT, QN, VI, LS}
const HIPAugmentedKKTSystem = HIPKKTSystem{T, VT, MT, QN, VI, LS, :augmented} where {T, VT, MT, QN, VI, LS}
const HIPScaledAugmentedKKTSystem = HIPKKTSystem{T, VT, MT, QN, VI, LS, :scaled_augmented} where {T, VT, MT, QN, VI, LS}
const HIPNormalKKTSystem = HIPKKTSystem{T, VT, MT, QN, VI, LS, :normal} where {T, VT, MT, QN, VI, LS}
form(::HIPKKTSystem{T, VT, MT, QN, VI, LS, F}) where {T, VT, MT, QN, VI, LS, F} = F

function coo_map(ctx::Context, I::Vector{Int32}, J::Vector{Int32}, nrows, ncols, symmetric)
    ref = Ref{Ptr{Cvoid}}(C_NULL)
    check(ctx, ccall((:madqp_coo_map_create, libmadqp), Int32,
                     (Ptr{Cvoid}, Int64, Ptr{Int32}, Ptr{Int32}, Int64, Int64, Int32, Ref{Ptr{Cvoid}}),
                     ctx.ptr, length(I), I, J, nrows, ncols, symmetric, ref))
    return ref[]
end

function _create(F::Symbol, cb::MadNLP.SparseCallback{T, VT}, ind_cons, linear_solver::Type, opt_linear_solver) where {T, VT}
    nx, m = cb.nvar, cb.ncon
    ind_ineq = ind_cons.ind_ineq
    ns = length(ind_ineq)
    n = nx + ns
    nlb, nub = length(ind_cons.ind_lb), length(ind_cons.ind_ub)
    if F == :normal && cb.nnzh > 0                                    # src/KKT/normalkkt.jl:45-48
        error("The KKT system NormalKKTSystem supports only linear programs.")
    end
    ctx = Context()
    # sparsity patterns of the callbacks (src/KKT/normalkkt.jl:51-53)
    jI = MadNLP.create_array(cb, Int32, cb.nnzj); jJ = MadNLP.create_array(cb, Int32, cb.nnzj)
    MadNLP._jac_sparsity_wrapper!(cb, jI, jJ)
    hI = MadNLP.create_array(cb, Int32, cb.nnzh); hJ = MadNLP.create_array(cb, Int32, cb.nnzh)
    cb.nnzh > 0 && MadNLP._hess_sparsity_wrapper!(cb, hI, hJ)
    jIh, jJh = Vector{Int32}(Array(jI)), Vector{Int32}(Array(jJ))
    # target of the Jacobian values: entry (i, k) of A at [i*nx + k] (rows of A contiguous), or, for the normal
    # equations, at [k*m + i] (rows of A' contiguous: the pattern is handed over transposed)
    jac_map = F == :normal ? coo_map(ctx, jJh, jIh, nx, m, 0) : coo_map(ctx, jIh, jJh, m, nx, 0)
    hess_map = cb.nnzh > 0 ? coo_map(ctx, Vector{Int32}(Array(hI)), Vector{Int32}(Array(hJ)), nx, nx, 1) : C_NULL
    nh = cb.nnzh > 0 ? nx : 0
    H = reshape(fill!(VT(undef, nh * nh), zero(T)), nh, nh)
    At = F == :normal ? reshape(fill!(VT(undef, nx * m), zero(T)), m, nx) :
                        reshape(fill!(VT(undef, nx * m), zero(T)), nx, m)
    lda = F == :normal ? max(m, 1) : max(nx, 1)
    mk(k) = VT(undef, k)
    ref = Ref{Ptr{Cvoid}}(C_NULL)
    ineq0 = Int64.(Array(ind_ineq)) .- 1
    Hp = nh > 0 ? dptr(H) : NULLF
    if F == :condensed
        rc = ccall((:madqp_kkt_create, libmadqp), Int32,
                   (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Int64}, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ref{Ptr{Cvoid}}),
                   ctx.ptr, nx, m, ns, ineq0, Hp, max(nx, 1), dptr(At), lda, ref)
    elseif F == :augmented
        rc = ccall((:madqp_kkt_create_augmented, libmadqp), Int32,
                   (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Int64}, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ref{Ptr{Cvoid}}),
                   ctx.ptr, nx, m, ns, ineq0, Hp, max(nx, 1), dptr(At), lda, ref)
    elseif F == :scaled_augmented
        rc = ccall((:madqp_kkt_create_scaled_augmented, libmadqp), Int32,
                   (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Int64}, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ref{Ptr{Cvoid}}),
                   ctx.ptr, nx, m, ns, ineq0, Hp, max(nx, 1), dptr(At), lda, ref)
    else   # A' with variable k contiguous = a Julia m x nx matrix as it stands
        rc = ccall((:madqp_kkt_create_normal, libmadqp), Int32,
                   (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Int64}, Ptr{Float64}, Int64, Ref{Ptr{Cvoid}}),
                   ctx.ptr, nx, m, ns, ineq0, dptr(At), lda, ref)
    end
    check(ctx, rc)
    # MadIPM's solve_system! (src/linear_solver.jl:19-45) calls solve!(kkt, d) once and only looks at the residual: the
    # refinement step that small ill-conditioned problems need (DESIGN.md section 4.2) therefore runs INSIDE madqp_kkt_solve --
    # the AUTO rule (-1): one step while the factorised matrix has order <= 1024, none above.  ENV["MADQP_KKT_REFINE"]
    # overrides ("0" = the plain solve! of src/KKT/normalkkt.jl:182-205).
    check(ctx, ccall((:madqp_kkt_set_refine, libmadqp), Int32, (Ptr{Cvoid}, Int32), ref[],
                     parse(Int32, get(ENV, "MADQP_KKT_REFINE", "-1"))))
    order = (F == :augmented || F == :scaled_augmented) ? (cld(nx, 128) * 128 + m) : (F == :normal ? m : nx)
    aug_com = HIPDenseKKTMatrix{T}(ref[], ctx, order, filter(!=(C_NULL), [jac_map, hess_map]))
    ls = linear_solver(aug_com; opt = opt_linear_solver)             # src/KKT/normalkkt.jl:99-101
    reg, pr_diag, du_diag = mk(n), mk(n), mk(m)
    l_diag, u_diag, l_lower, u_lower = mk(nlb), mk(nub), mk(nlb), mk(nub)
    ind_lb0, ind_ub0 = ind_cons.ind_lb .- 1, ind_cons.ind_ub .- 1
    # the KKT's own view: everything build_kkt! / solve! / mul! read lives in the KKT object, so these three work
    # from the first factorize_wrapper! on (src/solver.jl:16-21 calls it BEFORE any set_aug_diagonal_reg!)
    cs = CState(n, m, nlb, nub, dptr(ind_lb0), dptr(ind_ub0),
                NULLF, NULLF, NULLF, NULLF, NULLF, NULLF, NULLF, NULLF, NULLF, NULLF, NULLF, NULLF, NULLF,
                dptr(reg), dptr(pr_diag), dptr(du_diag), dptr(l_diag), dptr(l_lower), dptr(u_diag), dptr(u_lower))
    VI = typeof(ind_cons.ind_lb)
    kkt = HIPKKTSystem{T, VT, typeof(H), MadNLP.ExactHessian{T, VT}, VI, typeof(ls), F}(
        aug_com, ref[], ctx, H, At, lda, mk(cb.nnzj), mk(cb.nnzh), jac_map, hess_map,
        reg, pr_diag, du_diag, l_diag, u_diag, l_lower, u_lower, ls,
        ind_ineq, ind_cons.ind_lb, ind_cons.ind_ub, ind_lb0, ind_ub0, Ref(cs), n, m, nx)
    return kkt
end

for (TY, F) in ((:HIPCondensedKKTSystem, :condensed), (:HIPAugmentedKKTSystem, :augmented),
                (:HIPScaledAugmentedKKTSystem, :scaled_augmented), (:HIPNormalKKTSystem, :normal))
    @eval function MadNLP.create_kkt_system(
        ::Type{$TY}, cb::MadNLP.SparseCallback{T, VT}, ind_cons, linear_solver::Type;
        opt_linear_solver = MadNLP.default_options(linear_solver),
        hessian_approximation = MadNLP.ExactHessian, qn_options = MadNLP.QuasiNewtonOptions(),
    ) where {T, VT}
        return _create($(QuoteNode(F)), cb, ind_cons, linear_solver, opt_linear_solver)
    end
end

MadNLP.num_variables(kkt::HIPKKTSystem) = kkt.n                 # src/KKT/normalkkt.jl:128
MadNLP.get_jacobian(kkt::HIPKKTSystem) = kkt.jac                # :129 -- the nnzj buffer SparseCallback fills
MadNLP.get_hessian(kkt::HIPKKTSystem) = kkt.hess                # :130
function MadNLP.is_inertia_correct(kkt::HIPKKTSystem, num_pos, num_zero, num_neg)   # :132-134
    form(kkt) in (:augmented, :scaled_augmented) && return (num_zero == 0) && (num_neg == kkt.m)
    return (num_zero == 0) && (num_pos == kkt.aug_com.order)
end

function MadNLP.initialize!(kkt::HIPKKTSystem{T}) where {T}      # src/KKT/normalkkt.jl:136-147 (+ K2.5's scaling factor)
    check(kkt.ctx, ccall((:madqp_kkt_initialize, libmadqp), Int32, (Ptr{Cvoid}, Ref{CState}), kkt.handle, kkt.cstate))
    return
end

# src/KKT/normalkkt.jl:149-158: callback values (COO order) -> the matrix storage.  The slack columns (-1) are implicit
# in the library (ind_ineq), so only the nnzj model entries travel.
function MadNLP.compress_jacobian!(kkt::HIPKKTSystem)
    check(kkt.ctx, ccall((:madqp_coo_map_apply, libmadqp), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64),
                         kkt.jac_map, dptr(kkt.jac), dptr(kkt.At), kkt.lda))
    return
end
function MadNLP.compress_hessian!(kkt::HIPKKTSystem)
    kkt.hess_map == C_NULL && return
    check(kkt.ctx, ccall((:madqp_coo_map_apply, libmadqp), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64),
                         kkt.hess_map, dptr(kkt.hess), dptr(kkt.H), max(kkt.nx, 1)))
    return
end

function MadNLP.jtprod!(y::AbstractVector, kkt::HIPKKTSystem, x::AbstractVector)   # :162-164
    check(kkt.ctx, ccall((:madqp_kkt_jtprod, libmadqp), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
                         kkt.handle, dptr(y), dptr(x)))
    return y
end

function MadNLP.build_kkt!(kkt::HIPKKTSystem)                  # src/KKT/normalkkt.jl:166-180
    check(kkt.ctx, ccall((:madqp_kkt_build, libmadqp), Int32, (Ptr{Cvoid}, Ref{CState}), kkt.handle, kkt.cstate))
    return
end

function MadNLP.solve!(kkt::HIPKKTSystem, w::MadNLP.AbstractKKTVector)   # :182-205
    check(kkt.ctx, ccall((:madqp_kkt_solve, libmadqp), Int32, (Ptr{Cvoid}, Ref{CState}, Ptr{Float64}),
                         kkt.handle, kkt.cstate, dptr(MadNLP.full(w))))
    return w
end

function mul!(w::MadNLP.AbstractKKTVector{T}, kkt::HIPKKTSystem, v::MadNLP.AbstractKKTVector,
              alpha = one(T), beta = zero(T)) where {T}           # :207-219
    check(kkt.ctx, ccall((:madqp_kkt_mul, libmadqp), Int32,
                         (Ptr{Cvoid}, Ref{CState}, Ptr{Float64}, Ptr{Float64}, Float64, Float64),
                         kkt.handle, kkt.cstate, dptr(MadNLP.full(w)), dptr(MadNLP.full(v)), alpha, beta))
    return w
end

# --------------------------------------------------------------------------- one QP over a P x Q grid of GPUs
# SURVEY.md 8e / BASELINE configs[4]: every rank runs the SAME MadIPM loop on replicated vectors (identical branches:
# every scalar is computed from bitwise identical vectors); only this KKT system knows that K = H + Sigma_x + A' Theta A,
# its Cholesky factor and the products with H and A are spread over the grid (madqp_dist_*, madqp_dkkt_* of
# include/madqp.h; the collectives -- RCCL over xGMI -- are issued by the library).  One process per GPU; the 128 bytes
# of RCCL's unique id travel from rank 0 over whatever the launcher offers (`share_id`, e.g. `id -> MPI.Bcast!(id, 0,
# comm)`), once.
#
#   MadQPHIP.configure_distributed!(rank, world; P, Q, nb = 1024, share_id = id -> MPI.Bcast!(id, 0, comm))
#   solver = MadIPM.MPCSolver(qp; kkt_system = MadQPHIP.HIPDistributedKKTSystem,
#                             linear_solver = MadQPHIP.HIPDistributedCholeskySolver, ...)
#
# The model callbacks are evaluated on every rank as in a single-process run (the COO buffers `jac`, `hess` are
# replicated: nnz values); compress_jacobian! / compress_hessian! keep only this rank's pieces -- the columns of A of its
# tile rows (A_I) and of its tile columns (A_J), its lower tiles of H (madqp_coo_map_create_cols_cyclic /
# _tiles_cyclic): 1/(PQ) of H and of K, (1/P + 1/Q) of A per rank.  The factorisation additionally stores the operands
# of its lazy updates -- this rank's tile rows and tile columns of L, (1/P + 1/Q) n^2/2 doubles: the factor is replicated
# Q-fold along process rows and P-fold along process columns (madqp_dist_memory reports the bytes; C5 on 2 x 4: 84 GB).
Base.@kwdef mutable struct DistributedConfig
    rank::Int = 0
    world::Int = 1
    P::Int = 1
    Q::Int = 1
    nb::Int = 1024
    share_id::Function = identity     # in-place broadcast of a Vector{UInt8}(128) from rank 0 to all ranks
end
const DIST_CONFIG = Ref(DistributedConfig())
function configure_distributed!(rank, world; P, Q, nb = 1024, share_id = identity)
    P * Q == world || error("P * Q must equal the number of ranks")
    DIST_CONFIG[] = DistributedConfig(rank, world, P, Q, nb, share_id)
end

mutable struct HIPDistributedMatrix{T}      # aug_com of the distributed system: the grid + the KKT object on it
    dist::Ptr{Cvoid}          # madqp_dist*
    dkkt::Ptr{Cvoid}          # madqp_dkkt* (set once the operands exist)
    ctx::Context
    order::Int
    maps::Vector{Ptr{Cvoid}}
    function HIPDistributedMatrix{T}(dist, ctx, order) where {T}
        A = new{T}(dist, C_NULL, ctx, order, Ptr{Cvoid}[])
        finalizer(A) do a
            a.dkkt != C_NULL && ccall((:madqp_dkkt_destroy, libmadqp), Int32, (Ptr{Cvoid},), a.dkkt)
            ccall((:madqp_dist_destroy, libmadqp), Int32, (Ptr{Cvoid},), a.dist)
            foreach(mp -> ccall((:madqp_coo_map_destroy, libmadqp), Int32, (Ptr{Cvoid},), mp), a.maps)
        end
        return A
    end
end

mutable struct HIPDistributedCholeskySolver{T} <: MadNLP.AbstractLinearSolver{T}
    aug_com::HIPDistributedMatrix{T}
    info::Int32
    opt::HIPCholeskyOptions
    logger::MadNLP.MadNLPLogger
end
HIPDistributedCholeskySolver(aug_com::HIPDistributedMatrix{T}; opt = HIPCholeskyOptions(),
                             logger = MadNLP.MadNLPLogger()) where {T} =
    HIPDistributedCholeskySolver{T}(aug_com, Int32(0), opt, logger)
MadNLP.default_options(::Type{HIPDistributedCholeskySolver}) = HIPCholeskyOptions()
MadNLP.introduce(::HIPDistributedCholeskySolver) = "madqp-hip 2-D block-cyclic distributed fp64 Cholesky (MFMA + RCCL, gfx950)"
MadNLP.is_supported(::Type{HIPDistributedCholeskySolver}, ::Type{Float64}) = true
MadNLP.is_inertia(::HIPDistributedCholeskySolver) = true
MadNLP.inertia(s::HIPDistributedCholeskySolver) = s.info == 0 ? (s.aug_com.order, 0, 0) : (Int(s.info) - 1, 0, 1)
MadNLP.improve!(::HIPDistributedCholeskySolver) = false
MadIPM.is_factorized(s::HIPDistributedCholeskySolver) = s.info == 0
function MadNLP.factorize!(s::HIPDistributedCholeskySolver)    # info is identical on every rank (dist_core.inc)
    info = Ref{Int32}(0)
    check(s.aug_com.ctx, ccall((:madqp_dkkt_factorize, libmadqp), Int32, (Ptr{Cvoid}, Ref{Int32}), s.aug_com.dkkt, info))
    s.info = info[]
    return s
end
function MadNLP.solve!(s::HIPDistributedCholeskySolver, rhs::AbstractVector)   # replicated right-hand side, in place
    check(s.aug_com.ctx, ccall((:madqp_dist_solve, libmadqp), Int32, (Ptr{Cvoid}, Ptr{Float64}), s.aug_com.dist, dptr(rhs)))
    return rhs
end

struct HIPDistributedKKTSystem{T, VT, MT, QN, VI, LS} <: MadNLP.AbstractKKTSystem{T, VT, MT, QN}
    aug_com::HIPDistributedMatrix{T}
    ctx::Context
    Hloc::MT; A_I::MT; A_J::MT          # this rank's pieces (device), layouts of madqp_dkkt_create
    ld::Int; ncp::Int                   # leading dimension / padded column count of the local matrix
    jac::VT; hess::VT                   # replicated callback buffers
    jacI_map::Ptr{Cvoid}; jacJ_map::Ptr{Cvoid}; hess_map::Ptr{Cvoid}
    reg::VT; pr_diag::VT; du_diag::VT
    l_diag::VT; u_diag::VT; l_lower::VT; u_lower::VT
    linear_solver::LS
    ind_ineq::VI; ind_lb::VI; ind_ub::VI
    ind_lb0::VI; ind_ub0::VI
    cstate::Base.RefValue{CState}
    n::Int; m::Int; nx::Int
end

function MadNLP.create_kkt_system(
    ::Type{HIPDistributedKKTSystem}, cb::MadNLP.SparseCallback{T, VT}, ind_cons, linear_solver::Type;
    opt_linear_solver = MadNLP.default_options(linear_solver),
    hessian_approximation = MadNLP.ExactHessian, qn_options = MadNLP.QuasiNewtonOptions(),
) where {T, VT}
    cfg = DIST_CONFIG[]
    nx, m = cb.nvar, cb.ncon
    ind_ineq = ind_cons.ind_ineq
    ns = length(ind_ineq); n = nx + ns
    nlb, nub = length(ind_cons.ind_lb), length(ind_cons.ind_ub)
    ctx = Context()
    id = zeros(UInt8, 128)                       # RCCL's unique id: drawn on rank 0, shipped by the launcher's channel
    if cfg.world > 1
        cfg.rank == 0 && check(ctx, ccall((:madqp_dist_unique_id, libmadqp), Int32, (Ptr{Cvoid}, Ptr{UInt8}), ctx.ptr, id))
        cfg.share_id(id)
    end
    dref = Ref{Ptr{Cvoid}}(C_NULL)
    check(ctx, ccall((:madqp_dist_create, libmadqp), Int32,
                     (Ptr{Cvoid}, Int32, Int32, Int32, Int32, Int64, Int64, Ptr{UInt8}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}),
                     ctx.ptr, cfg.rank, cfg.world, cfg.P, cfg.Q, nx, cfg.nb, cfg.world > 1 ? pointer(id) : C_NULL, C_NULL, dref))
    aug_com = HIPDistributedMatrix{T}(dref[], ctx, nx)
    lay = zeros(Int64, 8)                        # (p, q, tile rows, tile columns, mloc, nloc, ld, ncp)
    check(ctx, ccall((:madqp_dist_layout, libmadqp), Int32, (Ptr{Cvoid}, Ptr{Int64}), dref[], lay))
    p, q, ld, ncp = Int(lay[1]), Int(lay[2]), Int(lay[7]), Int(lay[8])
    m16 = max(cld(m, 16) * 16, 1)
    zeros_dev(r, c) = reshape(fill!(VT(undef, r * c), zero(T)), r, c)
    A_I, A_J = zeros_dev(ld, m16), zeros_dev(ncp, m16)     # Julia (ld x m16) = the library's m16 rows of length ld
    Hloc = cb.nnzh > 0 ? zeros_dev(ld, ncp) : zeros_dev(0, 0)   # Julia (ld x ncp) = the local matrix, column-major
    jI = MadNLP.create_array(cb, Int32, cb.nnzj); jJ = MadNLP.create_array(cb, Int32, cb.nnzj)
    MadNLP._jac_sparsity_wrapper!(cb, jI, jJ)
    hI = MadNLP.create_array(cb, Int32, cb.nnzh); hJ = MadNLP.create_array(cb, Int32, cb.nnzh)
    cb.nnzh > 0 && MadNLP._hess_sparsity_wrapper!(cb, hI, hJ)
    jIh, jJh = Vector{Int32}(Array(jI)), Vector{Int32}(Array(jJ))
    function cols_map(R, r)
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        check(ctx, ccall((:madqp_coo_map_create_cols_cyclic, libmadqp), Int32,
                         (Ptr{Cvoid}, Int64, Ptr{Int32}, Ptr{Int32}, Int64, Int64, Int64, Int32, Int32, Ref{Ptr{Cvoid}}),
                         ctx.ptr, length(jIh), jIh, jJh, m, nx, cfg.nb, R, r, ref))
        return ref[]
    end
    jacI_map, jacJ_map = cols_map(cfg.P, p), cols_map(cfg.Q, q)
    hess_map = C_NULL
    if cb.nnzh > 0
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        hIh, hJh = Vector{Int32}(Array(hI)), Vector{Int32}(Array(hJ))
        check(ctx, ccall((:madqp_coo_map_create_tiles_cyclic, libmadqp), Int32,
                         (Ptr{Cvoid}, Int64, Ptr{Int32}, Ptr{Int32}, Int64, Int64, Int32, Int32, Int32, Int32, Ref{Ptr{Cvoid}}),
                         ctx.ptr, length(hIh), hIh, hJh, nx, cfg.nb, cfg.P, p, cfg.Q, q, ref))
        hess_map = ref[]
    end
    append!(aug_com.maps, filter(!=(C_NULL), [jacI_map, jacJ_map, hess_map]))
    kref = Ref{Ptr{Cvoid}}(C_NULL)
    ineq0 = Int64.(Array(ind_ineq)) .- 1
    check(ctx, ccall((:madqp_dkkt_create, libmadqp), Int32,
                     (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Int64}, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Float64},
                      Int64, Ref{Ptr{Cvoid}}),
                     dref[], nx, m, ns, ineq0, cb.nnzh > 0 ? dptr(Hloc) : NULLF, ld, dptr(A_I), ld, dptr(A_J), ncp, kref))
    aug_com.dkkt = kref[]
    ls = linear_solver(aug_com; opt = opt_linear_solver)
    mk(k) = VT(undef, k)
    reg, pr_diag, du_diag = mk(n), mk(n), mk(m)
    l_diag, u_diag, l_lower, u_lower = mk(nlb), mk(nub), mk(nlb), mk(nub)
    ind_lb0, ind_ub0 = ind_cons.ind_lb .- 1, ind_cons.ind_ub .- 1
    cs = CState(n, m, nlb, nub, dptr(ind_lb0), dptr(ind_ub0),
                NULLF, NULLF, NULLF, NULLF, NULLF, NULLF, NULLF, NULLF, NULLF, NULLF, NULLF, NULLF, NULLF,
                dptr(reg), dptr(pr_diag), dptr(du_diag), dptr(l_diag), dptr(l_lower), dptr(u_diag), dptr(u_lower))
    VI = typeof(ind_cons.ind_lb)
    return HIPDistributedKKTSystem{T, VT, typeof(A_I), MadNLP.ExactHessian{T, VT}, VI, typeof(ls)}(
        aug_com, ctx, Hloc, A_I, A_J, ld, ncp, mk(cb.nnzj), mk(cb.nnzh), jacI_map, jacJ_map, hess_map,
        reg, pr_diag, du_diag, l_diag, u_diag, l_lower, u_lower, ls,
        ind_ineq, ind_cons.ind_lb, ind_cons.ind_ub, ind_lb0, ind_ub0, Ref(cs), n, m, nx)
end

MadNLP.num_variables(kkt::HIPDistributedKKTSystem) = kkt.n
MadNLP.get_jacobian(kkt::HIPDistributedKKTSystem) = kkt.jac
MadNLP.get_hessian(kkt::HIPDistributedKKTSystem) = kkt.hess
MadNLP.is_inertia_correct(kkt::HIPDistributedKKTSystem, num_pos, num_zero, num_neg) =
    (num_zero == 0) && (num_pos == kkt.aug_com.order)
function MadNLP.initialize!(kkt::HIPDistributedKKTSystem{T}) where {T}     # src/KKT/normalkkt.jl:136-147
    fill!(kkt.reg, one(T)); fill!(kkt.pr_diag, one(T)); fill!(kkt.du_diag, zero(T))
    fill!(kkt.l_lower, zero(T)); fill!(kkt.u_lower, zero(T)); fill!(kkt.l_diag, one(T)); fill!(kkt.u_diag, one(T))
    return
end
function MadNLP.compress_jacobian!(kkt::HIPDistributedKKTSystem)          # this rank's two column subsets of A
    for (mp, dst, ld) in ((kkt.jacI_map, kkt.A_I, kkt.ld), (kkt.jacJ_map, kkt.A_J, kkt.ncp))
        check(kkt.ctx, ccall((:madqp_coo_map_apply, libmadqp), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64),
                             mp, dptr(kkt.jac), dptr(dst), ld))
    end
    return
end
function MadNLP.compress_hessian!(kkt::HIPDistributedKKTSystem)           # this rank's lower tiles of H
    kkt.hess_map == C_NULL && return
    check(kkt.ctx, ccall((:madqp_coo_map_apply, libmadqp), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64),
                         kkt.hess_map, dptr(kkt.hess), dptr(kkt.Hloc), kkt.ld))
    return
end
function MadNLP.jtprod!(y::AbstractVector, kkt::HIPDistributedKKTSystem, x::AbstractVector)
    check(kkt.ctx, ccall((:madqp_dkkt_jtprod, libmadqp), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
                         kkt.aug_com.dkkt, dptr(y), dptr(x)))
    return y
end
function MadNLP.build_kkt!(kkt::HIPDistributedKKTSystem)                  # no communication: K_loc = H_loc + A_I' (Theta A_J)
    check(kkt.ctx, ccall((:madqp_dkkt_build, libmadqp), Int32, (Ptr{Cvoid}, Ref{CState}), kkt.aug_com.dkkt, kkt.cstate))
    return
end
function MadNLP.solve!(kkt::HIPDistributedKKTSystem, w::MadNLP.AbstractKKTVector)
    check(kkt.ctx, ccall((:madqp_dkkt_solve, libmadqp), Int32, (Ptr{Cvoid}, Ref{CState}, Ptr{Float64}),
                         kkt.aug_com.dkkt, kkt.cstate, dptr(MadNLP.full(w))))
    return w
end
function mul!(w::MadNLP.AbstractKKTVector{T}, kkt::HIPDistributedKKTSystem, v::MadNLP.AbstractKKTVector,
              alpha = one(T), beta = zero(T)) where {T}
    check(kkt.ctx, ccall((:madqp_dkkt_mul, libmadqp), Int32,
                         (Ptr{Cvoid}, Ref{CState}, Ptr{Float64}, Ptr{Float64}, Float64, Float64),
                         kkt.aug_com.dkkt, kkt.cstate, dptr(MadNLP.full(w)), dptr(MadNLP.full(v)), alpha, beta))
    return w
end
# The per-variable kernels below (set_aug_diagonal_reg! ... get_fraction_to_boundary_step) act on replicated vectors:
# the distributed system takes the plain (non-K2.5) forms of the library through the same overrides.
function MadIPM.set_aug_diagonal_reg!(kkt::HIPDistributedKKTSystem{T}, solver::MadNLP.AbstractMadNLPSolver{T}) where {T}
    check(kkt.ctx, ccall((:madqp_set_aug_diagonal_reg, libmadqp), Int32, (Ptr{Cvoid}, Ref{CState}, Float64, Float64),
                         kkt.ctx.ptr, state(solver), solver.del_w, solver.del_c))
    return
end

# --------------------------------------------------------------------------- src/kernels.jl on the device
# full state view of one solver (iterates + the KKT's diagonals), built on first use
const STATES = WeakKeyDict{Any, Base.RefValue{CState}}()
function state(solver)
    get!(STATES, solver) do
        kkt = solver.kkt
        Ref(CState(solver.n, solver.m, solver.nlb, solver.nub, dptr(kkt.ind_lb0), dptr(kkt.ind_ub0),
                   dptr(MadNLP.full(solver.x)), dptr(MadNLP.full(solver.xl)), dptr(MadNLP.full(solver.xu)),
                   dptr(MadNLP.full(solver.zl)), dptr(MadNLP.full(solver.zu)), dptr(MadNLP.full(solver.f)),
                   dptr(solver.y), dptr(solver.c), dptr(solver.jacl),
                   dptr(MadNLP.full(solver.d)), dptr(MadNLP.full(solver.p)),
                   dptr(solver.correction_lb), dptr(solver.correction_ub),
                   dptr(kkt.reg), dptr(kkt.pr_diag), dptr(kkt.du_diag),
                   dptr(kkt.l_diag), dptr(kkt.l_lower), dptr(kkt.u_diag), dptr(kkt.u_lower)))
    end
end

macro k(name, argtypes, args...)
    esc(:(check(solver.kkt.ctx, ccall(($(QuoteNode(name)), libmadqp), Int32,
                                      (Ptr{Cvoid}, Ref{CState}, $(argtypes.args...)),
                                      solver.kkt.ctx.ptr, state(solver), $(args...)))))
end
const HIPSolver = MadIPM.MPCSolver{T, VT, VI, K} where {T, VT, VI, K <: Union{HIPKKTSystem, HIPDistributedKKTSystem}}

# kernels.jl:128-146, and :149-165 for the K2.5 form: the library dispatches on the KKT object
function MadIPM.set_aug_diagonal_reg!(kkt::HIPKKTSystem{T}, solver::MadNLP.AbstractMadNLPSolver{T}) where {T}
    check(kkt.ctx, ccall((:madqp_kkt_set_aug_diagonal_reg, libmadqp), Int32, (Ptr{Cvoid}, Ref{CState}, Float64, Float64),
                         kkt.handle, state(solver), solver.del_w, solver.del_c))
    return
end
MadIPM.set_initial_primal_rhs!(solver::HIPSolver) = @k madqp_set_initial_primal_rhs ()
MadIPM.set_initial_dual_rhs!(solver::HIPSolver) = @k madqp_set_initial_dual_rhs ()
const AnyHIPKKT = Union{HIPKKTSystem, HIPDistributedKKTSystem}
MadIPM.set_predictive_rhs!(solver::MadNLP.AbstractMadNLPSolver, ::AnyHIPKKT) = @k madqp_set_predictive_rhs ()
MadIPM.set_correction_rhs!(solver::MadNLP.AbstractMadNLPSolver, ::AnyHIPKKT, mu::Float64,
                           clb::AbstractVector{Float64}, cub::AbstractVector{Float64}, ilb, iub) =
    @k madqp_set_correction_rhs (Float64,) mu
MadIPM.get_correction!(solver::HIPSolver, clb, cub) = @k madqp_get_correction ()
MadIPM.set_extra_correction!(solver::HIPSolver, clb, cub, ap, ad, bmin, bmax, mu) =
    @k madqp_set_extra_correction (Float64, Float64, Float64, Float64, Float64) ap ad bmin bmax mu

function MadIPM.get_complementarity_measure(solver::HIPSolver)            # kernels.jl:171-190
    out = Ref{Float64}(0.0)
    @k madqp_get_complementarity_measure (Ref{Float64},) out
    return out[]
end
function MadIPM.get_affine_complementarity_measure(solver::HIPSolver, ap, ad)   # kernels.jl:192-224
    out = Ref{Float64}(0.0)
    @k madqp_get_affine_complementarity_measure (Float64, Float64, Ref{Float64}) ap ad out
    return out[]
end
function MadIPM.get_fraction_to_boundary_step(solver::HIPSolver, tau)      # kernels.jl:290-305
    a = zeros(Float64, 4); ib = zeros(Int64, 4)
    @k madqp_get_alpha_max (Float64, Ptr{Float64}, Ptr{Int64}) tau a ib
    return min(a[1], a[2]), min(a[3], a[4])
end

# The four axpy! of src/solver.jl:332-335, MadNLP.adjust_boundary! (:342), the residual norms of :264-272 and the
# map!/mapreduce of init_starting_point! (:41-123) are plain broadcasts over the solver's ROCArrays in the reference:
# they need no override to be correct, only to be fused (madqp_update_iterates, madqp_adjust_boundary, madqp_get_inf,
# madqp_sp_*).  MehrotraAdaptiveStep's scalar reads (src/kernels.jl:349-369) need `AMDGPU.@allowscalar` as they do
# with CUDA.  INTEGRATION.md lists these optional bindings.

end # module
