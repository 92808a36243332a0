# MadQPHIP.jl -- Julia glue that drops libmadqp_hip.so into MadIPM's solver loop.
#
# STATUS: written against the MadNLP 0.8.x plugin contract as used by MadIPM
# (/root/reference/src/KKT/normalkkt.jl, src/linear_solver.jl, src/kernels.jl); it has NOT been
# executed in the authoring environment (no Julia toolchain there, SURVEY.md section 0).  Every
# ccall below binds one prototype of include/madqp.h; the same ABI is exercised end to end by the
# Python host mirror (madqp_jl_amd/) and its GPU tests.
#
# What it provides
#   HIPCondensedKKTSystem <: MadNLP.AbstractKKTSystem   (modelled on NormalKKTSystem)
#   HIPCholeskySolver     <: MadNLP.AbstractLinearSolver
#   methods of MadIPM's per-variable kernels specialised on HIPCondensedKKTSystem so that `mpc!`
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
    rc < 0 && error("libmadqp_hip: $(last_error(ctx)) ($rc)")
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

# --------------------------------------------------------------------------- linear solver
mutable struct HIPCholeskySolver{T} <: MadNLP.AbstractLinearSolver{T}
    kkt_handle::Ptr{Cvoid}                # the library object that owns K and its factor
    ctx::Context
    info::Int32
    opt::MadNLP.AbstractOptions
    logger::MadNLP.MadNLPLogger
end

MadNLP.@kwdef mutable struct HIPCholeskyOptions <: MadNLP.AbstractOptions end
MadNLP.default_options(::Type{HIPCholeskySolver}) = HIPCholeskyOptions()
MadNLP.introduce(::HIPCholeskySolver) = "madqp-hip blocked left-looking fp64 Cholesky (MFMA, gfx950)"
MadNLP.is_supported(::Type{HIPCholeskySolver}, ::Type{Float64}) = true
MadNLP.is_inertia(::HIPCholeskySolver) = true
MadNLP.inertia(s::HIPCholeskySolver) = s.info == 0 ? (typemax(Int), 0, 0) : (Int(s.info) - 1, 0, 1)
MadNLP.improve!(::HIPCholeskySolver) = false
MadIPM.is_factorized(s::HIPCholeskySolver) = s.info == 0        # src/utils.jl:54-62

function MadNLP.factorize!(s::HIPCholeskySolver)
    info = Ref{Int32}(0)
    check(s.ctx, ccall((:madqp_kkt_factorize, libmadqp), Int32, (Ptr{Cvoid}, Ref{Int32}), s.kkt_handle, info))
    s.info = info[]              # > 0: not positive definite -> x100 regularization retry (src/linear_solver.jl:11-15)
    return s
end

# --------------------------------------------------------------------------- KKT system
struct HIPCondensedKKTSystem{T, VT, MT, QN, VI} <: MadNLP.AbstractKKTSystem{T, VT, MT, QN}
    handle::Ptr{Cvoid}
    ctx::Context
    H::MT                 # nx x nx dense symmetric (device)
    At::MT                # nx x m column-major == A with row k contiguous (device)
    jac::VT               # callback buffers (dense row-major Jacobian / Hessian values)
    hess::VT
    # fields MadIPM reads generically (src/kernels.jl:135-144, src/solver.jl:16-18)
    reg::VT; pr_diag::VT; du_diag::VT
    l_diag::VT; u_diag::VT; l_lower::VT; u_lower::VT
    linear_solver::HIPCholeskySolver{T}
    ind_ineq::VI; ind_lb::VI; ind_ub::VI      # 1-based, as MadNLP keeps them
    ind_lb0::VI; ind_ub0::VI                  # 0-based device copies handed to the library
    n::Int; m::Int; nx::Int
end

function MadNLP.create_kkt_system(
    ::Type{HIPCondensedKKTSystem}, cb::MadNLP.AbstractCallback{T, VT}, ind_cons, linear_solver::Type;
    opt_linear_solver = MadNLP.default_options(linear_solver),
    hessian_approximation = MadNLP.ExactHessian, qn_options = MadNLP.QuasiNewtonOptions(),
) where {T, VT}
    nx, m = cb.nvar, cb.ncon
    ind_ineq = ind_cons.ind_ineq
    ns = length(ind_ineq)
    n = nx + ns
    nlb, nub = length(ind_cons.ind_lb), length(ind_cons.ind_ub)
    ctx = Context()
    H = fill!(similar(VT, nx * nx), zero(T)); H = reshape(H, nx, nx)
    At = fill!(similar(VT, nx * m), zero(T)); At = reshape(At, nx, m)
    mk(k) = VT(undef, k)
    ref = Ref{Ptr{Cvoid}}(C_NULL)
    ineq0 = Int64.(Array(ind_ineq)) .- 1
    rc = ccall((:madqp_kkt_create, libmadqp), Int32,
               (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Int64}, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ref{Ptr{Cvoid}}),
               ctx.ptr, nx, m, ns, ineq0, dptr(H), nx, dptr(At), nx, ref)
    check(ctx, rc)
    ls = HIPCholeskySolver{T}(ref[], ctx, Int32(0), opt_linear_solver, MadNLP.MadNLPLogger())
    VI = typeof(ind_cons.ind_lb)
    kkt = HIPCondensedKKTSystem{T, VT, typeof(H), MadNLP.ExactHessian{T, VT}, VI}(
        ref[], ctx, H, At, mk(nx * m), mk(nx * nx),
        mk(n), mk(n), mk(m), mk(nlb), mk(nub), mk(nlb), mk(nub), ls,
        ind_ineq, ind_cons.ind_lb, ind_cons.ind_ub, ind_cons.ind_lb .- 1, ind_cons.ind_ub .- 1, n, m, nx)
    return kkt
end

# The K2 form (MadNLP's default SparseKKTSystem, src/utils.jl:108) is the same glue with
# `:madqp_kkt_create_augmented` in the ccall above (same argument list) -- the object then answers every
# madqp_kkt_* call below as [H + Sigma_x, A'; A, -D] factorised L diag(I,-I) L'; its linear solver reports
# inertia (nx, 0, m) and `is_inertia_correct(kkt, p, z, n) = (z == 0) && (n == kkt.m)`; equality rows need
# no dual regularization (the reference's default FixedRegularization(1e-8, 0.0) works as is).

MadNLP.num_variables(kkt::HIPCondensedKKTSystem) = kkt.n
MadNLP.get_jacobian(kkt::HIPCondensedKKTSystem) = kkt.jac
MadNLP.get_hessian(kkt::HIPCondensedKKTSystem) = kkt.hess
MadNLP.is_inertia_correct(kkt::HIPCondensedKKTSystem, p, z, n) = (z == 0) && (n == 0)

function MadNLP.initialize!(kkt::HIPCondensedKKTSystem{T}) where {T}      # src/KKT/normalkkt.jl:136-147
    fill!(kkt.reg, one(T)); fill!(kkt.pr_diag, one(T)); fill!(kkt.du_diag, zero(T))
    fill!(kkt.l_lower, zero(T)); fill!(kkt.u_lower, zero(T))
    fill!(kkt.l_diag, one(T)); fill!(kkt.u_diag, one(T))
    return
end

# dense callbacks write row-major values: the Jacobian buffer IS A with row k contiguous, i.e. At
MadNLP.compress_jacobian!(kkt::HIPCondensedKKTSystem) = copyto!(vec(kkt.At), kkt.jac)
MadNLP.compress_hessian!(kkt::HIPCondensedKKTSystem) = copyto!(vec(kkt.H), kkt.hess)

# state view for one solver: built once per MPCSolver and cached in a WeakKeyDict
const STATES = WeakKeyDict{Any, CState}()
function state(solver)
    get!(STATES, solver) do
        kkt = solver.kkt
        CState(solver.n, solver.m, solver.nlb, solver.nub, dptr(kkt.ind_lb0), dptr(kkt.ind_ub0),
               dptr(MadNLP.full(solver.x)), dptr(MadNLP.full(solver.xl)), dptr(MadNLP.full(solver.xu)),
               dptr(MadNLP.full(solver.zl)), dptr(MadNLP.full(solver.zu)), dptr(MadNLP.full(solver.f)),
               dptr(solver.y), dptr(solver.c), dptr(solver.jacl),
               dptr(MadNLP.full(solver.d)), dptr(MadNLP.full(solver.p)),
               dptr(solver.correction_lb), dptr(solver.correction_ub),
               dptr(kkt.reg), dptr(kkt.pr_diag), dptr(kkt.du_diag),
               dptr(kkt.l_diag), dptr(kkt.l_lower), dptr(kkt.u_diag), dptr(kkt.u_lower))
    end
end
# solve!/mul! receive only the KKT object: the owning solver registers its state here
const KKT_STATE = WeakKeyDict{Any, CState}()
register!(solver) = (KKT_STATE[solver.kkt] = state(solver); solver)

function MadNLP.jtprod!(y::AbstractVector, kkt::HIPCondensedKKTSystem, x::AbstractVector)
    check(kkt.ctx, ccall((:madqp_kkt_jtprod, libmadqp), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
                         kkt.handle, dptr(y), dptr(x)))
    return y
end

function MadNLP.build_kkt!(kkt::HIPCondensedKKTSystem)                  # src/KKT/normalkkt.jl:166-180
    check(kkt.ctx, ccall((:madqp_kkt_build, libmadqp), Int32, (Ptr{Cvoid}, Ref{CState}), kkt.handle, KKT_STATE[kkt]))
end

function MadNLP.solve!(kkt::HIPCondensedKKTSystem, w::MadNLP.AbstractKKTVector)   # :182-205
    check(kkt.ctx, ccall((:madqp_kkt_solve, libmadqp), Int32, (Ptr{Cvoid}, Ref{CState}, Ptr{Float64}),
                         kkt.handle, KKT_STATE[kkt], dptr(MadNLP.full(w))))
    return w
end

function LinearAlgebra.mul!(w::MadNLP.AbstractKKTVector{T}, kkt::HIPCondensedKKTSystem,
                            v::MadNLP.AbstractKKTVector, alpha = one(T), beta = zero(T)) where {T}   # :207-219
    check(kkt.ctx, ccall((:madqp_kkt_mul, libmadqp), Int32,
                         (Ptr{Cvoid}, Ref{CState}, Ptr{Float64}, Ptr{Float64}, Float64, Float64),
                         kkt.handle, KKT_STATE[kkt], dptr(MadNLP.full(w)), dptr(MadNLP.full(v)), alpha, beta))
    return w
end

# --------------------------------------------------------------------------- src/kernels.jl on the device
macro k(name, argtypes, args...)
    esc(:(check(solver.kkt.ctx, ccall(($(QuoteNode(name)), libmadqp), Int32,
                                      (Ptr{Cvoid}, Ref{CState}, $(argtypes.args...)),
                                      solver.kkt.ctx.ptr, state(solver), $(args...)))))
end
const HIPSolver = MadIPM.MPCSolver{T, VT, VI, <:HIPCondensedKKTSystem} where {T, VT, VI}

function MadIPM.set_aug_diagonal_reg!(kkt::HIPCondensedKKTSystem, solver::MadNLP.AbstractMadNLPSolver)   # kernels.jl:128-146
    register!(solver)
    @k madqp_set_aug_diagonal_reg (Float64, Float64) solver.del_w solver.del_c
end
MadIPM.set_initial_primal_rhs!(solver::HIPSolver) = @k madqp_set_initial_primal_rhs ()
MadIPM.set_initial_dual_rhs!(solver::HIPSolver) = @k madqp_set_initial_dual_rhs ()
MadIPM.set_predictive_rhs!(solver::MadNLP.AbstractMadNLPSolver, ::HIPCondensedKKTSystem) = @k madqp_set_predictive_rhs ()
MadIPM.set_correction_rhs!(solver::MadNLP.AbstractMadNLPSolver, ::HIPCondensedKKTSystem, mu::Float64, clb, cub, ilb, iub) =
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

# The four axpy! of src/solver.jl:332-335, MadNLP.adjust_boundary! (:342) and the residual norms of
# :264-272 are reached the same way (madqp_update_iterates, madqp_adjust_boundary, madqp_get_inf); they
# are plain broadcasts over ROCArrays in the reference and need no override to be correct, only to
# be fused.  INTEGRATION.md lists the remaining optional bindings.

end # module
