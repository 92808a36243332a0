"""Options and rule types of the MPC solver (src/utils.jl:10-48, 69-117)."""
from __future__ import annotations


class ConservativeStep:  # src/utils.jl:19-21
    def __init__(self, tau=0.995):
        self.tau = tau


class AdaptiveStep:  # src/utils.jl:23-25
    def __init__(self, tau_min=0.99):
        self.tau_min = tau_min


class MehrotraAdaptiveStep:  # src/utils.jl:27-29
    def __init__(self, gamma_f=0.99):
        self.gamma_f = gamma_f


class NoRegularization:  # src/utils.jl:37
    pass


class FixedRegularization:  # src/utils.jl:39-42
    def __init__(self, delta_p, delta_d):
        self.delta_p, self.delta_d = delta_p, delta_d


class AdaptiveRegularization:  # src/utils.jl:44-48
    def __init__(self, delta_p, delta_d, delta_min):
        self.delta_p, self.delta_d, self.delta_min = delta_p, delta_d, delta_min


class IPMOptions:
    """``IPMOptions`` with the reference defaults (src/utils.jl:69-103; tol preset :110).

    ``kkt_system`` / ``linear_solver`` default to the HIP plugin types; the
    condensed system needs ``delta_d < 0`` when the problem has equality rows.
    """

    _DEFAULTS = dict(
        tol=1e-8, max_iter=3000, scaling=True, bound_push=1e-2, bound_fac=1e-2,
        bound_relax_factor=1e-8, max_ncorr=0, mu_init=1e-1, mu_min=1e-11,
        tol_linear_solve=1e-8, check_residual=False, rethrow_error=False, print_level=0,
        # "condensed" (HIPCondensedKKTSystem), "normal" (HIPNormalKKTSystem, LP) or "augmented" (HIPAugmentedKKTSystem:
        # the K2 form, equality rows without dual regularization)
        kkt_system="condensed",
        # src/utils.jl:81: MadNLP.RelaxBound ("relax_bound": a fixed variable keeps both bounds, relaxed by
        # bound_relax_factor like any other) for condensed KKT systems -- here the condensed and augmented systems --,
        # MadNLP.MakeParameter ("make_parameter": fixed variables leave the problem, DeviceQP.eliminate_fixed) otherwise
        # -- here the normal equations.  None = that rule; "error" refuses fixed variables.
        fixed_variable_treatment=None,
        driver="python",  # "python": this package drives each kernel; "native": one C call per iteration
        # extension (0 = the reference's solve_system!, src/linear_solver.jl:19-45): steps of iterative refinement with the
        # residual that solve_system! forms anyway (+1 solve and +1 mul! per solve and step).  The device factorisation
        # multiplies with explicit inverses of 16 x 16 sub-blocks (diagonal kernel, panel solve, sweeps) where LAPACK
        # substitutes scalar by scalar; on small ill-conditioned problems that shows in the per-iteration traces (tens of
        # times the CPU noise floor on the random soak problems of order <= 260), and one step removes
        # it (0 of 398 cases beyond the floor).  None (default) = AUTO: one step while the factorised matrix has order
        # <= REFINE_AUTO_MAX -- where a solve costs microseconds -- and none above, where the 16-blocked arithmetic is
        # what LAPACK's own blocked kernels do and the larger parity cases sit within the stated bar without it.
        refine_steps=None,
    )
    REFINE_AUTO_MAX = 1024  # MADQP_REFINE_AUTO_MAX overrides

    @staticmethod
    def refine_auto(order: int) -> int:
        """refine_steps of the AUTO rule for a factorised matrix of this order"""
        import os

        return 1 if order <= int(os.environ.get("MADQP_REFINE_AUTO_MAX", IPMOptions.REFINE_AUTO_MAX)) else 0

    def __init__(self, **kw):
        for k, v in self._DEFAULTS.items():
            setattr(self, k, v)
        self.regularization = FixedRegularization(1e-8, 0.0)
        self.step_rule = AdaptiveStep(0.99)
        for k, v in kw.items():
            if not hasattr(self, k):
                raise TypeError(f"unknown option {k!r}")  # MadNLP warns; we are strict
            setattr(self, k, v)
