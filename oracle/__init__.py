"""CPU oracle for the Mehrotra predictor-corrector KKT path of MadIPM / MadQP.jl.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  Nothing under
``madqp_jl_amd/`` imports it; the product path fails loudly when the HIP
library is missing.

It is a line-cited numpy/scipy restatement of the reference's algorithm
(``/root/reference/src/{solver,kernels,linear_solver}.jl``,
``src/KKT/normalkkt.jl``) plus the pieces of the un-vendored dependency
MadNLP.jl 0.8.x (compat bound ``Project.toml:17``) that the path calls
(``reduce_rhs!``, ``finish_aug_solve!``, ``_kktmul!``, ``get_inf_*``,
``adjust_boundary!``, ``initialize!``, ``set_scaling!``,
``get_index_constraints``), restated from their published behaviour.

PARITY UNPINNED: the reference is pure Julia, Julia is absent from this
container, and the reference's tests hold no absolute golden vectors (every
assertion in ``test/runtests.jl`` is solver-A == solver-B on inputs drawn from
Julia's RNG).  The oracle is therefore pinned only by
known-answer problems (``simple_lp`` of ``test/runtests.jl:24-55`` -> objective
1.0 at x = (0.5, 0.5); HS21 -> -99.96 at (2, 0)), by cross-checks against
``scipy.optimize.linprog`` (HiGHS), and by the cross-formulation equalities the
reference itself tests (K2 == normal equations == condensed,
``test/runtests.jl:102-115,165-180``).
"""
