"""madqp_jl_amd -- MI355X-native Mehrotra predictor-corrector KKT path (MadIPM / MadQP.jl).

Hand-written HIP for gfx950 behind a C ABI (``include/madqp.h``, built into
``libmadqp_hip.so`` from ``csrc/``) plus the host-side mirror of the reference's
plugin interface: :class:`HIPCondensedKKTSystem` / :class:`HIPCholeskySolver`
(``MadNLP.AbstractKKTSystem`` / ``AbstractLinearSolver``) and :class:`MPCSolver`
(the ``mpc!`` loop of ``src/solver.jl``).  There is no CPU fallback.
"""
from ._lib import EXPORTED_SYMBOLS, LIB_PATH, MadQPError, load_cdll
from .backend import HipBackend, State
from .batch import shard, solve_batch
from .batched import BatchedMPCSolver
from .kkt import (HIPScaledAugmentedKKTSystem, HIPAugmentedKKTSystem, HIPCholeskySolver, HIPCondensedKKTSystem, HIPNormalKKTSystem, HIPSparseAugmentedKKTSystem, HIPSparseCondensedKKTSystem,
                  HIPSparseNormalKKTSystem)
from .options import (AdaptiveRegularization, AdaptiveStep, ConservativeStep, FixedRegularization,
                      IPMOptions, MehrotraAdaptiveStep, NoRegularization)
from .qp import DeviceCSR, DeviceQP, stream_key
from .solver import (ERROR_IN_STEP_COMPUTATION, MAXIMUM_ITERATIONS_EXCEEDED, SOLVE_SUCCEEDED,
                     MPCSolver, SolveException, solve)

__all__ = [
    "HipBackend", "State", "shard", "solve_batch", "BatchedMPCSolver", "solve", "HIPCholeskySolver", "HIPAugmentedKKTSystem", "HIPScaledAugmentedKKTSystem", "HIPCondensedKKTSystem", "HIPNormalKKTSystem", "HIPSparseAugmentedKKTSystem", "HIPSparseCondensedKKTSystem", "HIPSparseNormalKKTSystem", "MPCSolver", "DeviceQP",
    "DeviceCSR",
    "IPMOptions", "AdaptiveStep", "ConservativeStep", "MehrotraAdaptiveStep", "NoRegularization",
    "FixedRegularization", "AdaptiveRegularization", "MadQPError", "SolveException", "load_cdll",
    "EXPORTED_SYMBOLS", "LIB_PATH", "stream_key", "SOLVE_SUCCEEDED", "MAXIMUM_ITERATIONS_EXCEEDED",
    "ERROR_IN_STEP_COMPUTATION",
]
