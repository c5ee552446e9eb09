"""MI355X-native NLP-callback engine for DirectTrajOpt.jl -- host-side mirror (Python).

The product is the C-ABI library ``libdto_engine.so`` (``csrc/``, ``include/dto_engine.h``); this
package mirrors the reference's plugin surface for the hot path (same names, argument meaning and
error behaviour) on top of it through ``ctypes``:

    NamedTrajectory, BilinearIntegrator, DerivativeIntegrator, QuadraticRegularizer,
    LinearRegularizer, MinimumTimeObjective, NullObjective, CompositeObjective (``+`` / ``*``),
    NonlinearKnotPointConstraint (built-in g kinds), DirectTrajOptProblem, Evaluator (MOI surface)

There is no CPU fallback: constructing an ``Evaluator`` without the HIP library or without a GPU
raises.  The directory name carries a dot, so import it through ``dto_amd`` (repo root shim).
"""
from .host.trajectory import NamedTrajectory  # noqa: F401
from .host.problem import (  # noqa: F401
    BilinearIntegrator,
    DerivativeIntegrator,
    QuadraticRegularizer,
    LinearRegularizer,
    MinimumTimeObjective,
    NullObjective,
    CompositeObjective,
    KnotPointObjective,
    TerminalObjective,
    GlobalObjective,
    GlobalKnotPointObjective,
    NonlinearGlobalConstraint,
    HostIntegrator,
    TimeDependentBilinearIntegrator,
    ModulatedGenerators,
    ket_fidelity_factor,
    NonlinearKnotPointConstraint,
    DirectTrajOptProblem,
)
from .host.evaluator import Evaluator, EngineError, load_library, library_path  # noqa: F401
from .host import capi, synthetic, distributed  # noqa: F401
from . import host  # noqa: F401
