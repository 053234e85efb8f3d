"""nonlin_amd -- MI355X (gfx950) implementation of nonlin's Jacobian-evaluate + linear-solve
inner loop (least_squares_solver / newton_solver / vecfcn_helper%jacobian) behind the C ABI of
include/nonlin_hip.h.  See DESIGN.md.  There is no CPU fallback."""
from ._lib import NonlinHipUnavailable, LIB_PATH  # noqa: F401
from .api import (  # noqa: F401
    NonlinError, iteration_behavior, vecfcn_helper, equation_solver, least_squares_solver,
    line_search, line_search_solver, newton_solver, quasi_newton_solver,
    constrained_equation_solver, constrained_least_squares_solver, polynomial,
    fcnnvar_helper, equation_optimizer, line_search_optimizer, bfgs,
    NL_NO_ERROR, NL_INVALID_INPUT_ERROR, NL_ARRAY_SIZE_ERROR, NL_OUT_OF_MEMORY_ERROR,
    NL_INVALID_OPERATION_ERROR, NL_CONVERGENCE_ERROR, NL_DIVERGENT_BEHAVIOR_ERROR,
    NL_SPURIOUS_CONVERGENCE_ERROR, NL_TOLERANCE_TOO_SMALL_ERROR, NL_INDEX_OUT_OF_RANGE_ERROR,
    NL_DIVIDE_BY_ZERO_ERROR, NL_UNDEFINED_FUNCTION_ERROR, NL_UNDERDEFINED_PROBLEM_ERROR,
)

__all__ = [n for n in dir() if not n.startswith("_")]
