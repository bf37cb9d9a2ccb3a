"""cmf.jl_amd -- MI355X (gfx950) implementation of CMF.jl's multiplicative-update hot path.

The directory name is not a Python identifier; import it through the shim at the
repo root:  ``import cmf_jl_amd as cmf``.
"""
from ._lib import CMFError, LIB_PATH, SYMBOLS, load as load_library  # noqa: F401
from .host import (  # noqa: F401
    EPSILON, AbsoluteLoss, AbsolutePenalty, AbstractCFUpdate, AlternatingOptimizer, CNMF_results, HALSUpdate, HIPHALSUpdate,
    HIPMultUpdate, HIPPGDUpdate, MaskedLoss, MultUpdate, NonnegConstraint, PGDUpdate, SquareLoss, SquarePenalty, UnitNormConstraint,
    compute_loss, converged, evaluate_convergence, evaluate_mse, evaluate_test, fit, fit_cnmf, gen_synthetic,
    init_rand, load_model, parameter_sweep, rccl_version, save_model, tensor_conv, tensor_transconv,
)

__version__ = "0.1.0"
