"""bayesssm_amd: MI355X-native particle filter + PMMH engine behind bayesSSM's API surface.

Host side mirrors the reference's R functions (same names, argument meaning and error text);
the work runs in hand-written HIP kernels behind the C ABI of include/bayesssm_amd.h.
Importing this package does not touch the GPU; the first call does, and fails loudly if the
HIP library or a device is missing (there is no CPU fallback)."""
from . import models
from .diagnostics import PmmhOutput, ess, rhat, summary
from .filters import (auxiliary_filter, auxiliary_filter_batch, batch_max_particles, bootstrap_filter, bootstrap_filter_batch, bootstrap_filter_multi, dump_draws,
                      particle_filter_core, resample_move_filter, resample_move_filter_batch)
from .pmmh import (default_tune_control, pmmh, prior_exponential, prior_flat, prior_halfnormal, prior_normal,
                   prior_uniform)
from .resampling import (resample_multinomial, resample_multinomial_cpp, resample_stratified,
                         resample_stratified_cpp, resample_systematic, resample_systematic_cpp, set_seed)
from .sharded import bootstrap_filter_sharded
from ._lib import BssmError, Context, default_context

__all__ = [
    "models", "bootstrap_filter_sharded", "ess", "rhat", "summary", "PmmhOutput", "auxiliary_filter", "bootstrap_filter", "bootstrap_filter_batch", "bootstrap_filter_multi", "auxiliary_filter_batch", "resample_move_filter_batch", "batch_max_particles", "resample_move_filter", "particle_filter_core", "dump_draws",
    "default_tune_control", "pmmh", "prior_exponential", "prior_flat", "prior_halfnormal", "prior_normal", "prior_uniform",
    "resample_multinomial", "resample_multinomial_cpp", "resample_stratified", "resample_stratified_cpp",
    "resample_systematic", "resample_systematic_cpp", "set_seed", "BssmError", "Context", "default_context",
]
