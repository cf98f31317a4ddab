"""torch_nf_amd -- the torch_nf coupling-flow hot path on AMD MI355X (gfx950).

Same classes and module layout as the reference's hot path
(`bijectors`, `density_estimator`, `conditional_density_estimator`,
`error_formatters`); the arithmetic runs in hand-written HIP kernels behind the C ABI
of include/tnf.h.  `install_as_torch_nf()` registers these modules under the
reference's import names so existing `import torch_nf.bijectors` code picks them up.
"""
import sys

from . import _lib  # noqa: F401  (fails loudly if libtnf_hip.so is missing)
from . import bijectors, conditional_density_estimator, density_estimator, error_formatters, graphs, lfi, systems
from .bijectors import MAF, ToInterval, ToSimplex, Affine, BatchNorm, Bijector, RealNVP
from .conditional_density_estimator import ConditionalDensityEstimator
from .density_estimator import DensityEstimator, NormFlow

__version__ = "0.1.0"
__all__ = ["Bijector", "RealNVP", "MAF", "ToInterval", "ToSimplex", "Affine", "BatchNorm", "DensityEstimator", "NormFlow",
           "ConditionalDensityEstimator", "install_as_torch_nf"]


def install_as_torch_nf():
    """Alias this package's modules as torch_nf.* (the reference's namespace package)."""
    this = sys.modules[__name__]
    sys.modules.setdefault("torch_nf", this)
    for name in ("bijectors", "density_estimator", "conditional_density_estimator", "error_formatters", "lfi", "systems"):
        sys.modules["torch_nf." + name] = getattr(this, name)
    return this
