from .spectrum_model import SpectrumModel  # noqa: F401
from .echelle_model import EchelleModel  # noqa: F401
from .kernels import global_covariance_matrix, local_covariance_matrix  # noqa: F401

__all__ = ["SpectrumModel", "EchelleModel", "global_covariance_matrix", "local_covariance_matrix"]
