"""enlsip_gn — host mirror of Enlsip.jl's Gauss-Newton subproblem interface over libenlsip_gn.so."""
from .api import GNSolver, GNResult, GNError, FactorView, SQRT_EPS  # noqa: F401
from ._lib import FACTOR_A, FACTOR_L11, FACTOR_J2, FLAG_UPDATE_MFMA, FLAG_UPDATE_REFLECTORS, LIB_PATH  # noqa: F401
