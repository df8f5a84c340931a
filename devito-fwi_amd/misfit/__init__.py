"""Data misfit functions: least squares (reference misfit/misfit.py:5-9) and the trace-wise
quadratic Wasserstein misfit (misfit/misfit.py:11-104, method='1d')."""
from .misfit import least_square, qWasserstein, Misfit  # noqa
