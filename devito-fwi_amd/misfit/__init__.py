"""Data misfit functions: least squares (reference misfit/misfit.py:5-9) and the trace-wise
and 2-D quadratic Wasserstein misfits (misfit/misfit.py:11-104; the 2-D one through the back-and-forth
solver of bfm.py)."""
from .misfit import least_square, qWasserstein, Misfit  # noqa
