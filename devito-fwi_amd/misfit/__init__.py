"""Data misfit functions.  Only the least-squares misfit of the reference (misfit/misfit.py:5-9)
is part of this package; its Wasserstein misfits are not."""
from .misfit import least_square, qWasserstein  # noqa
