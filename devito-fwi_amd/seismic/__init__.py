"""
Host-side mirror of the reference's `seismic` package for the isotropic acoustic FWI path:
same class and function names, numpy-backed data, propagation on the GPU through libmifwi.
"""
from .fields import *  # noqa
from .fields import Grid, Function, TimeFunction, Constant, DeviceTimeFunction  # noqa
from .model import *  # noqa
from .source import *  # noqa
from .preset_models import *  # noqa
from .utils import *  # noqa
from .plotting import *  # noqa
