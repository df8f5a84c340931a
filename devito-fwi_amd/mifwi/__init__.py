"""
ctypes binding of libmifwi.so, the MI355X-native acoustic propagator library
(C ABI declared in include/mifwi.h).  There is no CPU fallback: importing works anywhere the
shared library loads, but creating a plan without a GPU raises.
"""
from ._lib import (Plan, MifwiError, load_library, library_path, EXPORTED_SYMBOLS, l2_misfit_noalloc,  # noqa: F401
                   F32, F64, FLAG_ASYNC, FLAG_NO_GRAPH, FLAG_GENERIC, FLAG_CHECKPOINT,
                   FLAG_FREE_SURFACE, FLAG_OT4, FLAG_SAVE_F16, FLAG_TIME_BLOCK, FLAG_PML)
