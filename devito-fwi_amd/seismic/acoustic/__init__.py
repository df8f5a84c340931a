from .wavesolver import *  # noqa
