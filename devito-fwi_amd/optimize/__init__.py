"""Gradient-based optimisers with trial-by-trial line searches, interface-compatible with the
reference's `optimize` package (optimize/__init__.py:1-3)."""
from .base import NLCG, LBFGS, SteepestDescent, base, Writer  # noqa: F401
from . import line_search, optimizer, math  # noqa: F401
