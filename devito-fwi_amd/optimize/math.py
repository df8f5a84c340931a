"""Small numerical helpers of the optimisation package.

Same results as the reference's optimize/math.py:28-66,68-77 (safeguarded parabolic backtrack,
three-point parabola vertex, angle between vectors).
"""
import numpy as np


def dot(x, y):
    """Inner product of two arrays of any (equal) shape (optimize/math.py:74-77)."""
    return np.dot(np.squeeze(x), np.squeeze(y))


def angle(x, y):
    """Angle between x and y in radians (optimize/math.py:68-72)."""
    return np.arccos(dot(x, y) / (dot(x, x) * dot(y, y)) ** 0.5)


def backtrack2(f0, g0, x1, f1, b1=0.1, b2=0.5):
    """Minimiser of the parabola through (0, f0) with slope g0 and (x1, f1), kept inside
    [b1*x1, b2*x1] (optimize/math.py:28-39)."""
    x2 = -g0 * x1 ** 2 / (2 * (f1 - f0 - g0 * x1))
    if x2 > b2 * x1:
        return b2 * x1
    if x2 < b1 * x1:
        return b1 * x1
    return x2


def polyfit2(x, f):
    """Vertex of the parabola through the lowest sample and its two neighbours
    (optimize/math.py:48-57); raises when those three points are not convex."""
    i = int(np.argmin(f))
    a, b, _ = np.polyfit(x[i - 1:i + 2], f[i - 1:i + 2], 2)
    if not a > 0:
        raise ArithmeticError('polyfit2: the bracketing samples are not convex')
    return -b / (2 * a)
