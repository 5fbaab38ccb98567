"""Couple / Point value holders - same attribute and operator surface as the
reference's lib/couple.py:5-88 (a 2-value holder with arithmetic dunders)."""
import numpy as np


class Couple:
    def __init__(self, x, y):
        self.x = x
        self.y = y

    def _pair(self, other):
        return (other.x, other.y) if isinstance(other, Couple) else (other, other)

    def __add__(self, other):
        ox, oy = self._pair(other)
        return Couple(self.x + ox, self.y + oy)

    def __sub__(self, other):
        ox, oy = self._pair(other)
        return Couple(self.x - ox, self.y - oy)

    def __mul__(self, other):
        ox, oy = self._pair(other)
        return Couple(self.x * ox, self.y * oy)

    def __truediv__(self, other):
        ox, oy = self._pair(other)
        return Couple(self.x / ox, self.y / oy)

    def __floordiv__(self, other):
        ox, oy = self._pair(other)
        return Couple(self.x // ox, self.y // oy)

    def __repr__(self):
        return "%s(%s, %s)" % (type(self).__name__, self.x, self.y)

    __str__ = __repr__

    def copy(self):
        return Couple(self.x, self.y)

    def toList(self):
        return [self.x, self.y]

    def toTuple(self):
        return (self.x, self.y)

    def toArray(self):
        return np.array([self.x, self.y])


class Point(Couple):
    pass
