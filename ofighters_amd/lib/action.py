"""Action - what an agent hands back for one ship and one tick.

API mirror of the reference's lib/action.py:12-79 (constructor keywords,
`.shoot/.thrust/.pointing/.vector`, class constants `size`/`shape`, and the two
exception messages); the device-side form is the 12-byte ofx_action record."""
import numpy as np

from .couple import Point


class Action:
    size = 4          # shoot, thrust, pointing.x, pointing.y
    shape = (4, 1)    # column-vector form accepted by `vector=`

    def __init__(self, shoot=False, thrust=False, pointing=None, vector=None):
        if vector is None:
            if not pointing:
                raise Exception("pointing argument must be specified.")
            self.shoot, self.thrust, self.pointing = shoot, thrust, pointing
            self.vector = None
            self.toVector()
        else:
            self.vector = vector
            self.fromVector(vector)

    def toVector(self):
        """(4,1) int column; `.vector` keeps the squeezed (4,) form like the reference."""
        col = np.array([[int(self.shoot)], [int(self.thrust)], [self.pointing.x], [self.pointing.y]], dtype=int)
        self.vector = col[:, 0].copy()
        return col

    def fromVector(self, vector):
        if vector.shape != Action.shape:
            raise Exception("Invalid vector : expected shape {} but got shape {}.".format(Action.shape, vector.shape))
        flat = vector[:, 0]
        self.shoot, self.thrust = bool(flat[0]), bool(flat[1])
        self.pointing = Point(flat[2], flat[3])

    def packed(self):
        """(valid, shoot, thrust, px, py) for the engine."""
        return 1, int(bool(self.shoot)), int(bool(self.thrust)), int(self.pointing.x), int(self.pointing.y)

    def __repr__(self):
        return "Action({0})".format(str(self.vector))

    __str__ = __repr__
