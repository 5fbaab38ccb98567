"""Exploration schedules (API mirror of the reference's lib/epsilon.py:36-86).

Host-side scalars: `Epsilon_cos(period)` walks one raised-cosine period from 1 to 0 and back,
`Epsilon_decay()` multiplies by 0.9999 per step until it drops below the soft floor 0.01.
Both expose `next()`, `get()`, `set(value)`.
"""
import math


def _raised_cosine(t, period, amplitude):
    return amplitude * ((math.cos((t / period) * 2 * math.pi) + 1) / 2)


def _raised_cosine_inverse(value, period, amplitude):
    return period * math.acos(((2 * value) / amplitude) - 1) / (2 * math.pi)


def _check_unit_interval(value):
    if value > 1.0 or value < 0.0:
        raise Exception("Value must me in range [0,1]")   # sic: the reference's message


class Epsilon_cos:
    def __init__(self, period):
        self.t = 0
        self.amplitude = 1
        self.period = period
        self.epsilon = _raised_cosine(self.t, self.period, self.amplitude)

    def next(self):
        self.t = (self.t + 1) % self.period
        self.epsilon = _raised_cosine(self.t, self.period, self.amplitude)
        return self.epsilon

    def get(self):
        return self.epsilon

    def set(self, value):
        _check_unit_interval(value)
        self.epsilon = value
        self.t = _raised_cosine_inverse(value, self.period, self.amplitude)


class Epsilon_decay:
    def __init__(self):
        self.epsilon = 1
        self.epsilon_min = 0.01
        self.decay = 0.99990

    def next(self):
        if self.epsilon > self.epsilon_min:   # below the floor the value is left alone
            self.epsilon *= self.decay
        return self.epsilon

    def get(self):
        return self.epsilon

    def set(self, value):
        _check_unit_interval(value)
        self.epsilon = value
