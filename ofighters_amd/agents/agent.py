"""Agent: per-ship bookkeeping + the bot plug-in protocol + the scripted bots.

API mirror of the reference's agents/agent.py:17-155: `Agent(behavior=None,
bot=None)`, attributes `score, scores, reward, steps, total_steps, episode,
behavior, bot`, methods `step(obs)` / `reset()`, and the six behaviour names.
A plug-in bot is any object with `.play(obs)` returning an Action or None.

The scripted bots draw from Python's `random` in the same order as the
reference, so a seeded facade run consumes the Mersenne-Twister stream exactly
like the reference does (tests replay the golden traces through it).
"""
import random as _random

import numpy as np

from ..lib.action import Action
from ..lib.couple import Point


def _repoint(obs):
    return Point(_random.randint(0, obs.dim.x), _random.randint(0, obs.dim.y))


def _idle(obs):            # agent.py:99-104
    return Action(pointing=obs.pointing)


def _thrust(obs):          # agent.py:107-112
    return Action(thrust=True, pointing=obs.pointing)


def _shoot(obs):           # agent.py:115-120
    return Action(shoot=True, pointing=obs.pointing)


def _random_play(obs):     # agent.py:123-133: one of three, re-point draws only when chosen
    pick = _random.choice(["shoot", "thrust", "pointing"])
    target = _repoint(obs) if pick == "pointing" else obs.pointing
    return Action(shoot=pick == "shoot", thrust=pick == "thrust", pointing=target)


def _turret(obs):          # agent.py:136-144: draw order shoot, then re-point test, then the point
    fire = _random.random() < 0.8
    target = _repoint(obs) if _random.random() < 0.3 else obs.pointing
    return Action(shoot=fire, pointing=target)


def _runner(obs):          # agent.py:147-155
    go = _random.random() < 0.9
    target = _repoint(obs) if _random.random() < 0.1 else obs.pointing
    return Action(thrust=go, pointing=target)


BEHAVIOURS = {None: _idle, "idle": _idle, "random": _random_play, "turret": _turret, "runner": _runner,
              "thrust": _thrust, "shoot": _shoot}


class Agent:
    def __init__(self, behavior=None, bot=None):
        self.score, self.reward = 0, 0
        self.scores = []
        self.steps = self.total_steps = self.episode = 0
        self.behavior = behavior
        self.obs_vector = np.array([])
        self.act_vector = np.array([])
        self.bot = bot if bot else None
        if self.bot is not None:
            self.bot_play = self.bot.play
        elif (behavior or None) in BEHAVIOURS:
            self.bot_play = BEHAVIOURS[behavior or None]
        else:
            raise Exception("You must give a bot in parameter or select an existing behavior.")

    def reset(self):
        """Episode boundary: bank the score (agent.py:59-64); `reward` is left alone."""
        self.scores.append(self.score)
        self.score = 0
        self.steps = 0
        self.episode += 1

    def step(self, obs):
        """Called for every ship every tick, dead ones included (ship.py:260-262)."""
        self.steps += 1
        self.total_steps += 1
        self.score += self.reward      # agent.py:73-74; libofx does the same inside ofx_step
        self.reward = 0
        self.obs_vector = obs._vector_or_none()
        action = self.bot_play(obs)
        if not action:
            return None
        self.act_vector = action.vector
        return action
