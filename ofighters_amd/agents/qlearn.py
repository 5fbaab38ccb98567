"""QlearnIA - the bi-head policy agent, forward only (API mirror of the reference's agents/qlearnIA_V2.py:324-456
control flow incl. Trainer.remember / the memory deque; the DQN replay / fit / save of :240-298 is out of scope,
SURVEY section 8f rank 3).  The batched, device-resident form of the memory is ofx_replay_* (include/ofx.h).

`play(obs)`: nothing once done; the first `collecting_steps` (20) total steps are `random_play()`; afterwards
epsilon-greedy over the device forward (`Trainer.get_best_action`, :199-235): `iaction = argmax(act_values)`,
`ipointer = (x, y)` of the heat-map arg-max.  The action always has exactly one of shoot / thrust set and the
pointer set (:447-454).
"""
import random
from collections import deque

import numpy as np

from .agent import Agent
from .policy_weights import synthetic
from ..lib.action import Action
from ..lib.epsilon import Epsilon_cos
from ..lib.observation import DEFAULT_HEIGHT, DEFAULT_WIDTH


def random_play():
    """qlearnIA_V2.py:317-321"""
    iaction = random.randint(0, 1)
    ipointer = (random.randint(0, DEFAULT_WIDTH - 1), random.randint(0, DEFAULT_HEIGHT - 1))
    return [iaction, ipointer]


class Trainer:
    """Holds what the forward needs: the weight blob and the exploration schedule shared by every QlearnIA
    (the reference's module-level TRAINER singleton, qlearnIA_V2.py:308-310)."""

    def __init__(self, weights=None, epsilon=None, memory_size=400):
        self.memory = deque(maxlen=memory_size)      # qlearnIA_V2.py:58
        self.weights = synthetic() if weights is None else np.ascontiguousarray(weights, np.float32)
        self.epsilon = epsilon if epsilon is not None else Epsilon_cos(period=110 * 400)
        self.act_values = None
        self.ptr_values = None

    def decay_epsilon(self):
        self.epsilon.next()

    def save(self, id=None, overwrite=False, folder="networks", name=None):
        """qlearnIA_V2.py:289-298 (`keras-model-<name>[-<id>]` under the networks folder).  The file is an .npz of
        the tensors in `model.get_weights()` order (policy_weights.to_keras): h5py / Keras are not needed to write
        it and `model.set_weights(list(np.load(f).values()))` reads it back on the Keras side."""
        import os
        from .policy_weights import save_npz
        fname = "keras-model-" + (name or "ofx") + ("-" + str(id) if id else "") + ".npz"
        os.makedirs(folder, exist_ok=True)
        path = os.path.join(folder, fname)
        if os.path.exists(path) and not overwrite:
            raise Exception("%s exists (overwrite=False)" % path)
        save_npz(path, self.weights)
        return path

    def load(self, path):
        """Weights exported from the reference's Keras model: np.savez(path, *model.get_weights())."""
        from .policy_weights import load_npz
        self.weights = load_npz(path)

    def remember(self, state, iaction, ipointer, reward, next_state, done):
        """qlearnIA_V2.py:237-238"""
        self.memory.append([state, iaction, ipointer, reward, next_state, done])

    def get_best_action(self, obs, rand=True):
        if rand and np.random.rand() <= self.epsilon.get():
            return random_play()
        out = obs.battleground._policy_forward(obs._ship, self.weights)
        self.act_values = out["act"]
        return [int(out["iaction"]), (int(out["ipointer"][0]), int(out["ipointer"][1]))]


TRAINER = None


def shared_trainer():
    global TRAINER
    if TRAINER is None:
        TRAINER = Trainer()
    return TRAINER


class QlearnIA(Agent):
    max_id = 1

    def __init__(self, trainer=None, is_learning=False):
        super().__init__(behavior="QlearnIA", bot=self)
        self.id = QlearnIA.max_id
        QlearnIA.max_id += 1
        self.done = False
        self.is_learning = is_learning      # True only advances the epsilon schedule here (no replay)
        self.trainer = trainer if trainer is not None else shared_trainer()
        self.collecting_steps = 20
        self.previous_obs = self.previous_action = self.previous_pointer = None
        self.epsilons = []

    def reset(self):
        super().reset()
        if self.id == 1:
            self.epsilons.append(self.trainer.epsilon.get())
        self.done = False
        self.previous_obs = self.previous_action = self.previous_pointer = None

    def play(self, obs):
        if self.done:
            return None
        if obs.done:
            self.done = True
        if self.previous_obs is not None and self.previous_action is not None and self.previous_pointer is not None:
            self.trainer.remember(self.previous_obs, self.previous_action, self.previous_pointer, obs.reward, obs,
                                  obs.done)
        if self.total_steps < self.collecting_steps:
            iaction, ipointer = random_play()
        else:
            iaction, ipointer = self.trainer.get_best_action(obs)
            if self.is_learning and self.id == 1:
                self.trainer.decay_epsilon()
        self.previous_obs, self.previous_action, self.previous_pointer = obs, iaction, ipointer
        act_vector = np.zeros((Action.size, 1))
        act_vector[iaction] = 1
        act_vector[2] = ipointer[0]
        act_vector[3] = ipointer[1]
        return Action(vector=act_vector)
