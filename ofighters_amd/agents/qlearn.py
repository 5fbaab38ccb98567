"""QlearnIA - the bi-head policy agent and its Trainer (API mirror of the reference's agents/qlearnIA_V2.py:46-456:
control flow of `play`, `Trainer.remember` / the memory deque, `Trainer.replay` on the reference's schedule, `save`).
The arithmetic - forward, TD targets, fit - runs in libofx on the GPU through the facade's engine; the batched,
device-resident form of the same agent is ofighters_amd.rollout.TrainingRollout + ofighters_amd.trainer.DeviceTrainer.

`play(obs)`: nothing once done; a learning agent replays when it sees its death (:375-378); the first
`collecting_steps` (20) total steps are `random_play()`; afterwards epsilon-greedy over the device forward
(`Trainer.get_best_action`, :199-235): `iaction = argmax(act_values)`, `ipointer = (x, y)` of the heat-map arg-max;
agent 1 also replays every 50 total steps and snapshots every `snapshot` episodes (:412-417).  The action always has
exactly one of shoot / thrust set and the pointer set (:447-454).
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

    def __init__(self, weights=None, epsilon=None, memory_size=400, learning_rate=0.0001, batch_size=8,
                 reference_quirks=True, name="ofx"):
        self.memory = deque(maxlen=memory_size)      # qlearnIA_V2.py:58
        self.weights = synthetic() if weights is None else np.ascontiguousarray(weights, np.float32)
        self.epsilon = epsilon if epsilon is not None else Epsilon_cos(period=110 * 400)
        self.gamma = 0.9                             # :60
        self.learning_rate = learning_rate           # :306
        self.batch_size = batch_size                 # :307
        self.name = name
        # True: Trainer.replay as the reference wrote it (ptr_target[x][y], fit on next_state's inputs, :279-283);
        # False: the textbook DQN step (include/ofx.h: ofx_dqn_fit_reference / ofx_dqn_fit)
        self.reference_quirks = bool(reference_quirks)
        self.act_values = None
        self.ptr_values = None
        self.fit_steps = 0
        self._dev = None                             # (weights, adam_m, adam_v) DeviceBuffers once replay has run

    def decay_epsilon(self):
        self.epsilon.next()

    def save(self, id=None, overwrite=False, folder="networks", name=None):
        """qlearnIA_V2.py:289-298 (`keras-model-<name>[-<id>]` under the networks folder).  The file is an .npz of
        the tensors in `model.get_weights()` order (policy_weights.to_keras): h5py / Keras are not needed to write
        it and `model.set_weights(list(np.load(f).values()))` reads it back on the Keras side."""
        import os
        from .policy_weights import save_npz
        fname = "keras-model-" + (name or self.name) + ("-" + str(id) if id else "") + ".npz"
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
        self._dev = None

    def remember(self, state, iaction, ipointer, reward, next_state, done):
        """qlearnIA_V2.py:237-238"""
        self.memory.append([state, iaction, ipointer, reward, next_state, done])

    def replay(self, batch_size):
        """qlearnIA_V2.py:240-287: random.sample of the memory, the two predictions per transition, the targets, one
        `model.fit` step - on the device, through the engine of the battleground the observations came from.
        Returns an object with `.history['loss']` like Keras' fit (total loss = mse(output1) + mse(output2))."""
        batch_size = min(batch_size, len(self.memory))
        if batch_size == 0:
            raise ValueError("Sample larger than population or is negative")   # random.sample on an empty deque
        minibatch = random.sample(self.memory, batch_size)
        engine = minibatch[0][0].battleground._e
        if not hasattr(engine, "dqn_fit_reference"):
            raise Exception("Trainer.replay needs the HIP engine (ArenaBatch): there is no CPU fit")
        from ..engine import DeviceBuffer
        n = batch_size
        rows = np.zeros(n, engine.TRANSITION_DTYPE)
        words = DEFAULT_WIDTH * DEFAULT_HEIGHT // 32
        bits = np.zeros((2, n, 2, words), np.uint32)
        for i, (obs, iaction, ipointer, reward, next_obs, done) in enumerate(minibatch):
            rows[i]["ship"] = 0
            rows[i]["iaction"], rows[i]["px"], rows[i]["py"] = iaction, ipointer[0], ipointer[1]
            rows[i]["reward"], rows[i]["done"] = reward, int(bool(done))
            rows[i]["head_prev"], rows[i]["head_next"] = obs.head(), next_obs.head()
            for k, o in enumerate((obs, next_obs)):
                for c, m in enumerate((o.ship_map, o.laser_map)):   # pixel p = y*W + x -> bit p & 31 of word p >> 5
                    bits[k, i, c] = np.packbits(np.asarray(m) != 0, bitorder="little").view(np.uint32)
        if self._dev is None:
            z = np.zeros_like(self.weights)
            self._dev = tuple(DeviceBuffer(z.nbytes).upload(a) for a in (self.weights, z, z))
        w, m_, v_ = self._dev
        engine.sync()
        d_rows = DeviceBuffer(rows.nbytes).upload(rows)
        d_prev, d_next = DeviceBuffer(bits[0].nbytes).upload(bits[0]), DeviceBuffer(bits[1].nbytes).upload(bits[1])
        self.fit_steps += 1
        if self.reference_quirks:
            loss = engine.dqn_fit_reference(w, m_, v_, self.fit_steps, self.learning_rate, n, d_rows.ptr, d_prev.ptr,
                                            d_next.ptr, self.gamma)
        else:
            from .. import _native as nat
            outs = [DeviceBuffer(4 * n) for _ in range(4)]
            nat.check(nat.lib().ofx_dqn_targets(engine.handle, w.ptr, n, d_rows.ptr, d_prev.ptr, d_next.ptr,
                                                 float(self.gamma), *[o.ptr for o in outs]))
            loss = engine.dqn_fit(w, m_, v_, self.fit_steps, self.learning_rate, n, d_rows.ptr, d_prev.ptr,
                                  outs[2].ptr, outs[3].ptr)
        engine.sync()
        self.weights = w.download(np.float32, self.weights.shape)

        class _History:
            history = {"loss": [loss[0] + loss[1]], "output1_loss": [loss[0]], "output2_loss": [loss[1]]}
        return _History()

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

    def __init__(self, trainer=None, is_learning=True):
        super().__init__(behavior="QlearnIA", bot=self)
        self.id = QlearnIA.max_id
        QlearnIA.max_id += 1
        self.done = False
        self.is_learning = is_learning      # the reference's default (:345); False = forward only (no replay / fit)
        self.trainer = trainer if trainer is not None else shared_trainer()
        self.batch_size = 8                 # :337
        self.losses = []
        self.collecting_steps = 20
        self.snapshot = 50                  # save the model every N episodes (:353)
        self.snapshot_folder = None         # None: the reference's NETWORKS_FOLDER write is skipped
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
            if self.is_learning and len(self.trainer.memory):
                self.losses.append(self.trainer.replay(self.batch_size).history["loss"][0])   # :376-378
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
        if self.is_learning and self.id == 1:       # all bots share the trainer: replay / save once (:412-417)
            if self.total_steps % 50 == 0 and len(self.trainer.memory):
                self.losses.append(self.trainer.replay(self.batch_size).history["loss"][0])
            if self.episode > 0 and self.episode % self.snapshot == 0 and self.steps < 2 and self.snapshot_folder:
                self.trainer.save(id="iteration-%s" % self.episode, overwrite=True, folder=self.snapshot_folder)
        act_vector = np.zeros((Action.size, 1))
        act_vector[iaction] = 1
        act_vector[2] = ipointer[0]
        act_vector[3] = ipointer[1]
        return Action(vector=act_vector)
