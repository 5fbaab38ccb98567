"""DeviceTrainer - the batched, device-resident counterpart of the reference's `Trainer` (agents/qlearnIA_V2.py:46-298)
for an ArenaBatch: weights, Adam state and the replay memory live in HBM; `get_best_action` is
ArenaBatch.policy_forward + policy_explore, `remember` is ArenaBatch.replay_capture, and `replay(batch_size)`
(:240-285) is sample -> gather -> targets -> one fit step, all through the C-ABI (ofx_replay_sample, ofx_replay_gather,
ofx_dqn_targets, ofx_dqn_fit - or ofx_dqn_fit_reference with `reference_quirks=True`: the reference's step as written,
ptr_target[x][y] and the fit on next_state's inputs included, :280-283)."""
import numpy as np

from .engine import DeviceBuffer
from .lib.epsilon import Epsilon_cos


class DeviceTrainer:
    def __init__(self, batch, weights, learning_rate=0.0001, epsilon=None, batch_size=8, memory_size=400, frames=0,
                 seed=0x0F160003, fit_batch=256, reference_quirks=False):
        self.batch = batch                                  # the ArenaBatch this trainer plays and learns on
        w = np.ascontiguousarray(weights, np.float32)
        self.n_floats = w.size
        self.weights = DeviceBuffer(w.nbytes).upload(w)
        zeros = np.zeros_like(w)
        self.adam_m = DeviceBuffer(w.nbytes).upload(zeros)
        self.adam_v = DeviceBuffer(w.nbytes).upload(zeros)
        self.learning_rate = learning_rate                  # lr = 0.0001 (qlearnIA_V2.py:306)
        self.gamma = 0.9                                    # :51
        self.epsilon = epsilon if epsilon is not None else Epsilon_cos(period=110 * 400)
        self.batch_size = batch_size                        # 8 (:307)
        self.seed = seed
        self.fit_batch = fit_batch                          # rows per optimisation step (the reference fits on 8): one
                                                            # replay takes 3.6 ms at 64 rows, 5.8 at 256 - about one
                                                            # lock-step of a 4096-arena batch -, 42 ms at 4096
        self.reference_quirks = bool(reference_quirks)      # Trainer.replay as written instead of the textbook DQN step
        self.fit_steps = 0
        self.draws = 0
        self.losses = []
        self._buf = {}                                       # replay scratch kept between calls (grow-only)
        batch.replay_create(memory_size, frames)

    def _scratch(self, name, nbytes):
        """A device buffer of at least nbytes that lives as long as the trainer: no hipMalloc / hipFree per replay."""
        b = self._buf.get(name)
        if b is None or b.nbytes < nbytes:
            if b is not None:
                self.batch.sync()
                b.free()
            b = self._buf[name] = DeviceBuffer(int(nbytes))
        return b

    def decay_epsilon(self):
        self.epsilon.next()

    def weights_host(self):
        self.batch.sync()
        return self.weights.download(np.float32, (self.n_floats,))

    def save(self, id=None, overwrite=False, folder="networks", name="bi_head_pointer"):
        """Trainer.save (qlearnIA_V2.py:289-298): `keras-model-<name>[-<id>]` under the networks folder, as the .npz
        of `model.get_weights()` (agents/policy_weights.py) that the facade's Trainer.load and a Keras
        `model.set_weights(list(np.load(f).values()))` read back."""
        import os
        from .agents.policy_weights import save_npz
        fname = "keras-model-" + name + ("-" + str(id) if id else "") + ".npz"
        os.makedirs(folder, exist_ok=True)
        path = os.path.join(folder, fname)
        if os.path.exists(path) and not overwrite:
            raise Exception("%s exists (overwrite=False)" % path)
        save_npz(path, self.weights_host())
        return path

    def replay(self, batch_size=None):
        """One Trainer.replay: a minibatch of min(batch_size, len(memory)) rows per arena, targets, one fit step.
        Returns (mse(output1), mse(output2)) or None while every memory is still empty."""
        b = self.batch
        bs = int(batch_size or self.batch_size)
        cnt, _ = b.replay_count()
        if int(cnt.max()) == 0:
            return None
        slot, n_s = b.replay_sample(self.seed, self.draws, bs, self._scratch("slot", 4 * b.N * bs), self._scratch("n_s", 4 * b.N))
        self.draws += 1
        # the sampled transitions of all arenas, WITHOUT the -1 pads of arenas that hold fewer than bs (a pad would enter
        # the BatchNorm batch statistics and the loss scale of the fit; the reference's batch is min(bs, len(memory)) real
        # rows); one optimisation step takes a window of them that moves with the draw counter
        # n_sampled, not min(len, bs): a row whose `state` frame has left the arena's frame ring is not sampled
        b.sync()
        n_valid = int(n_s.download(np.int32, (b.N,)).sum())
        if n_valid == 0:
            return None
        n = min(n_valid, int(self.fit_batch))
        start = ((self.draws - 1) * n) % (n_valid - n + 1)
        words = b.W * b.H // 32
        rows = self._scratch("rows", n * b.TRANSITION_DTYPE.itemsize)
        bits_prev, bits_next = self._scratch("bits_prev", 4 * n * 2 * words), self._scratch("bits_next", 4 * n * 2 * words)
        got = b.replay_gather_valid_into(slot, n_s, bs, start, n, rows, bits_prev, bits_next)
        if got != n:
            raise Exception("DeviceTrainer.replay: gathered %d of %d rows" % (got, n))
        rows_p, prev_p, next_p = rows.ptr, bits_prev.ptr, bits_next.ptr
        if self.reference_quirks:
            self.fit_steps += 1
            loss = b.dqn_fit_reference(self.weights, self.adam_m, self.adam_v, self.fit_steps, self.learning_rate, n,
                                       rows_p, prev_p, next_p, self.gamma)
            self.losses.append(loss)
            return loss
        y_act, y_ptr = self._scratch("y_act", 4 * n), self._scratch("y_ptr", 4 * n)
        from . import _native as nat
        # q_sa / p_sp (the current values at the chosen action / pointer) are not asked for: the fit's own training-mode
        # forward produces them, so the targets need the forward on next_state only
        nat.check(nat.lib().ofx_dqn_targets(b.handle, self.weights.ptr, n, rows_p, prev_p, next_p, float(self.gamma),
                                             None, None, y_act.ptr, y_ptr.ptr))
        self.fit_steps += 1
        loss = b.dqn_fit(self.weights, self.adam_m, self.adam_v, self.fit_steps, self.learning_rate, n, rows_p, prev_p,
                         y_act.ptr, y_ptr.ptr)
        self.losses.append(loss)
        return loss
