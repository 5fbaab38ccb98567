/*
 * ofx.h - C-ABI of the MI355X-native batched Ofighters arena engine.
 *
 * This is the drop-in boundary for the hot path named in BASELINE.json
 * (north_star): Battleground.frame() = request_actions -> generate_frame ->
 * Observation(battleground) for thousands of independent arenas in lock-step,
 * plus the policy forwards.  The reference has no FFI of its own (it is pure
 * Python); each entry point below cites the reference interface it replaces
 * (paths relative to /root/reference/ofighters).  INTEGRATION.md shows the
 * ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes only; no C++ / torch types.
 *  - Every bulk pointer is a DEVICE (HBM) pointer unless the name ends in
 *    `_host`.  ofx_malloc/ofx_free/ofx_memcpy_* let a host without torch own
 *    device buffers; a torch tensor's data_ptr() is equally valid.
 *  - All kernels are enqueued on the handle's HIP stream and return
 *    immediately; ofx_sync() waits.  Entry points that copy to `_host`
 *    pointers synchronise the stream themselves.
 *  - Return value: 0 = OFX_OK, <0 = error; ofx_last_error() returns the
 *    thread-local message (the reference raises bare Exception(msg):
 *    lib/battleground.py:30, lib/action.py:40,63, agents/agent.py:51).
 *  - There is NO CPU fallback in this library: without a HIP device every
 *    compute entry point fails with OFX_ERR_NO_DEVICE.
 *  - A handle is thread-compatible, not thread-safe (the reference is single
 *    threaded: lib/ofighters.py:697).
 */
#ifndef OFX_H
#define OFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OFX_VERSION 1

enum {
  OFX_OK = 0,
  OFX_ERR_INVALID = -1,   /* bad argument / config                         */
  OFX_ERR_NO_DEVICE = -2, /* no HIP device: there is no CPU fallback       */
  OFX_ERR_HIP = -3,       /* a HIP runtime call failed (message has detail) */
  OFX_ERR_STATE = -4,     /* call order violated (e.g. step before spawn)  */
  OFX_ERR_OVERFLOW = -5   /* laser capacity exceeded since the last check  */
};

/* One POD with the reference's module-level constants as defaults
 * (ofx_default_config fills them in):
 *   width/height 400           lib/observation.py:10-11
 *   ship_radius 8, hull 1      lib/ship.py:43-45
 *   laser_radius 2             lib/laser.py:23
 *   ship_speed 8               lib/ship.py:24
 *   laser_speed 10 * LIGHT 1   lib/ship.py:86, lib/laser.py:13,24
 *   rewards death 0 kill 0 aim 2 trajectory 1   agents/qlearnIA_V2.py:39-44
 *   episode_ticks 200          lib/ofighters.py:59 (MAX_TIME)               */
typedef struct ofx_config {
  int32_t n_arenas;     /* N  arenas advanced in lock-step on this device   */
  int32_t n_ships;      /* M  ships per arena, 1..64                        */
  int32_t laser_cap;    /* L  laser slots per arena (multiple of 64)        */
  int32_t width, height;
  int32_t ship_radius, laser_radius;
  int32_t ship_speed, laser_speed;
  int32_t reward_death, reward_kill, reward_aim, reward_trajectory;
  int32_t episode_ticks;
  int32_t device;       /* HIP device ordinal                               */
  int32_t arena_base;   /* global id of local arena 0 (multi-GPU sharding:
                           keys the counter RNG so results do not depend on
                           the number of GPUs)                              */
} ofx_config;

/* One ship's action for one tick == lib/action.py:12-56 Action
 * (shoot, thrust, pointing) ; valid==0 is the reference's `None`
 * (lib/ship.py:308: no-op).  Array layout [N][M].                           */
typedef struct ofx_action {
  int32_t px, py;
  uint8_t shoot, thrust, valid, _pad;
} ofx_action;

/* scripted behaviours of agents/agent.py:38-51 (+ "none" = always None)     */
enum {
  OFX_BOT_IDLE = 0,   /* idlebot          agent.py:99-104  */
  OFX_BOT_RANDOM = 1, /* random_play      agent.py:123-133 */
  OFX_BOT_TURRET = 2, /* crazy_turret     agent.py:136-144 */
  OFX_BOT_RUNNER = 3, /* crazy_runner     agent.py:147-155 */
  OFX_BOT_THRUST = 4, /* never_back_down  agent.py:107-112 */
  OFX_BOT_SHOOT = 5   /* mass_shooter     agent.py:115-120 */
};

/* state fields readable through ofx_get_host / ofx_device_ptr               */
enum {
  OFX_F_SHIP_X = 0,      /* int32 [N][M]   Ship.body.x            ship.py:43   */
  OFX_F_SHIP_Y = 1,      /* int32 [N][M]                                       */
  OFX_F_SHIP_PX = 2,     /* int32 [N][M]   Ship.pointing.x        ship.py:53   */
  OFX_F_SHIP_PY = 3,     /* int32 [N][M]                                       */
  OFX_F_SHIP_ALIVE = 4,  /* uint8 [N][M]   Ship.is_playable()     ship.py:108  */
  OFX_F_REWARD = 5,      /* int32 [N][M]   Agent.reward           agent.py:25  */
  OFX_F_SCORE = 6,       /* int32 [N][M]   Agent.score            agent.py:23  */
  OFX_F_N_LASERS = 7,    /* int32 [N]      len(Battleground.lasers)            */
  OFX_F_LASER_X = 8,     /* float64 [N][L] Laser.body.x           laser.py:23  */
  OFX_F_LASER_Y = 9,     /* float64 [N][L]                                     */
  OFX_F_LASER_OWNER = 10,/* uint8 [N][L]   index of Laser.owner in ships       */
  OFX_F_LASER_DEAD = 11, /* uint8 [N][L]   Laser.state=="destroyed" laser.py:66 */
  OFX_F_KILLER = 12,     /* int16 [N][M]   list index of the laser that killed
                            the ship THIS tick, -1 otherwise (laser.py:52-60) */
  OFX_F_TIME = 13,       /* int32 [N]      Battleground.time  battleground.py:154 */
  OFX_F_LAST_SCORES = 14,/* int32 [N][M]   Agent.scores[-1]       agent.py:62  */
  OFX_F_HULL = 15,       /* int32 [N][M]   Ship.hull              ship.py:45,129 */
  OFX_F_LASER_DX = 16,   /* float64 [N][L] per-tick displacement  laser.py:45  */
  OFX_F_LASER_DY = 17,   /* float64 [N][L]                                     */
  OFX_F_OBS_REWARD = 18, /* int32 [N][M]   reward as seen by Agent.step this
                            tick (obs.reward, observation.py:103)             */
  OFX_F_COUNT = 19
};

/* observation map element types for ofx_rasterise                           */
enum {
  OFX_MAP_U8 = 0,   /* 1 byte / cell  (default; 320 000 B per arena)         */
  OFX_MAP_F32 = 1,
  OFX_MAP_F64 = 2,  /* the reference's dtype (observation.py:86,91)          */
  OFX_MAP_BITS = 3  /* 1 bit / cell, MSB-first per byte == numpy.packbits;
                       row stride = ceil(W/8)*... see ofx_map_bytes          */
};

typedef struct ofx_handle ofx_handle;

/* ---- library ---------------------------------------------------------- */
int ofx_version(void);
const char *ofx_last_error(void);
int ofx_device_count(void); /* 0 when no HIP device is visible              */
void ofx_default_config(ofx_config *cfg);

/* ---- device memory helpers (host side needs no torch) ------------------ */
int ofx_malloc(void **dev_ptr, size_t bytes);
int ofx_free(void *dev_ptr);
int ofx_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes);
int ofx_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes);

/* ---- lifetime ----------------------------------------------------------
 * replaces Battleground.__init__ (lib/battleground.py:13-106) for N arenas   */
int ofx_create(const ofx_config *cfg, ofx_handle **out);
int ofx_destroy(ofx_handle *h);
int ofx_sync(ofx_handle *h);
void *ofx_stream(ofx_handle *h);             /* the hipStream_t of the handle */
int ofx_set_stream(ofx_handle *h, void *hip_stream); /* adopt caller's stream */

/* ---- spawn / restart ---------------------------------------------------
 * ofx_spawn: ships at draws[N][M][2] (x,y), pointing = position, alive,
 *   reward = score = 0, no lasers            (battleground.py:79-81, ship.py:35-58)
 * ofx_restart: Battleground.restart (battleground.py:108-117) -> Ship.reset
 *   (ship.py:92-106) -> Agent.reset (agent.py:59-64) with their quirks:
 *   pointing = OLD position, `x or old` keeps the coordinate when the draw is 0,
 *   hull not restored, reward NOT cleared, score appended then zeroed.
 *   arena_mask[N] (uint8, may be NULL = all) selects the arenas to restart.
 * draws are the reference's randint(0, dim) values (inclusive: may equal W/H). */
int ofx_spawn(ofx_handle *h, const int32_t *draws);
int ofx_restart(ofx_handle *h, const int32_t *draws, const uint8_t *arena_mask);
/* same, draws generated on device by the counter RNG (seed, episode)         */
int ofx_spawn_random(ofx_handle *h, uint64_t seed);
int ofx_restart_random(ofx_handle *h, uint64_t seed, uint32_t episode);

/* ---- the tick ----------------------------------------------------------
 * ofx_step = Agent.step bookkeeping for every ship, dead included
 *            (score += reward; reward = 0: agents/agent.py:66-74, ship.py:260-262)
 *          + GUI laser clean-up of lasers destroyed in the previous tick
 *            (lib/ofighters.py:619-625,702-707)
 *          + Battleground.generate_frame(actions) (battleground.py:153-160):
 *            Laser.move for every laser in list order (laser.py:36-62), then
 *            Ship.move(action) in index order (ship.py:303-339: pointing,
 *            thrust ship.py:213-222, shoot ship.py:134-156 + form.py:159-188,
 *            aim / trajectory rewards ship.py:158-210).
 * actions: device [N][M] ofx_action.                                         */
int ofx_step(ofx_handle *h, const ofx_action *actions);

/* ---- observation -------------------------------------------------------
 * ofx_rasterise = Observation.analyse_battleground (lib/observation.py:79-95)
 *   with Circle.binary_draw -> skimage.draw.disk((y,x), r, shape) arithmetic
 *   (lib/form.py:222-228; scikit-image 0.18.3 draw.py:11-43,46-143).
 *   Maps are [N][W rows = y][H cols = x]; playable ships (r=ship_radius) into
 *   ship_map, every listed laser incl. just-destroyed (r=laser_radius) into
 *   laser_map.  NULL output pointers select the handle's internal buffers
 *   (see ofx_map_ptr).
 * ofx_observe_head = Observation.analyse_ship + toVector head
 *   (observation.py:101-123): float64 [N][M][8] =
 *   reward, can_shoot(1), pointing.x, pointing.y, dim.x, dim.y, pos.x, pos.y;
 *   done[N][M] uint8 = not playable (may be NULL).                           */
int ofx_rasterise(ofx_handle *h, int map_type, void *ship_map, void *laser_map);
size_t ofx_map_bytes(const ofx_handle *h, int map_type); /* bytes of ONE arena's ONE map */
void *ofx_map_ptr(ofx_handle *h, int map_type, int which /*0 ship,1 laser*/);
int ofx_observe_head(ofx_handle *h, double *head, uint8_t *done);

/* ---- scripted bots on device -------------------------------------------
 * behaviours[M] (host, OFX_BOT_*) ; actions out device [N][M].  Draws come
 * from Philox4x32-10 keyed by (seed) with counter (global arena, ship, tick),
 * so any GPU count reproduces the same stream.  Same action LAW as
 * agents/agent.py:99-155 (not the Mersenne-Twister stream).                  */
int ofx_bot_actions(ofx_handle *h, const int32_t *behaviours_host, uint64_t seed,
                    uint32_t tick, ofx_action *actions);

/* ---- the headless loop ---------------------------------------------------
 * Battleground.run (lib/battleground.py:169-173) for arenas flown by the scripted bots: n_ticks lock-steps of
 * request_actions (the ofx_bot_actions law, counter ticks tick0 .. tick0 + n_ticks - 1) -> generate_frame (ofx_step)
 * -> Observation(battleground) (ofx_rasterise into the handle's maps of observe_map_type, or -1 = no maps) enqueued by
 * ONE host call; state afterwards is bit-identical to n_ticks x (ofx_bot_actions, ofx_step[, ofx_rasterise]).  The
 * bots' law is evaluated inside the step kernel, and without an observer between them all n_ticks lock-steps run
 * inside one launch (an arena belongs to one wavefront for the whole launch).  Episode ends stay with the caller
 * (ofx_restart* between two calls).  With ofx_policy_profile switched on, one event pair brackets the call's
 * dominant kernel (the K-tick step launch, or the last lock-step's rasteriser).                                  */
int ofx_rollout(ofx_handle *h, const int32_t *behaviours_host, uint64_t seed, uint32_t tick0, int32_t n_ticks,
                int32_t observe_map_type);

/* ---- state access ------------------------------------------------------ */
int ofx_get_host(ofx_handle *h, int field, void *dst_host, size_t bytes);
void *ofx_device_ptr(ofx_handle *h, int field);
size_t ofx_field_bytes(const ofx_handle *h, int field);
/* ---- zero-copy views of the handle's arrays -----------------------------
 * The reference's plugin seam is "any object with .play(obs)" (agents/agent.py:34-37); its learning agent reads
 * obs.ship_map / obs.laser_map / obs.vector[:8] and answers an Action (agents/qlearnIA_V2.py:206-220,447-454).  For a
 * batch the same seam is an EXTERNAL POLICY that reads the maps and the state of all N x M ships where they lie in HBM
 * and writes its [N][M] ofx_action array there: ofx_field_desc / ofx_map_desc describe a handle-owned array (pointer,
 * element type, shape, strides in elements, device) so that a host framework can wrap it WITHOUT a copy - a DLPack
 * capsule, __cuda_array_interface__, torch.as_tensor (ofighters_amd/engine.py: ArenaBatch.tensor / maps_tensor).
 * Ownership: the memory belongs to the handle and lives until ofx_destroy; the CONTENTS of a state field are
 * those of the last ofx_step / ofx_restart* / ofx_spawn*, of a map those of the last ofx_rasterise of that type, and the
 * next such call overwrites them in place.  Everything runs on ofx_stream(h): read and write the views on that stream
 * (or order yours against it), never concurrently with an entry point that writes them.                            */
enum { OFX_DT_U8 = 0, OFX_DT_I16 = 1, OFX_DT_I32 = 2, OFX_DT_I64 = 3, OFX_DT_F32 = 4, OFX_DT_F64 = 5 };
typedef struct ofx_tensor_desc {
  void *data;          /* device pointer                                    */
  int32_t dtype;       /* OFX_DT_*                                          */
  int32_t itemsize;    /* bytes per element                                 */
  int32_t ndim;        /* 1..4                                              */
  int32_t device;      /* HIP device ordinal of the handle                  */
  int64_t shape[4];
  int64_t stride[4];   /* in elements; dense row-major                      */
} ofx_tensor_desc;
/* state field `field` (OFX_F_*): [N][M], [N][L] or [N]                      */
int ofx_field_desc(ofx_handle *h, int field, ofx_tensor_desc *out);
/* map `which` (0 ship, 1 laser) of `map_type`: [N][W rows = y][H cols = x] of uint8 / float32 / float64, or
 * [N][W*H/8] uint8 for OFX_MAP_BITS; the handle's internal buffer (allocated at the first ofx_rasterise of that type
 * with NULL outputs, or here)                                               */
int ofx_map_desc(ofx_handle *h, int map_type, int which, ofx_tensor_desc *out);

/* number of lasers dropped because an arena's list was full since the last
 * call (never silent truncation); resets the counter.                        */
int ofx_overflow_count(ofx_handle *h, int64_t *count_host);

/* ---- episodic scores (the only cross-GPU quantity) ----------------------
 * Writes int64 [M+1] to a DEVICE buffer: sum over local arenas of the score
 * each ship slot banked at the most recent ofx_restart (Agent.reset:
 * scores.append(score), agents/agent.py:61-63) and, last, the arena count.
 * The caller all-reduces it (RCCL via torch.distributed).                    */
int ofx_episode_scores(ofx_handle *h, int64_t *sums);
/* The same followed by the all-reduce itself, for a host that is not torch: ncclAllReduce(sum, int64, in place) of the
 * [M+1] vector over the ranks of `nccl_comm` - an ncclComm_t the caller made with ncclCommInitRank, one rank per GPU
 * (RCCL over xGMI) - enqueued on the handle's stream.  libofx.so does not link RCCL; the symbol is resolved in the
 * process (or librccl.so.1 is opened) at the first call.  SURVEY 8b / 8e: the only collective of the path.          */
int ofx_scores_allreduce(ofx_handle *h, void *nccl_comm, int64_t *sums);

/* ---- scratch MLP forward -----------------------------------------------
 * Neural_network.feed (agents/neural_network.py:396-420): a <- sigmoid(W a + b)
 * per layer, float64, column vectors.  layers_host[n_layers] sizes;
 * weights: concatenated row-major W_i (n_{i+1} x n_i); biases concatenated;
 * x [batch][layers[0]] ; y [batch][layers[-1]] ; argmax int32 [batch] may be
 * NULL (Neural_network.max_sol_index, neural_network.py:423-429).            */
int ofx_scratch_feed(ofx_handle *h, const int32_t *layers_host, int32_t n_layers,
                     const double *weights, const double *biases, const double *x,
                     int32_t batch, double *y, int32_t *argmax);
/* same network fed with the live observation vector of every (arena, ship)
 * (toVector order, observation.py:119-125; needs layers[0] == 8 + 2*W*H):
 * the binary map tail is consumed as a sparse gather of W_1 columns.
 * y [N][M][layers[-1]].                                                      */
int ofx_scratch_feed_obs(ofx_handle *h, const int32_t *layers_host, int32_t n_layers,
                         const double *weights, const double *biases, double *y,
                         int32_t *argmax);

/* ---- bi-head policy forward --------------------------------------------
 * Trainer.pointer_model graph + inference glue
 * (agents/qlearnIA_V2.py:123-190, 206-220).  See ofx_policy.h-style layout
 * notes in DESIGN.md; weights are one float32 blob described by
 * ofx_policy_layout().                                                       */
typedef struct ofx_policy_desc {
  int32_t n_floats;        /* total length of the weight blob                 */
  int32_t offset[64];      /* start of each tensor, order given in DESIGN.md  */
  int32_t count[64];
  int32_t n_tensors;
} ofx_policy_desc;
int ofx_policy_layout(const ofx_handle *h, ofx_policy_desc *desc_host);
/* ship_mask[N][M] uint8 (NULL = every ship) selects which ships get a
 * forward; act_values float32 [N][M][2]; iaction int32 [N][M];
 * ipointer int32 [N][M][2] = (x, y) of the heat-map arg-max
 * (unravel_index(order='F'), qlearnIA_V2.py:218-220); heatmap float32
 * [N][M][W][H] or NULL (never materialised when NULL).                       */
int ofx_policy_forward(ofx_handle *h, const float *weights, const uint8_t *ship_mask,
                       float *act_values, int32_t *iaction, int32_t *ipointer,
                       float *heatmap);
/* Keras keeps a compiled model between predict() calls (the module-level TRAINER, agents/qlearnIA_V2.py:308); the
 * counterpart here: a forward first folds BatchNorm into the convolutions and builds its phase weights / tables from
 * the blob (two small launches).  Pinning a blob does that once: every following forward that names the same
 * pointer reuses the prepared weights, until another blob (or NULL) is pinned.  ofx_dqn_fit on a pinned blob
 * re-prepares behind its update; after changing a pinned blob any other way, pin it again.                     */
int ofx_policy_pin_weights(ofx_handle *h, const float *weights);
/* Diagnostic switches of the policy forward (value 0 / 1): results agree up to fp32 summation order.              */
#define OFX_OPT_TRUNK_PLAIN 1 /* the four trunk layers through the plain VALU convolution (test reference)        */
#define OFX_OPT_FRAMES_REF  2 /* frame lines of the head from the definition instead of the phase form (reference) */
#define OFX_OPT_TRUNK_FUSE  4 /* the streaming form of the trunk (conv1 -> conv2 fused in one persistent kernel, conv3 on
                                LDS-direct loads): 0 auto (when the images fill the CUs four times over), 1 always,
                                2 never; bit-identical results either way                                          */
/* NOT a diagnostic: which bilinear UpSampling2D((2,2), interpolation='bilinear') (qlearnIA_V2.py:166,172,178,184) means.
 * The reference's unpinned keras / tensorflow range admits two: 0 (default) half-pixel centres (TF2 tf.image.resize),
 * 1 the TF1 legacy resize_bilinear(align_corners=False), src = dst / 2.  Weights trained under one give a different
 * heat map under the other.  Applies to the forward, the DQN targets and ofx_dqn_fit; a pinned blob is re-prepared. */
#define OFX_OPT_BILINEAR_LEGACY 3
/* OPT-IN reduced precision, never the default and never the headline number: ofx_policy_forward (the rollout's forward
 * on the live state; ofx_policy_forward_obs, the DQN targets and the fit stay fp32) feeds conv2-4 (streaming trunk) and
 * upconv3-4 - 97 % of the per-ship work - to the 16-bit matrix instructions: operands rounded to nearest even, sums in
 * fp32.  value 1: bf16 operands (8 significant bits); value 2: fp16 operands (11 significant bits; operands beyond
 * 65504 would overflow - activations behind BatchNorm + ReLU and folded weights are far below); value 0: fp32.  The
 * reference's Keras model is fp32 (agents/qlearnIA_V2.py:123-190): with this switch the heat map differs from the fp32
 * path at the 3e-3 (bf16) / 4e-4 (fp16) level and near-ties of its arg-max can resolve differently; bench.py reports the
 * measured error against the float64 graph and the arg-max agreement next to the speed.                              */
#define OFX_OPT_POLICY_BF16 5
/* Diagnostic: ofx_dqn_fit / ofx_dqn_fit_reference in their PLAIN form (value 1): one kernel per layer and pass, every
 * activation, pooled / up-sampled input and gradient of the graph in HBM (61 MB per row of the minibatch).  The default
 * (0) is the lean form: only the pre-activation tensor of every convolution and the trunk's pooled activations are kept (9.4 MB per row; the first layer is never materialised), everything else
 * is recomputed inside fused tiles.  Same function; the results agree up to fp32 summation order (tests/test_train.py).*/
#define OFX_OPT_FIT_PLAIN 6
/* OPT-IN, EXACT: the streaming trunk (OFX_OPT_TRUNK_FUSE) uses the sparsity of its input.  The two input planes are 1-bit
 * maps with ~1 % of the cells set (lib/observation.py:79-95): conv1 of an empty neighbourhood is ONE value per channel,
 * and most of conv2's 16-pixel M-tiles multiply that constant.  value 1: a wave of pixel pairs whose bit windows are all
 * empty writes that value instead of looking its table rows up, and an M-tile whose whole input window holds it (and
 * touches no zero padding) stores the constant the dense matrix sequence produces for it instead of running the
 * sequence.  The results are BIT-IDENTICAL to the dense form (tests/test_gpu_policy.py), also under OFX_OPT_POLICY_BF16
 * (the constant then comes from the 16-bit sequence).  The forwards on stored observations (ofx_policy_forward_obs, the DQN
 * targets) always run this form.  bench.py reports it as a labelled secondary line; the headline stays on the dense trunk.
 * ofx_policy_trunk_stats: since the last call, M-tiles run / all and table passes run / all [4] (resets; synchronises). */
#define OFX_OPT_TRUNK_SPARSE 7
int ofx_policy_trunk_stats(ofx_handle *h, int64_t *counts_host);
int ofx_set_option(ofx_handle *h, int32_t option, int32_t value);
/* Exploration of the bi-head action space (Trainer.get_best_action epsilon branch, agents/qlearnIA_V2.py:199-204,
 * and the collecting phase :393-395): for every selected ship, with probability `epsilon` - or always when
 * `collecting` != 0 - replace (iaction, ipointer) by random_play() (:317-321): iaction = randint(0, 1),
 * ipointer = (randint(0, W-1), randint(0, H-1)).  Draws: Philox4x32-10 keyed by `seed`, counter
 * (global arena, ship, tick, stream 2).  NULL iaction / ipointer = the handle's workspace results.             */
int ofx_policy_explore(ofx_handle *h, double epsilon, uint64_t seed, uint32_t tick, int32_t collecting,
                       const uint8_t *ship_mask, int32_t *iaction, int32_t *ipointer);
/* QlearnIA.play action packing (qlearnIA_V2.py:447-454): exactly one of
 * shoot/thrust set, pointer always set.                                      */
int ofx_policy_actions(ofx_handle *h, const int32_t *iaction, const int32_t *ipointer,
                       const uint8_t *ship_mask, ofx_action *actions);

/* ---- transition capture: the replay memory -----------------------------
 * Trainer.memory = deque(maxlen=memory_size) + Trainer.remember
 * (agents/qlearnIA_V2.py:58,237-238), fed by the bookkeeping of QlearnIA.play
 * (:370-403: nothing once the agent's `done` latch is set; the losing frame is
 * still remembered; previous_obs/action/pointer) and cleared per episode by
 * QlearnIA.reset (:360-368) - ofx_spawn* / ofx_restart* do that here.  Every
 * arena owns one memory; rows are appended in (lock-step, ship index) order,
 * the order of request_actions (lib/battleground.py:146-150).
 * One row = [state, iaction, ipointer, reward, next_state, done]; the two
 * observations are kept as the lock-step numbers of their 1-bit maps in the
 * frame ring plus the ship's own toVector head.  (In the reference every ship
 * of a tick shares ONE mutated Observation object, battleground.py:150, so the
 * head found in its memory is the LAST analysed ship's; the per-ship head is
 * stored here and that aliasing is not reproduced.)                          */
typedef struct ofx_transition {
  int32_t tick_prev, tick_next;   /* lock-steps of state / next_state         */
  int32_t frame_prev, frame_next; /* their slots in the arena's frame ring     */
  int32_t ship;                   /* -1 in padding rows of a gathered batch    */
  int32_t iaction, px, py;        /* previous_action, previous_pointer (x, y) */
  int32_t reward;                 /* obs.reward of next_state                  */
  int32_t done;                   /* obs.done of next_state                    */
  float head_prev[8], head_next[8];
} ofx_transition;                 /* 104 bytes                                 */
/* capacity = memory_size (400 in the reference); frames = length of every
 * arena's ring of stored observation maps (0 = capacity + capacity/4 + 2).  An
 * arena stores a frame only on lock-steps where one of its agents plays, and
 * C rows made of runs of consecutive plays reference C + #runs frames (with K
 * capturing ships per arena about C / K); a row whose `state` frame has been
 * overwritten is no longer eligible for sampling.  A frame costs 2 * W*H/8
 * bytes per arena (40 KB at 400x400).                                         */
int ofx_replay_create(ofx_handle *h, int32_t capacity, int32_t frames);
int ofx_replay_destroy(ofx_handle *h); /* also done by ofx_destroy            */
/* Call once per lock-step after the action choice and BEFORE ofx_step: stores
 * the current observation maps as frame `tick` of every arena with a playing
 * agent and runs the play() bookkeeping for every ship selected by ship_mask
 * (NULL = all) with its chosen (iaction, ipointer) (NULL = the workspace
 * results of ofx_policy_forward / ofx_policy_explore).  `tick` must increase by
 * one per call and not restart at episode boundaries.                         */
int ofx_replay_capture(ofx_handle *h, uint32_t tick, const uint8_t *ship_mask, const int32_t *iaction,
                       const int32_t *ipointer);
/* QlearnIA.play's `if obs.done: ... self.done = True` (agents/qlearnIA_V2.py:376-384) for a caller that drives the
 * learning schedule: counts the selected ships (ship_mask [N][M] uint8, NULL = all) that are destroyed now and were
 * not yet marked in seen[N][M] (device uint8, caller-owned, zeroed by the caller at episode starts), marks them, and
 * returns the count - the number of learning agents that see their death for the first time on this lock-step, i.e.
 * how often the reference would call Trainer.replay here.  One 4-byte read instead of an [N][M] download.
 * Synchronises.                                                                */
int ofx_agents_first_done(ofx_handle *h, const uint8_t *ship_mask, uint8_t *seen, int32_t *count_host);
/* len(memory) per arena [N] and the number of rows ever appended [N]; either
 * may be NULL.  Synchronises.                                                 */
int ofx_replay_count(ofx_handle *h, int32_t *count_host, int64_t *appended_host);
/* list(memory) of one arena, oldest first; rows_host holds `capacity` rows.  */
int ofx_replay_rows_host(ofx_handle *h, int32_t arena, ofx_transition *rows_host, int32_t *n_host);
/* 1-bit maps of a stored lock-step of one arena, W*H/8 bytes each: pixel
 * p = y*W + x -> bit (p & 7) of byte p >> 3 (numpy.unpackbits bitorder=
 * 'little').  OFX_ERR_STATE when the arena's ring does not hold that frame.   */
int ofx_replay_frame_host(ofx_handle *h, int32_t arena, int32_t tick, void *ship_bits_host, void *laser_bits_host);
/* random.sample(memory, min(batch, len(memory))) for every arena
 * (qlearnIA_V2.py:241-243): slot[N][batch] (device) receives row indices
 * (oldest first, -1 pads), n_sampled[N] (device, may be NULL) the number drawn.
 * Uniform without replacement (Floyd), Philox counter (global arena, j, draw,
 * stream 3).                                                                  */
int ofx_replay_sample(ofx_handle *h, uint64_t seed, uint32_t draw, int32_t batch, int32_t *slot, int32_t *n_sampled);
/* Materialise sampled rows (device pointers): rows[N][batch] and, when not
 * NULL, the 1-bit maps of state / next_state [N][batch][2 (ship, laser)]
 * [W*H/32] uint32 in the layout ofx_policy_forward's trunk reads.             */
int ofx_replay_gather(ofx_handle *h, const int32_t *slot, int32_t batch, ofx_transition *rows, void *bits_prev,
                      void *bits_next);
/* The same minibatch WITHOUT padding, the form Trainer.replay works on (its batch is min(batch_size, len(memory)) real
 * transitions, qlearnIA_V2.py:241-243): the sampled entries of all arenas in (arena, j) order, packed; entries `first`
 * .. `first + max_rows - 1` of that sequence land in rows[max_rows] / bits_*[max_rows][2][W*H/32] (device, maps may be
 * NULL) and *n_rows_host receives how many were written.  slot / n_sampled as written by ofx_replay_sample.
 * Synchronises.  This is what ofx_dqn_targets / ofx_dqn_fit take: the fit refuses padding rows.                      */
int ofx_replay_gather_valid(ofx_handle *h, const int32_t *slot, const int32_t *n_sampled, int32_t batch, int32_t first,
                            int32_t max_rows, ofx_transition *rows, void *bits_prev, void *bits_next,
                            int32_t *n_rows_host);

/* ---- forward on stored observations, TD targets -------------------------
 * The predictions Trainer.replay makes on a minibatch (agents/qlearnIA_V2.py:251-268): n_obs observations given as
 * 1-bit map pairs bits[n_obs][2 (ship, laser)][W*H/32] uint32 (the layout ofx_replay_gather writes) + their toVector
 * heads vec8[n_obs][8]; every observation gets its own trunk run.  Outputs (device, any may be NULL): act_values
 * [n_obs][2], iaction [n_obs], ipointer [n_obs][2], ptr_max [n_obs] = np.max of the heat map; with probe [n_obs][2]
 * (x, y) also ptr_probe [n_obs] = the heat-map value at that pointer.  Uses the policy workspace: the (iaction,
 * ipointer) a previous ofx_policy_forward left there are gone afterwards.                                        */
int ofx_policy_forward_obs(ofx_handle *h, const float *weights, int32_t n_obs, const void *bits, const float *vec8,
                           float *act_values, int32_t *iaction, int32_t *ipointer, float *ptr_max,
                           const int32_t *probe, float *ptr_probe);
/* The targets Trainer.replay builds (:269-270) for n gathered transitions (rows, bits_prev, bits_next from
 * ofx_replay_gather): two forwards, then per row
 *   q_sa  = act_values(state)[iaction]           y_act = reward + gamma * max(act_values(next_state)) * (not done)
 *   p_sp  = heat(state) at ipointer              y_ptr = reward + gamma * max(heat(next_state))       * (not done)
 * i.e. the current values and the values written into target[iaction] / ptr_target[ipointer]; their squared
 * differences are the two MSE terms ofx_dqn_fit minimises.  The reference indexes ptr_target[x][y] on a [y][x] map
 * (:280) and fits on next_state's inputs (:282-283); here the pointer addresses the pixel it was chosen as
 * (ofx_dqn_fit_reference reproduces the reference as written).  Padding rows (ship < 0) give zeros.  q_sa and p_sp may
 * both be NULL: the forward on `state` is then skipped (ofx_dqn_fit needs only y_act / y_ptr).                    */
int ofx_dqn_targets(ofx_handle *h, const float *weights, int32_t n, const ofx_transition *rows, const void *bits_prev,
                    const void *bits_next, float gamma, float *q_sa, float *p_sp, float *y_act, float *y_ptr);

/* One fit step of Trainer.replay (model.fit, agents/qlearnIA_V2.py:284; loss 'mse' on both heads, Adam(lr) :190) on n
 * gathered transitions: forward in training mode (BatchNorm on the batch statistics, moving statistics updated with
 * momentum 0.99 like Keras), targets = the current predictions except target[iaction] = y_act[i] and
 * ptr_target at the pointer = y_ptr[i] (from ofx_dqn_targets), backward, Adam (beta 0.9 / 0.999, eps 1e-7, `step`
 * 1-based).  weights / adam_m / adam_v: device float32 [n_floats] in the ofx_policy_layout order, updated in place;
 * grad_out (device, may be NULL) receives the gradient; loss_host[2] = the two mse terms.  Inputs are the
 * transitions' `state` observations and the pointer addresses heat[y][x] (the reference fits on next_state's inputs
 * and indexes [x][y], :280-283: that form is ofx_dqn_fit_reference).  Every row must be a real transition (ship >= 0; use
 * ofx_replay_gather_valid): a padding row would enter the BatchNorm batch statistics and the loss scale, so the call
 * fails with OFX_ERR_INVALID before anything is updated.  The handle keeps a workspace of 9.4 MB per row between calls (given back when a call needs less than a quarter of it)
 * (OFX_OPT_FIT_PLAIN: 61 MB); every reduction has a fixed order, so the same call on the same state gives the same bits.
 * fp32 on the vector ALU (not the hot path): 3.6 ms for 64 rows, 42 ms for 4096 with the target forward (ofx_dqn_fit_reference: 4.7 / 65 ms); synchronises. */
int ofx_dqn_fit(ofx_handle *h, float *weights, float *adam_m, float *adam_v, int32_t step, float lr, int32_t n,
                const ofx_transition *rows, const void *bits_prev, const float *y_act, const float *y_ptr,
                float *grad_out, float *loss_host);
/* The same step with Trainer.replay's quirks reproduced as written (agents/qlearnIA_V2.py:251-285), for a user who
 * wants the reference's training dynamics rather than the textbook DQN step above:
 *   - targets are whole predictions of `state` ([target, ptr_target] = predict(state), inference-mode BatchNorm) with
 *     target[iaction] and ptr_target[ipointer] replaced (:279-280);
 *   - ipointer = (x, y) indexes the (400, 400, 1) prediction as [x][y] - row x, column y, the transpose of the pixel
 *     get_best_action named (:218-220 vs :280);
 *   - the fit's inputs are NEXT_state's maps and head (img_input is re-bound at :273; :282-283), so every output
 *     element carries an error (training-mode forward on next_state against targets built from state).
 * Computes the targets itself (gamma = Trainer.gamma, 0.9 at :60): rows / bits_prev / bits_next from
 * ofx_replay_gather_valid.  Everything else (loss scale, Adam, moving statistics, padding refusal, outputs) as
 * ofx_dqn_fit.  Parity with Keras is unpinned for both forms.                                                     */
int ofx_dqn_fit_reference(ofx_handle *h, float *weights, float *adam_m, float *adam_v, int32_t step, float lr, int32_t n,
                          const ofx_transition *rows, const void *bits_prev, const void *bits_next, float gamma,
                          float *grad_out, float *loss_host);

/* ---- timing helpers (HIP events on the handle's stream) ---------------- */
int ofx_timer_start(ofx_handle *h);
int ofx_timer_stop(ofx_handle *h, float *ms_host); /* synchronises */
/* Non-blocking per-kernel timing for bench.py: record numbered events on the
 * handle's stream (created on first use, idx in [0, 65536)), read the elapsed
 * time between two of them later (synchronises on the second).              */
int ofx_event_record(ofx_handle *h, int32_t idx);
/* When event_base >= 0 every following ofx_policy_forward records events event_base / event_base+1 around its
 * dominant kernel (k_head_stream: upconv2-4 + arg-max) and then advances event_base by 2; it switches itself off
 * when the event ring is full; -1 switches it off.                                                               */
int ofx_policy_profile(ofx_handle *h, int32_t event_base);
int ofx_event_elapsed(ofx_handle *h, int32_t idx_from, int32_t idx_to, float *ms_host);

#ifdef __cplusplus
}
#endif
#endif /* OFX_H */
