// ofx_api.hip - C-ABI entry points of libofx.so (see include/ofx.h) and the
// small per-(arena, ship) kernels: spawn, restart, scripted bots, obs head.
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include "ofx_internal.h"

// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

void ofx_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *ofx_last_error(void) { return g_err; }
extern "C" int ofx_version(void) { return OFX_VERSION; }

extern "C" int ofx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" void ofx_default_config(ofx_config *c) {
  memset(c, 0, sizeof(*c));
  c->n_arenas = 1;
  c->n_ships = 8;
  c->laser_cap = 512;
  c->width = 400;           // lib/observation.py:10-11
  c->height = 400;
  c->ship_radius = 8;       // lib/ship.py:43
  c->laser_radius = 2;      // lib/laser.py:23
  c->ship_speed = 8;        // lib/ship.py:24
  c->laser_speed = 10;      // lib/ship.py:86 x lib/laser.py:13
  c->reward_death = 0;      // agents/qlearnIA_V2.py:39-44
  c->reward_kill = 0;
  c->reward_aim = 2;
  c->reward_trajectory = 1;
  c->episode_ticks = 200;   // lib/ofighters.py:59
  c->device = 0;
  c->arena_base = 0;
}

static int require_device(void) {
  if (ofx_device_count() <= 0) {
    ofx_set_error("no HIP device visible: libofx has no CPU fallback");
    return OFX_ERR_NO_DEVICE;
  }
  return OFX_OK;
}

// ------------------------------------------------------------ device memory
extern "C" int ofx_malloc(void **dev_ptr, size_t bytes) {
  int rc = require_device();
  if (rc) return rc;
  OFX_HIP(hipMalloc(dev_ptr, bytes ? bytes : 1));
  return OFX_OK;
}
extern "C" int ofx_free(void *dev_ptr) {
  if (dev_ptr) OFX_HIP(hipFree(dev_ptr));
  return OFX_OK;
}
extern "C" int ofx_memcpy_h2d(void *dst, const void *src, size_t bytes) {
  OFX_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  // a copy from pageable memory may still be in flight when hipMemcpy returns, and handles run on
  // non-blocking streams: make the bytes visible to every stream before returning (host-convenience path only)
  OFX_HIP(hipDeviceSynchronize());
  return OFX_OK;
}
extern "C" int ofx_memcpy_d2h(void *dst, const void *src, size_t bytes) {
  OFX_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return OFX_OK;
}

// ---------------------------------------------------------------- lifetime
// largest double t with rn(sqrt(t)) <= r  (x86 sqrt is correctly rounded)
static double sqrt_le_threshold(double r) {
  double t = r * r;
  while (sqrt(nextafter(t, INFINITY)) <= r) t = nextafter(t, INFINITY);
  while (sqrt(t) > r) t = nextafter(t, -INFINITY);
  return t;
}

template <typename T>
static int dev_alloc_zero(T **p, size_t count) {
  OFX_HIP(hipMalloc((void **)p, count * sizeof(T) ? count * sizeof(T) : 1));
  // hipMemset on the null stream returns before the fill has run and is NOT ordered with the handle's
  // non-blocking stream: ofx_create synchronises the device once after all allocations (see there)
  OFX_HIP(hipMemset(*p, 0, count * sizeof(T)));
  return OFX_OK;
}

extern "C" int ofx_create(const ofx_config *cfg, ofx_handle **out) {
  if (!cfg || !out) { ofx_set_error("ofx_create: null argument"); return OFX_ERR_INVALID; }
  *out = nullptr;
  if (cfg->n_arenas < 1 || cfg->n_ships < 1 || cfg->n_ships > 64) {
    ofx_set_error("ofx_create: need n_arenas >= 1 and 1 <= n_ships <= 64 (got %d, %d)", cfg->n_arenas, cfg->n_ships);
    return OFX_ERR_INVALID;
  }
  if (cfg->laser_cap < 64 || cfg->laser_cap % 64 || cfg->laser_cap > 32704) {
    ofx_set_error("ofx_create: laser_cap must be a multiple of 64 in [64, 32704] (got %d)", cfg->laser_cap);
    return OFX_ERR_INVALID;
  }
  if (cfg->width < 1 || cfg->height < 1 || ((long long)cfg->width * cfg->height) % 32 ||
      (long long)cfg->width * cfg->height > 524288) {
    ofx_set_error("ofx_create: width*height must be a positive multiple of 32 and <= 524288 (got %d x %d)",
                  cfg->width, cfg->height);
    return OFX_ERR_INVALID;
  }
  if (cfg->ship_radius < 1 || cfg->laser_radius < 1 || cfg->ship_speed < 0 || cfg->laser_speed < 0) {
    ofx_set_error("ofx_create: radii must be >= 1 and speeds >= 0");
    return OFX_ERR_INVALID;
  }
  int rc = require_device();
  if (rc) return rc;
  if (cfg->device < 0 || cfg->device >= ofx_device_count()) {
    ofx_set_error("ofx_create: device %d out of range", cfg->device);
    return OFX_ERR_INVALID;
  }
  OFX_HIP(hipSetDevice(cfg->device));
  ofx_handle *h = new (std::nothrow) ofx_handle();
  if (!h) { ofx_set_error("ofx_create: out of host memory"); return OFX_ERR_INVALID; }
  memset(h, 0, sizeof(*h));
  h->cfg = *cfg;
  h->hit_thresh = sqrt_le_threshold((double)(cfg->laser_radius + cfg->ship_radius));
  const size_t NM = (size_t)cfg->n_arenas * cfg->n_ships, NL = (size_t)cfg->n_arenas * cfg->laser_cap,
               N = (size_t)cfg->n_arenas;
  ofx_state &s = h->st;
#define A(field, count) if ((rc = dev_alloc_zero(&s.field, count))) { ofx_destroy(h); return rc; }
  A(ship_x, NM) A(ship_y, NM) A(ship_px, NM) A(ship_py, NM) A(hull, NM)
  A(reward, NM) A(score, NM) A(obs_reward, NM) A(last_score, NM)
  A(alive, NM) A(killer, NM)
  A(time, N) A(n_lasers, N)
  A(laser_x, NL) A(laser_y, NL) A(laser_dx, NL) A(laser_dy, NL)
  A(laser_owner, NL) A(laser_dead, NL)
  A(overflow, 1) A(episode_sums, (size_t)cfg->n_ships + 1)
#undef A
  if ((rc = dev_alloc_zero(&h->bot_behaviours, (size_t)cfg->n_ships))) { ofx_destroy(h); return rc; }
  h->n_cus = 256;  // MI355X; read from the device below
  { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && v > 0) h->n_cus = v; }
  hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { ofx_set_error("hipStreamCreate: %s", hipGetErrorString(e)); ofx_destroy(h); return OFX_ERR_HIP; }
  h->own_stream = true;
  h->prof_base = -1;
  // the zero fills above ran on the null stream; the handle's stream is non-blocking, so without this a kernel
  // enqueued right after ofx_create (k_spawn) could be overtaken by a late memset (seen with two handles in a row)
  e = hipDeviceSynchronize();
  if (e != hipSuccess) { ofx_set_error("hipDeviceSynchronize: %s", hipGetErrorString(e)); ofx_destroy(h); return OFX_ERR_HIP; }
  *out = h;
  return OFX_OK;
}

extern "C" int ofx_destroy(ofx_handle *h) {
  if (!h) return OFX_OK;
  (void)hipSetDevice(h->cfg.device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  ofx_replay_free(h);
  ofx_state &s = h->st;
  void *ptrs[] = {s.ship_x, s.ship_y, s.ship_px, s.ship_py, s.hull, s.reward, s.score, s.obs_reward, s.last_score,
                  s.alive, s.killer, s.time, s.n_lasers, s.laser_x, s.laser_y, s.laser_dx, s.laser_dy,
                  s.laser_owner, s.laser_dead, s.overflow, s.episode_sums, h->bot_behaviours, h->scratch, h->aux, h->prep, h->prep_tmp, h->counter, h->fitws, h->fitws2, h->trunk_stat};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  for (int t = 0; t < 5; t++) for (int w = 0; w < 2; w++) if (h->maps[t][w]) (void)hipFree(h->maps[t][w]);
  if (h->events) { (void)hipEventDestroy(h->ev0); (void)hipEventDestroy(h->ev1); }
  if (h->ring) {
    for (int i = 0; i < h->ring_n; i++) if (h->ring[i]) (void)hipEventDestroy(h->ring[i]);
    free(h->ring);
  }
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return OFX_OK;
}

extern "C" int ofx_sync(ofx_handle *h) {
  if (!h) { ofx_set_error("null handle"); return OFX_ERR_INVALID; }
  OFX_HIP(hipStreamSynchronize(h->stream));
  return OFX_OK;
}

extern "C" void *ofx_stream(ofx_handle *h) { return h ? (void *)h->stream : nullptr; }

extern "C" int ofx_set_stream(ofx_handle *h, void *stream) {
  if (!h) { ofx_set_error("null handle"); return OFX_ERR_INVALID; }
  OFX_HIP(hipStreamSynchronize(h->stream));
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  h->stream = (hipStream_t)stream;
  h->own_stream = false;
  return OFX_OK;
}

int ofx_ensure_scratch(ofx_handle *h, size_t bytes) {
  if (h->scratch_bytes >= bytes) return OFX_OK;
  OFX_HIP(hipStreamSynchronize(h->stream));
  if (h->scratch) (void)hipFree(h->scratch);
  h->scratch = nullptr;
  h->scratch_bytes = 0;
  OFX_HIP(hipMalloc(&h->scratch, bytes));
  h->scratch_bytes = bytes;
  return OFX_OK;
}

// ----------------------------------------------------------- spawn / restart
struct ResetParams {
  int N, M, W, H, arena_base;
  ofx_state st;
  const int32_t *draws;      // [N][M][2] or null -> counter RNG
  const uint8_t *mask;       // [N] or null
  uint32_t k0, k1, episode;
};

// Battleground.__init__ -> Ship.__init__ (battleground.py:79-81, ship.py:35-58)
__global__ void k_spawn(ResetParams p) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= p.N * p.M) return;
  const int a = t / p.M, i = t - a * p.M;
  int dx, dy;
  if (p.draws) {
    dx = p.draws[2 * t];
    dy = p.draws[2 * t + 1];
  } else {
    uint32_t r[4];
    ofx_philox4x32_10((uint32_t)(p.arena_base + a), (uint32_t)i, p.episode, OFX_STREAM_RESET, p.k0, p.k1, r);
    dx = ofx_draw_int(r[0], p.W);
    dy = ofx_draw_int(r[1], p.H);
  }
  p.st.ship_x[t] = dx; p.st.ship_y[t] = dy;
  p.st.ship_px[t] = dx; p.st.ship_py[t] = dy;
  p.st.hull[t] = 1;
  p.st.reward[t] = 0; p.st.score[t] = 0; p.st.obs_reward[t] = 0; p.st.last_score[t] = 0;
  p.st.alive[t] = 1;
  p.st.killer[t] = -1;
  if (i == 0) { p.st.time[a] = 0; p.st.n_lasers[a] = 0; }
}

// Battleground.restart -> Ship.reset -> Agent.reset
// (battleground.py:108-117, ship.py:92-106, agent.py:59-64)
__global__ void k_restart(ResetParams p) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= p.N * p.M) return;
  const int a = t / p.M, i = t - a * p.M;
  if (p.mask && !p.mask[a]) return;
  int dx, dy;
  if (p.draws) {
    dx = p.draws[2 * t];
    dy = p.draws[2 * t + 1];
  } else {
    uint32_t r[4];
    ofx_philox4x32_10((uint32_t)(p.arena_base + a), (uint32_t)i, p.episode, OFX_STREAM_RESET, p.k0, p.k1, r);
    dx = ofx_draw_int(r[0], p.W);
    dy = ofx_draw_int(r[1], p.H);
  }
  const int sc = p.st.score[t];
  p.st.last_score[t] = sc;                       // scores.append(score)
  atomicAdd((unsigned long long *)&p.st.episode_sums[i], (unsigned long long)(long long)sc);
  p.st.score[t] = 0;                             // reward is NOT cleared (agent.py:59-64)
  const int ox = p.st.ship_x[t], oy = p.st.ship_y[t];
  p.st.ship_px[t] = ox;                          // pointing = Point(old x, old y)  ship.py:99
  p.st.ship_py[t] = oy;
  p.st.ship_x[t] = dx ? dx : ox;                 // `x or self.body.x`            ship.py:100-101
  p.st.ship_y[t] = dy ? dy : oy;
  p.st.alive[t] = 1;                             // state = "flying"; hull NOT restored
  p.st.killer[t] = -1;
  if (i == 0) {
    p.st.time[a] = 0;
    p.st.n_lasers[a] = 0;                        // self.lasers = []
    atomicAdd((unsigned long long *)&p.st.episode_sums[p.M], 1ull);
  }
}

static ResetParams reset_params(ofx_handle *h, const int32_t *draws, const uint8_t *mask, uint64_t seed,
                                uint32_t episode) {
  ResetParams p;
  p.N = h->cfg.n_arenas; p.M = h->cfg.n_ships; p.W = h->cfg.width; p.H = h->cfg.height;
  p.arena_base = h->cfg.arena_base;
  p.st = h->st; p.draws = draws; p.mask = mask;
  p.k0 = (uint32_t)seed; p.k1 = (uint32_t)(seed >> 32); p.episode = episode;
  return p;
}

static int do_spawn(ofx_handle *h, const int32_t *draws, uint64_t seed) {
  if (!h) { ofx_set_error("null handle"); return OFX_ERR_INVALID; }
  OFX_HIP(hipSetDevice(h->cfg.device));
  ResetParams p = reset_params(h, draws, nullptr, seed, 0);
  const int T = p.N * p.M;
  OFX_HIP(hipMemsetAsync(h->st.episode_sums, 0, sizeof(long long) * (p.M + 1), h->stream));
  hipLaunchKernelGGL(k_spawn, dim3((T + 255) / 256), dim3(256), 0, h->stream, p);
  OFX_HIP(hipGetLastError());
  h->spawned = true;
  return ofx_replay_episode_reset(h, nullptr);  // fresh agents: previous_* = None
}

static int do_restart(ofx_handle *h, const int32_t *draws, const uint8_t *mask, uint64_t seed, uint32_t episode) {
  if (!h) { ofx_set_error("null handle"); return OFX_ERR_INVALID; }
  if (!h->spawned) { ofx_set_error("ofx_restart before ofx_spawn"); return OFX_ERR_STATE; }
  OFX_HIP(hipSetDevice(h->cfg.device));
  ResetParams p = reset_params(h, draws, mask, seed, episode);
  const int T = p.N * p.M;
  OFX_HIP(hipMemsetAsync(h->st.episode_sums, 0, sizeof(long long) * (p.M + 1), h->stream));
  hipLaunchKernelGGL(k_restart, dim3((T + 255) / 256), dim3(256), 0, h->stream, p);
  OFX_HIP(hipGetLastError());
  return ofx_replay_episode_reset(h, mask);  // QlearnIA.reset (qlearnIA_V2.py:360-368)
}

extern "C" int ofx_spawn(ofx_handle *h, const int32_t *draws) {
  if (!draws) { ofx_set_error("ofx_spawn: null draws"); return OFX_ERR_INVALID; }
  return do_spawn(h, draws, 0);
}
extern "C" int ofx_spawn_random(ofx_handle *h, uint64_t seed) { return do_spawn(h, nullptr, seed); }
extern "C" int ofx_restart(ofx_handle *h, const int32_t *draws, const uint8_t *arena_mask) {
  if (!draws) { ofx_set_error("ofx_restart: null draws"); return OFX_ERR_INVALID; }
  return do_restart(h, draws, arena_mask, 0, 0);
}
extern "C" int ofx_restart_random(ofx_handle *h, uint64_t seed, uint32_t episode) {
  return do_restart(h, nullptr, nullptr, seed, episode);
}

// --------------------------------------------------------------------- tick
extern "C" int ofx_step(ofx_handle *h, const ofx_action *actions) {
  if (!h || !actions) { ofx_set_error("ofx_step: null argument"); return OFX_ERR_INVALID; }
  if (!h->spawned) { ofx_set_error("ofx_step before ofx_spawn"); return OFX_ERR_STATE; }
  OFX_HIP(hipSetDevice(h->cfg.device));
  return ofx_launch_step(h, actions);
}

// ------------------------------------------------------------- scripted bots
struct BotParams {
  int N, M, W, H, arena_base;
  ofx_state st;
  const int32_t *beh;
  uint32_t k0, k1, tick;
  ofx_action *out;
};

// agents/agent.py:99-155 action laws on the counter RNG; a dead ship's agent is
// still stepped but its action is None (ship.py:260-262)
__global__ void k_bots(BotParams p) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= p.N * p.M) return;
  const int a = t / p.M, i = t - a * p.M;
  uint32_t r[4];
  ofx_philox4x32_10((uint32_t)(p.arena_base + a), (uint32_t)i, p.tick, OFX_STREAM_BOT, p.k0, p.k1, r);
  bool shoot = false, thrust = false, repoint = false;
  const double u0 = (double)r[0] * (1.0 / 4294967296.0), u1 = (double)r[1] * (1.0 / 4294967296.0);
  switch (p.beh[i]) {
    case OFX_BOT_RANDOM: {
      const uint32_t k = (uint32_t)(((uint64_t)r[0] * 3u) >> 32);
      shoot = k == 0; thrust = k == 1; repoint = k == 2;
    } break;
    case OFX_BOT_TURRET: shoot = u0 < 0.8; repoint = u1 < 0.3; break;
    case OFX_BOT_RUNNER: thrust = u0 < 0.9; repoint = u1 < 0.1; break;
    case OFX_BOT_THRUST: thrust = true; break;
    case OFX_BOT_SHOOT: shoot = true; break;
    default: break;
  }
  const bool alive = p.st.alive[t] != 0;
  ofx_action act;
  act.px = p.st.ship_px[t];
  act.py = p.st.ship_py[t];
  act.shoot = 0; act.thrust = 0; act.valid = alive ? 1 : 0; act._pad = 0;
  if (alive) {
    act.shoot = shoot; act.thrust = thrust;
    if (repoint) { act.px = ofx_draw_int(r[2], p.W); act.py = ofx_draw_int(r[3], p.H); }
  }
  p.out[t] = act;
}

extern "C" int ofx_bot_actions(ofx_handle *h, const int32_t *behaviours_host, uint64_t seed, uint32_t tick,
                               ofx_action *actions) {
  if (!h || !behaviours_host || !actions) { ofx_set_error("ofx_bot_actions: null argument"); return OFX_ERR_INVALID; }
  if (!h->spawned) { ofx_set_error("ofx_bot_actions before ofx_spawn"); return OFX_ERR_STATE; }
  for (int i = 0; i < h->cfg.n_ships; i++)
    if (behaviours_host[i] < OFX_BOT_IDLE || behaviours_host[i] > OFX_BOT_SHOOT) {
      // agents/agent.py:51
      ofx_set_error("You must give a bot in parameter or select an existing behavior.");
      return OFX_ERR_INVALID;
    }
  OFX_HIP(hipSetDevice(h->cfg.device));
  OFX_HIP(hipMemcpyAsync(h->bot_behaviours, behaviours_host, sizeof(int32_t) * h->cfg.n_ships,
                         hipMemcpyHostToDevice, h->stream));
  BotParams p;
  p.N = h->cfg.n_arenas; p.M = h->cfg.n_ships; p.W = h->cfg.width; p.H = h->cfg.height;
  p.arena_base = h->cfg.arena_base; p.st = h->st; p.beh = h->bot_behaviours;
  p.k0 = (uint32_t)seed; p.k1 = (uint32_t)(seed >> 32); p.tick = tick; p.out = actions;
  const int T = p.N * p.M;
  hipLaunchKernelGGL(k_bots, dim3((T + 255) / 256), dim3(256), 0, h->stream, p);
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}

// ------------------------------------------------ headless loop: K lock-steps per host call
// Battleground.run (lib/battleground.py:169-173: `while True: self.frame()`) for the scripted bots: request_actions,
// generate_frame and - when observing - Observation(battleground) for n_ticks lock-steps enqueued by ONE call.
int ofx_launch_step_bots(ofx_handle *h, uint64_t seed, uint32_t tick0, int n_ticks);  // ofx_step.hip

extern "C" int ofx_rollout(ofx_handle *h, const int32_t *behaviours_host, uint64_t seed, uint32_t tick0, int32_t n_ticks,
                           int32_t observe_map_type) {
  if (!h || !behaviours_host) { ofx_set_error("ofx_rollout: null argument"); return OFX_ERR_INVALID; }
  if (!h->spawned) { ofx_set_error("ofx_rollout before ofx_spawn"); return OFX_ERR_STATE; }
  if (n_ticks < 0 || observe_map_type < -1 || observe_map_type > OFX_MAP_BITS) {
    ofx_set_error("ofx_rollout: n_ticks must be >= 0 and observe_map_type -1 (none) or an OFX_MAP_* type");
    return OFX_ERR_INVALID;
  }
  for (int i = 0; i < h->cfg.n_ships; i++)
    if (behaviours_host[i] < OFX_BOT_IDLE || behaviours_host[i] > OFX_BOT_SHOOT) {
      ofx_set_error("You must give a bot in parameter or select an existing behavior.");  // agents/agent.py:51
      return OFX_ERR_INVALID;
    }
  if (n_ticks == 0) return OFX_OK;
  OFX_HIP(hipSetDevice(h->cfg.device));
  OFX_HIP(hipMemcpyAsync(h->bot_behaviours, behaviours_host, sizeof(int32_t) * h->cfg.n_ships, hipMemcpyHostToDevice,
                         h->stream));
  int rc;
  const int pb = h->prof_base;  // ofx_policy_profile: one event pair around this call's dominant kernel
  const bool prof = pb >= 0 && pb + 1 < OFX_RING_MAX;
  if (observe_map_type < 0) {
    // no observer between the lock-steps: all of them inside ONE launch (an arena belongs to one wavefront)
    if (prof && (rc = ofx_event_record(h, pb))) return rc;
    if ((rc = ofx_launch_step_bots(h, seed, tick0, n_ticks))) return rc;
    if (prof && (rc = ofx_event_record(h, pb + 1))) return rc;
  } else {
    for (int t = 0; t < n_ticks; t++) {
      if ((rc = ofx_launch_step_bots(h, seed, tick0 + (uint32_t)t, 1))) return rc;
      const bool last = t + 1 == n_ticks;  // the rasteriser of the last lock-step stands for the launch pair
      if (prof && last && (rc = ofx_event_record(h, pb))) return rc;
      if ((rc = ofx_launch_raster(h, observe_map_type, nullptr, nullptr))) return rc;
      if (prof && last && (rc = ofx_event_record(h, pb + 1))) return rc;
    }
  }
  if (pb >= 0) h->prof_base = pb + 3 < OFX_RING_MAX ? pb + 2 : -1;
  return OFX_OK;
}

// ------------------------------------------------------------------ obs head
// Observation.analyse_ship + toVector head (observation.py:101-123)
__global__ void k_obs_head(int N, int M, int W, int H, ofx_state st, double *head, uint8_t *done) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * M) return;
  double *v = head + (size_t)t * 8;
  v[0] = (double)st.reward[t];
  v[1] = 1.0;  // can_shoot is constant 1 (ship.py:58)
  v[2] = (double)st.ship_px[t];
  v[3] = (double)st.ship_py[t];
  v[4] = (double)W;
  v[5] = (double)H;
  v[6] = (double)st.ship_x[t];
  v[7] = (double)st.ship_y[t];
  if (done) done[t] = st.alive[t] ? 0 : 1;
}

extern "C" int ofx_observe_head(ofx_handle *h, double *head, uint8_t *done) {
  if (!h || !head) { ofx_set_error("ofx_observe_head: null argument"); return OFX_ERR_INVALID; }
  if (!h->spawned) { ofx_set_error("You must execute analyse_battleground first."); return OFX_ERR_STATE; }
  OFX_HIP(hipSetDevice(h->cfg.device));
  const int T = h->cfg.n_arenas * h->cfg.n_ships;
  hipLaunchKernelGGL(k_obs_head, dim3((T + 255) / 256), dim3(256), 0, h->stream, h->cfg.n_arenas, h->cfg.n_ships,
                     h->cfg.width, h->cfg.height, h->st, head, done);
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}

extern "C" int ofx_rasterise(ofx_handle *h, int map_type, void *ship_map, void *laser_map) {
  if (!h) { ofx_set_error("null handle"); return OFX_ERR_INVALID; }
  if (!h->spawned) { ofx_set_error("ofx_rasterise before ofx_spawn"); return OFX_ERR_STATE; }
  if (map_type < OFX_MAP_U8 || map_type > OFX_MAP_BITS_LSB) {
    ofx_set_error("ofx_rasterise: unknown map type %d", map_type);
    return OFX_ERR_INVALID;
  }
  if ((ship_map == nullptr) != (laser_map == nullptr)) {
    ofx_set_error("ofx_rasterise: pass both map pointers or neither");
    return OFX_ERR_INVALID;
  }
  OFX_HIP(hipSetDevice(h->cfg.device));
  return ofx_launch_raster(h, map_type, ship_map, laser_map);
}

extern "C" size_t ofx_map_bytes(const ofx_handle *h, int map_type) {
  if (!h) return 0;
  const size_t cells = (size_t)h->cfg.width * h->cfg.height;
  switch (map_type) {
    case OFX_MAP_U8: return cells;
    case OFX_MAP_F32: return cells * 4;
    case OFX_MAP_F64: return cells * 8;
    case OFX_MAP_BITS:
    case OFX_MAP_BITS_LSB: return cells / 8;
    default: return 0;
  }
}

extern "C" void *ofx_map_ptr(ofx_handle *h, int map_type, int which) {
  if (!h || map_type < 0 || map_type > OFX_MAP_BITS_LSB || which < 0 || which > 1) return nullptr;
  return h->maps[map_type][which];
}

// -------------------------------------------------------------- state access
static void *field_ptr(const ofx_handle *h, int f, size_t *bytes) {
  const ofx_state &s = h->st;
  const size_t NM = (size_t)h->cfg.n_arenas * h->cfg.n_ships, NL = (size_t)h->cfg.n_arenas * h->cfg.laser_cap,
               N = (size_t)h->cfg.n_arenas;
  switch (f) {
    case OFX_F_SHIP_X: *bytes = NM * 4; return s.ship_x;
    case OFX_F_SHIP_Y: *bytes = NM * 4; return s.ship_y;
    case OFX_F_SHIP_PX: *bytes = NM * 4; return s.ship_px;
    case OFX_F_SHIP_PY: *bytes = NM * 4; return s.ship_py;
    case OFX_F_SHIP_ALIVE: *bytes = NM; return s.alive;
    case OFX_F_REWARD: *bytes = NM * 4; return s.reward;
    case OFX_F_SCORE: *bytes = NM * 4; return s.score;
    case OFX_F_N_LASERS: *bytes = N * 4; return s.n_lasers;
    case OFX_F_LASER_X: *bytes = NL * 8; return s.laser_x;
    case OFX_F_LASER_Y: *bytes = NL * 8; return s.laser_y;
    case OFX_F_LASER_OWNER: *bytes = NL; return s.laser_owner;
    case OFX_F_LASER_DEAD: *bytes = NL; return s.laser_dead;
    case OFX_F_KILLER: *bytes = NM * 2; return s.killer;
    case OFX_F_TIME: *bytes = N * 4; return s.time;
    case OFX_F_LAST_SCORES: *bytes = NM * 4; return s.last_score;
    case OFX_F_HULL: *bytes = NM * 4; return s.hull;
    case OFX_F_LASER_DX: *bytes = NL * 8; return s.laser_dx;
    case OFX_F_LASER_DY: *bytes = NL * 8; return s.laser_dy;
    case OFX_F_OBS_REWARD: *bytes = NM * 4; return s.obs_reward;
    default: *bytes = 0; return nullptr;
  }
}

extern "C" size_t ofx_field_bytes(const ofx_handle *h, int field) {
  size_t b = 0;
  if (h) field_ptr(h, field, &b);
  return b;
}

extern "C" void *ofx_device_ptr(ofx_handle *h, int field) {
  size_t b;
  return h ? field_ptr(h, field, &b) : nullptr;
}

// ---- zero-copy views (ofx_field_desc / ofx_map_desc)
static void desc_fill(ofx_tensor_desc *d, void *data, int dtype, int device, int ndim, int64_t s0, int64_t s1, int64_t s2) {
  static const int isz[] = {1, 2, 4, 8, 4, 8};
  d->data = data;
  d->dtype = dtype;
  d->itemsize = isz[dtype];
  d->ndim = ndim;
  d->device = device;
  const int64_t sh[4] = {s0, s1, s2, 1};
  int64_t st = 1;
  for (int i = 3; i >= 0; i--) {
    d->shape[i] = i < ndim ? sh[i] : 1;
    d->stride[i] = i < ndim ? st : 0;
    if (i < ndim) st *= sh[i];
  }
}

extern "C" int ofx_field_desc(ofx_handle *h, int field, ofx_tensor_desc *out) {
  if (!h || !out) { ofx_set_error("ofx_field_desc: null argument"); return OFX_ERR_INVALID; }
  size_t b;
  void *ptr = field_ptr(h, field, &b);
  if (!ptr) { ofx_set_error("ofx_field_desc: unknown field %d", field); return OFX_ERR_INVALID; }
  const int64_t N = h->cfg.n_arenas, M = h->cfg.n_ships, L = h->cfg.laser_cap;
  switch (field) {
    case OFX_F_SHIP_ALIVE: desc_fill(out, ptr, OFX_DT_U8, h->cfg.device, 2, N, M, 1); break;
    case OFX_F_KILLER: desc_fill(out, ptr, OFX_DT_I16, h->cfg.device, 2, N, M, 1); break;
    case OFX_F_N_LASERS:
    case OFX_F_TIME: desc_fill(out, ptr, OFX_DT_I32, h->cfg.device, 1, N, 1, 1); break;
    case OFX_F_LASER_X:
    case OFX_F_LASER_Y:
    case OFX_F_LASER_DX:
    case OFX_F_LASER_DY: desc_fill(out, ptr, OFX_DT_F64, h->cfg.device, 2, N, L, 1); break;
    case OFX_F_LASER_OWNER:
    case OFX_F_LASER_DEAD: desc_fill(out, ptr, OFX_DT_U8, h->cfg.device, 2, N, L, 1); break;
    default: desc_fill(out, ptr, OFX_DT_I32, h->cfg.device, 2, N, M, 1); break;   // the [N][M] int32 ship / agent fields
  }
  if ((size_t)(out->shape[0] * (out->ndim > 1 ? out->shape[1] : 1)) * (size_t)out->itemsize != b) {
    ofx_set_error("ofx_field_desc: field %d: description and size disagree", field);
    return OFX_ERR_STATE;
  }
  return OFX_OK;
}

extern "C" int ofx_map_desc(ofx_handle *h, int map_type, int which, ofx_tensor_desc *out) {
  if (!h || !out) { ofx_set_error("ofx_map_desc: null argument"); return OFX_ERR_INVALID; }
  if (map_type < OFX_MAP_U8 || map_type > OFX_MAP_BITS || which < 0 || which > 1) {
    ofx_set_error("ofx_map_desc: map type %d / map %d", map_type, which);
    return OFX_ERR_INVALID;
  }
  OFX_HIP(hipSetDevice(h->cfg.device));
  const size_t per_map = ofx_map_bytes(h, map_type);
  for (int w = 0; w < 2; w++)
    if (!h->maps[map_type][w]) {   // the buffer ofx_rasterise(h, map_type, NULL, NULL) fills: the same lazy allocation
      OFX_HIP(hipMalloc(&h->maps[map_type][w], per_map * (size_t)h->cfg.n_arenas));
      OFX_HIP(hipMemsetAsync(h->maps[map_type][w], 0, per_map * (size_t)h->cfg.n_arenas, h->stream));
    }
  const int64_t N = h->cfg.n_arenas, W = h->cfg.width, H = h->cfg.height;
  void *ptr = h->maps[map_type][which];
  switch (map_type) {
    case OFX_MAP_U8: desc_fill(out, ptr, OFX_DT_U8, h->cfg.device, 3, N, W, H); break;
    case OFX_MAP_F32: desc_fill(out, ptr, OFX_DT_F32, h->cfg.device, 3, N, W, H); break;
    case OFX_MAP_F64: desc_fill(out, ptr, OFX_DT_F64, h->cfg.device, 3, N, W, H); break;
    default: desc_fill(out, ptr, OFX_DT_U8, h->cfg.device, 2, N, (int64_t)per_map, 1); break;
  }
  return OFX_OK;
}

extern "C" int ofx_get_host(ofx_handle *h, int field, void *dst_host, size_t bytes) {
  if (!h || !dst_host) { ofx_set_error("ofx_get_host: null argument"); return OFX_ERR_INVALID; }
  size_t b;
  void *src = field_ptr(h, field, &b);
  if (!src) { ofx_set_error("ofx_get_host: unknown field %d", field); return OFX_ERR_INVALID; }
  if (bytes != b) { ofx_set_error("ofx_get_host: field %d is %zu bytes, caller passed %zu", field, b, bytes); return OFX_ERR_INVALID; }
  OFX_HIP(hipSetDevice(h->cfg.device));
  OFX_HIP(hipMemcpyAsync(dst_host, src, b, hipMemcpyDeviceToHost, h->stream));
  OFX_HIP(hipStreamSynchronize(h->stream));
  return OFX_OK;
}

extern "C" int ofx_overflow_count(ofx_handle *h, int64_t *count_host) {
  if (!h || !count_host) { ofx_set_error("ofx_overflow_count: null argument"); return OFX_ERR_INVALID; }
  OFX_HIP(hipSetDevice(h->cfg.device));
  unsigned long long v = 0;
  OFX_HIP(hipMemcpyAsync(&v, h->st.overflow, sizeof(v), hipMemcpyDeviceToHost, h->stream));
  OFX_HIP(hipMemsetAsync(h->st.overflow, 0, sizeof(v), h->stream));
  OFX_HIP(hipStreamSynchronize(h->stream));
  *count_host = (int64_t)v;
  return OFX_OK;
}

extern "C" int ofx_episode_scores(ofx_handle *h, int64_t *sums) {
  if (!h || !sums) { ofx_set_error("ofx_episode_scores: null argument"); return OFX_ERR_INVALID; }
  OFX_HIP(hipSetDevice(h->cfg.device));
  OFX_HIP(hipMemcpyAsync(sums, h->st.episode_sums, sizeof(int64_t) * (h->cfg.n_ships + 1), hipMemcpyDeviceToDevice,
                         h->stream));
  return OFX_OK;
}

// The one collective of the path for a C consumer of libofx.so: ofx_episode_scores + an in-place RCCL sum over the
// ranks of `nccl_comm` (an ncclComm_t the caller created with ncclCommInitRank, one rank per GPU), on the handle's
// stream.  libofx.so does not link RCCL: ncclAllReduce is looked up in the process at the first call (the library the
// caller's communicator came from; otherwise librccl.so.1 is opened), so a single-GPU user never loads it.
#include <dlfcn.h>
typedef int (*ofx_nccl_allreduce_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
extern "C" int ofx_scores_allreduce(ofx_handle *h, void *nccl_comm, int64_t *sums) {
  if (!h || !nccl_comm || !sums) { ofx_set_error("ofx_scores_allreduce: null argument"); return OFX_ERR_INVALID; }
  static ofx_nccl_allreduce_fn fn = nullptr;
  if (!fn) {
    fn = (ofx_nccl_allreduce_fn)dlsym(RTLD_DEFAULT, "ncclAllReduce");
    if (!fn) {
      void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (lib) fn = (ofx_nccl_allreduce_fn)dlsym(lib, "ncclAllReduce");
    }
    if (!fn) { ofx_set_error("ofx_scores_allreduce: ncclAllReduce not found (librccl.so is not loadable)"); return OFX_ERR_STATE; }
  }
  int rc = ofx_episode_scores(h, sums);
  if (rc) return rc;
  const int kNcclInt64 = 4, kNcclSum = 0;  // rccl.h: ncclDataType_t ncclInt64 = 4, ncclRedOp_t ncclSum = 0
  const int e = fn(sums, sums, (size_t)h->cfg.n_ships + 1, kNcclInt64, kNcclSum, nccl_comm, h->stream);
  if (e != 0) { ofx_set_error("ofx_scores_allreduce: ncclAllReduce failed with ncclResult_t %d", e); return OFX_ERR_HIP; }
  return OFX_OK;
}

// -------------------------------------------------------------------- timing
extern "C" int ofx_timer_start(ofx_handle *h) {
  if (!h) { ofx_set_error("null handle"); return OFX_ERR_INVALID; }
  OFX_HIP(hipSetDevice(h->cfg.device));
  if (!h->events) {
    OFX_HIP(hipEventCreate(&h->ev0));
    OFX_HIP(hipEventCreate(&h->ev1));
    h->events = true;
  }
  OFX_HIP(hipEventRecord(h->ev0, h->stream));
  return OFX_OK;
}

extern "C" int ofx_timer_stop(ofx_handle *h, float *ms_host) {
  if (!h || !ms_host || !h->events) { ofx_set_error("ofx_timer_stop without ofx_timer_start"); return OFX_ERR_STATE; }
  OFX_HIP(hipEventRecord(h->ev1, h->stream));
  OFX_HIP(hipEventSynchronize(h->ev1));
  OFX_HIP(hipEventElapsedTime(ms_host, h->ev0, h->ev1));
  return OFX_OK;
}

extern "C" int ofx_event_record(ofx_handle *h, int32_t idx) {
  if (!h || idx < 0 || idx >= OFX_RING_MAX) { ofx_set_error("ofx_event_record: bad index %d", idx); return OFX_ERR_INVALID; }
  OFX_HIP(hipSetDevice(h->cfg.device));
  if (!h->ring) {
    h->ring = (hipEvent_t *)calloc(OFX_RING_MAX, sizeof(hipEvent_t));
    if (!h->ring) { ofx_set_error("out of host memory"); return OFX_ERR_INVALID; }
    h->ring_n = OFX_RING_MAX;
  }
  if (!h->ring[idx]) OFX_HIP(hipEventCreate(&h->ring[idx]));
  OFX_HIP(hipEventRecord(h->ring[idx], h->stream));
  return OFX_OK;
}

extern "C" int ofx_event_elapsed(ofx_handle *h, int32_t a, int32_t b, float *ms_host) {
  if (!h || !ms_host || !h->ring || a < 0 || b < 0 || a >= OFX_RING_MAX || b >= OFX_RING_MAX || !h->ring[a] ||
      !h->ring[b]) {
    ofx_set_error("ofx_event_elapsed: events %d / %d were not recorded", a, b);
    return OFX_ERR_STATE;
  }
  OFX_HIP(hipEventSynchronize(h->ring[b]));
  OFX_HIP(hipEventElapsedTime(ms_host, h->ring[a], h->ring[b]));
  return OFX_OK;
}

extern "C" int ofx_policy_profile(ofx_handle *h, int32_t event_base) {
  if (!h || event_base >= OFX_RING_MAX - 1) { ofx_set_error("ofx_policy_profile: bad event base %d", event_base); return OFX_ERR_INVALID; }
  h->prof_base = event_base < 0 ? -1 : event_base;
  return OFX_OK;
}
