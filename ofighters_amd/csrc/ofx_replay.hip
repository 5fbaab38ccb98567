// ofx_replay.hip - device-side transition capture: the batched form of Trainer.memory = deque(maxlen=memory_size)
// (agents/qlearnIA_V2.py:58), Trainer.remember (:237-238) and the bookkeeping of QlearnIA.play (:370-403) /
// QlearnIA.reset (:360-368).  Every arena owns one memory (the reference has one arena and one shared TRAINER).
//
// Layout in HBM (288 GB per GPU: the whole memory stays resident, nothing goes to the host):
//   frames  [N][F][2][W*H/32] u32   1-bit observation maps (20 KB per map): a ring of F frames per arena.  An arena
//                                    stores a frame only on lock-steps where one of its agents plays; the maps are
//                                    shared by every ship of the arena and by the two transitions that touch them
//                                    (next_state of t-1 -> t, state of t -> t+1), so they are stored once.
//   rows    [N][C] ofx_transition    ring of the last C transitions per arena in append order (lock-step, then ship
//                                    index: the order of request_actions, battleground.py:146-150)
//   per ship [N][M]                  previous_obs / previous_action / previous_pointer (+ the toVector head) and the
//                                    agent's `done` latch
#include "ofx_internal.h"
#include <string.h>

struct ofx_replay {
  int32_t capacity, frames, words;
  uint32_t *frame_bits;    // [N][F][2][words]
  int32_t *frame_tick;     // [N][F]  lock-step stored in the slot, -1 = empty
  int32_t *frame_head;     // [N] next slot
  int32_t *cur_slot;       // [N] slot written by the running capture, -1 = none
  ofx_transition *rows;    // [N][C]
  int32_t *head;           // [N] next write position
  int32_t *count;          // [N] min(appended, C)
  long long *appended;     // [N]
  // per ship
  uint8_t *has_prev, *latched;
  int32_t *prev_iaction, *prev_px, *prev_py, *prev_tick, *prev_slot;
  float *prev_head;        // [N][M][8]
  int32_t *scan_off;       // [N + 1] prefix sums of ofx_replay_gather_valid (kept: no allocation per replay)
};

void ofx_replay_free(ofx_handle *h) {
  ofx_replay *r = h->replay;
  if (!r) return;
  void *ptrs[] = {r->frame_bits, r->frame_tick, r->rows, r->head, r->count, r->appended, r->has_prev, r->latched,
                  r->prev_iaction, r->prev_px, r->prev_py, r->prev_tick, r->prev_head, r->frame_head, r->cur_slot,
                  r->prev_slot, r->scan_off};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  delete r;
  h->replay = nullptr;
}

template <typename T>
static int zalloc(T **p, size_t count, int fill = 0) {
  OFX_HIP(hipMalloc((void **)p, sizeof(T) * count));
  OFX_HIP(hipMemset(*p, fill, sizeof(T) * count));
  return OFX_OK;
}

extern "C" int ofx_replay_create(ofx_handle *h, int32_t capacity, int32_t frames) {
  if (!h) { ofx_set_error("ofx_replay_create: null handle"); return OFX_ERR_INVALID; }
  if (capacity <= 0 || frames < 0 || frames == 1) {
    ofx_set_error("ofx_replay_create: capacity must be > 0 and frames >= 2 (or 0 = capacity + capacity / 4 + 2), got %d, %d",
                  capacity, frames);
    return OFX_ERR_INVALID;
  }
  if (((size_t)h->cfg.width * h->cfg.height) % 128) {
    ofx_set_error("ofx_replay_create: width*height must be a multiple of 128");
    return OFX_ERR_INVALID;
  }
  OFX_HIP(hipSetDevice(h->cfg.device));
  OFX_HIP(hipStreamSynchronize(h->stream));
  ofx_replay_free(h);
  ofx_replay *r = new ofx_replay();
  memset(r, 0, sizeof(*r));
  h->replay = r;
  const size_t N = h->cfg.n_arenas, M = h->cfg.n_ships;
  r->capacity = capacity;
  // C rows made of runs of consecutive plays need C + (number of runs) frames; 2C covers every case
  r->frames = frames ? frames : capacity + capacity / 4 + 2;
  r->words = (int32_t)(((size_t)h->cfg.width * h->cfg.height) >> 5);
  int rc;
#define A(field, count, fill) if ((rc = zalloc(&r->field, (count), (fill)))) { ofx_replay_free(h); return rc; }
  A(frame_bits, (size_t)r->frames * 2 * N * r->words, 0)
  A(frame_tick, N * (size_t)r->frames, 0xFF)
  A(frame_head, N, 0) A(cur_slot, N, 0xFF) A(prev_slot, N * M, 0)
  A(rows, N * (size_t)capacity, 0)
  A(head, N, 0) A(count, N, 0) A(appended, N, 0)
  A(has_prev, N * M, 0) A(latched, N * M, 0)
  A(prev_iaction, N * M, 0) A(prev_px, N * M, 0) A(prev_py, N * M, 0) A(prev_tick, N * M, 0)
  A(prev_head, N * M * 8, 0)
  A(scan_off, N + 1, 0)
#undef A
  OFX_HIP(hipDeviceSynchronize());  // null-stream fills vs the handle's non-blocking stream
  return OFX_OK;
}

extern "C" int ofx_replay_destroy(ofx_handle *h) {
  if (!h) return OFX_OK;
  (void)hipSetDevice(h->cfg.device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  ofx_replay_free(h);
  return OFX_OK;
}

// QlearnIA.reset (qlearnIA_V2.py:360-368): done = False, previous_* = None.  The memory itself survives.
__global__ void k_replay_episode(int N, int M, const uint8_t *arena_mask, uint8_t *has_prev, uint8_t *latched) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * M) return;
  if (arena_mask && !arena_mask[t / M]) return;
  has_prev[t] = 0;
  latched[t] = 0;
}

int ofx_replay_episode_reset(ofx_handle *h, const uint8_t *arena_mask) {
  ofx_replay *r = h->replay;
  if (!r) return OFX_OK;
  const int T = h->cfg.n_arenas * h->cfg.n_ships;
  hipLaunchKernelGGL(k_replay_episode, dim3((T + 255) / 256), dim3(256), 0, h->stream, h->cfg.n_arenas, h->cfg.n_ships,
                     arena_mask, r->has_prev, r->latched);
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}

struct CaptureParams {
  int N, M, W, H, C, F;
  int tick;
  ofx_state st;
  const uint8_t *mask;
  const int32_t *iaction, *ipointer;
  ofx_replay r;
};

// One 64-lane wave per arena, lane = ship.  QlearnIA.play (qlearnIA_V2.py:370-403) per capturing ship:
//   if self.done: return                      -> nothing, not even previous_* changes
//   if obs.done: self.done = True             -> the losing frame is still remembered below
//   if previous_*: remember(previous_obs, previous_action, previous_pointer, obs.reward, obs, obs.done)
//   previous_* = obs, iaction, ipointer
// Rows are appended in ship-index order (ballot prefix), exactly the deque's append order.
__global__ __launch_bounds__(256) void k_replay_capture(CaptureParams p) {
  const int a = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (a >= p.N) return;  // wave-uniform
  const int t = a * p.M + lane;
  const bool ship = lane < p.M && (!p.mask || p.mask[t]);
  const bool plays = ship && !p.r.latched[t];
  float hd[8];
  int done = 0, reward = 0;
  if (plays) {
    reward = p.st.reward[t];  // obs.reward: sampled before Agent.step zeroes it (agent.py:73-74)
    done = p.st.alive[t] ? 0 : 1;
    hd[0] = (float)reward; hd[1] = 1.f;
    hd[2] = (float)p.st.ship_px[t]; hd[3] = (float)p.st.ship_py[t];
    hd[4] = (float)p.W; hd[5] = (float)p.H;
    hd[6] = (float)p.st.ship_x[t]; hd[7] = (float)p.st.ship_y[t];
  }
  // the arena keeps this lock-step's maps iff one of its agents plays (k_replay_frames copies them afterwards)
  int slot = -1;
  if (__ballot(plays)) {
    slot = p.r.frame_head[a];
    if (lane == 0) {
      p.r.frame_head[a] = (slot + 1) % p.F;
      p.r.frame_tick[(size_t)a * p.F + slot] = p.tick;
    }
  }
  if (lane == 0) p.r.cur_slot[a] = slot;
  const bool trans = plays && p.r.has_prev[t];
  const unsigned long long bal = __ballot(trans);
  const int n_new = __popcll(bal);
  const int pos = __popcll(bal & ((1ull << lane) - 1ull));
  const int head = p.r.head[a];
  if (trans) {
    ofx_transition row;
    row.tick_prev = p.r.prev_tick[t];
    row.tick_next = p.tick;
    row.frame_prev = p.r.prev_slot[t];
    row.frame_next = slot;
    row.ship = lane;
    row.iaction = p.r.prev_iaction[t];
    row.px = p.r.prev_px[t];
    row.py = p.r.prev_py[t];
    row.reward = reward;
    row.done = done;
#pragma unroll
    for (int k = 0; k < 8; k++) { row.head_prev[k] = p.r.prev_head[(size_t)t * 8 + k]; row.head_next[k] = hd[k]; }
    p.r.rows[(size_t)a * p.C + (head + pos) % p.C] = row;
  }
  if (plays) {
    if (done) p.r.latched[t] = 1;
    p.r.has_prev[t] = 1;
    p.r.prev_tick[t] = p.tick;
    p.r.prev_slot[t] = slot;
    p.r.prev_iaction[t] = p.iaction[t];
    p.r.prev_px[t] = p.ipointer[2 * t];
    p.r.prev_py[t] = p.ipointer[2 * t + 1];
#pragma unroll
    for (int k = 0; k < 8; k++) p.r.prev_head[(size_t)t * 8 + k] = hd[k];
  }
  if (lane == 0 && n_new) {
    p.r.head[a] = (head + n_new) % p.C;
    p.r.count[a] = min(p.r.count[a] + n_new, p.C);
    p.r.appended[a] += n_new;
  }
}

// copies the current 1-bit maps of every arena that stores a frame this lock-step into its ring slot
__global__ __launch_bounds__(256) void k_replay_frames(int N, int F, int words, const uint32_t *ship_bits,
                                                       const uint32_t *laser_bits, ofx_replay r) {
  const int a = blockIdx.x;
  const int slot = r.cur_slot[a];
  if (slot < 0) return;  // block-uniform
  uint4 *dst = reinterpret_cast<uint4 *>(r.frame_bits + ((size_t)a * F + slot) * 2 * words);
  const uint4 *s0 = reinterpret_cast<const uint4 *>(ship_bits + (size_t)a * words);
  const uint4 *s1 = reinterpret_cast<const uint4 *>(laser_bits + (size_t)a * words);
  const int q = words / 4;
  for (int k = threadIdx.x; k < q; k += 256) { dst[k] = s0[k]; dst[q + k] = s1[k]; }
}

int ofx_policy_results(ofx_handle *h, int32_t **iaction, int32_t **ipointer);  // ofx_policy.hip

extern "C" int ofx_replay_capture(ofx_handle *h, uint32_t tick, const uint8_t *ship_mask, const int32_t *iaction,
                                  const int32_t *ipointer) {
  if (!h) { ofx_set_error("ofx_replay_capture: null handle"); return OFX_ERR_INVALID; }
  ofx_replay *r = h->replay;
  if (!r) { ofx_set_error("ofx_replay_capture before ofx_replay_create"); return OFX_ERR_STATE; }
  if (!h->spawned) { ofx_set_error("You must execute analyse_battleground first."); return OFX_ERR_STATE; }
  if ((int32_t)tick < 0) { ofx_set_error("ofx_replay_capture: tick must be < 2^31"); return OFX_ERR_INVALID; }
  if (h->cfg.n_ships > OFX_WAVE) { ofx_set_error("ofx_replay_capture: n_ships > 64"); return OFX_ERR_INVALID; }
  OFX_HIP(hipSetDevice(h->cfg.device));
  int rc;
  if (!iaction || !ipointer) {  // the results the last ofx_policy_forward / ofx_policy_explore left in the workspace
    int32_t *ia, *ip;
    if ((rc = ofx_policy_results(h, &ia, &ip))) return rc;
    if (!iaction) iaction = ia;
    if (!ipointer) ipointer = ip;
  }
  // the observation maps of this lock-step (the same 1-bit maps the policy trunk reads)
  if ((rc = ofx_launch_raster(h, OFX_MAP_BITS_LSB, nullptr, nullptr))) return rc;
  CaptureParams p;
  p.N = h->cfg.n_arenas; p.M = h->cfg.n_ships; p.W = h->cfg.width; p.H = h->cfg.height; p.C = r->capacity;
  p.F = r->frames;
  p.tick = (int)tick; p.st = h->st; p.mask = ship_mask; p.iaction = iaction; p.ipointer = ipointer; p.r = *r;
  hipLaunchKernelGGL(k_replay_capture, dim3((p.N + 3) / 4), dim3(256), 0, h->stream, p);
  OFX_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_replay_frames, dim3((unsigned)p.N), dim3(256), 0, h->stream, p.N, r->frames, r->words,
                     (const uint32_t *)h->maps[OFX_MAP_BITS_LSB][0], (const uint32_t *)h->maps[OFX_MAP_BITS_LSB][1], *r);
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}

// first-seen deaths of the selected ships (QlearnIA.play's done latch, agents/qlearnIA_V2.py:376-384)
__global__ void k_first_done(int n, const uint8_t *alive, const uint8_t *mask, uint8_t *seen, int *count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool first = false;
  if (i < n && alive[i] == 0 && (!mask || mask[i]) && !seen[i]) {
    seen[i] = 1;
    first = true;
  }
  const unsigned long long b = __builtin_amdgcn_ballot_w64(first);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(count, __popcll(b));
}

extern "C" int ofx_agents_first_done(ofx_handle *h, const uint8_t *ship_mask, uint8_t *seen, int32_t *count_host) {
  if (!h || !seen || !count_host) { ofx_set_error("ofx_agents_first_done: null argument"); return OFX_ERR_INVALID; }
  if (!h->spawned) { ofx_set_error("ofx_agents_first_done before ofx_spawn"); return OFX_ERR_STATE; }
  OFX_HIP(hipSetDevice(h->cfg.device));
  int rc;
  if ((rc = ofx_ensure_scratch(h, 256))) return rc;
  int *cnt = (int *)h->scratch;
  OFX_HIP(hipMemsetAsync(cnt, 0, sizeof(int), h->stream));
  const int n = h->cfg.n_arenas * h->cfg.n_ships;
  hipLaunchKernelGGL(k_first_done, dim3((n + 255) / 256), dim3(256), 0, h->stream, n, h->st.alive, ship_mask, seen, cnt);
  OFX_HIP(hipGetLastError());
  OFX_HIP(hipMemcpyAsync(count_host, cnt, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  OFX_HIP(hipStreamSynchronize(h->stream));
  return OFX_OK;
}

extern "C" int ofx_replay_count(ofx_handle *h, int32_t *count_host, int64_t *appended_host) {
  if (!h || !h->replay) { ofx_set_error("ofx_replay_count: no replay memory"); return OFX_ERR_STATE; }
  OFX_HIP(hipSetDevice(h->cfg.device));
  OFX_HIP(hipStreamSynchronize(h->stream));
  const size_t N = h->cfg.n_arenas;
  if (count_host) OFX_HIP(hipMemcpy(count_host, h->replay->count, sizeof(int32_t) * N, hipMemcpyDeviceToHost));
  if (appended_host) OFX_HIP(hipMemcpy(appended_host, h->replay->appended, sizeof(int64_t) * N, hipMemcpyDeviceToHost));
  return OFX_OK;
}

// rows of one arena, oldest first (list(memory))
extern "C" int ofx_replay_rows_host(ofx_handle *h, int32_t arena, ofx_transition *rows_host, int32_t *n_host) {
  if (!h || !h->replay || !rows_host || !n_host) { ofx_set_error("ofx_replay_rows_host: bad argument"); return OFX_ERR_INVALID; }
  if (arena < 0 || arena >= h->cfg.n_arenas) { ofx_set_error("ofx_replay_rows_host: arena out of range"); return OFX_ERR_INVALID; }
  ofx_replay *r = h->replay;
  OFX_HIP(hipSetDevice(h->cfg.device));
  OFX_HIP(hipStreamSynchronize(h->stream));
  int32_t head, count;
  OFX_HIP(hipMemcpy(&head, r->head + arena, 4, hipMemcpyDeviceToHost));
  OFX_HIP(hipMemcpy(&count, r->count + arena, 4, hipMemcpyDeviceToHost));
  const int C = r->capacity;
  const int first = ((head - count) % C + C) % C;
  const ofx_transition *base = r->rows + (size_t)arena * C;
  const int n1 = min(count, C - first);
  if (n1 > 0) OFX_HIP(hipMemcpy(rows_host, base + first, sizeof(ofx_transition) * n1, hipMemcpyDeviceToHost));
  if (count > n1) OFX_HIP(hipMemcpy(rows_host + n1, base, sizeof(ofx_transition) * (count - n1), hipMemcpyDeviceToHost));
  *n_host = count;
  return OFX_OK;
}

// the 1-bit maps of one stored lock-step of one arena (pixel p -> bit (p & 7) of byte p >> 3, i.e.
// numpy.unpackbits(..., bitorder='little')); OFX_ERR_STATE when the frame has left the ring
extern "C" int ofx_replay_frame_host(ofx_handle *h, int32_t arena, int32_t tick, void *ship_bits_host,
                                     void *laser_bits_host) {
  if (!h || !h->replay || !ship_bits_host || !laser_bits_host) { ofx_set_error("ofx_replay_frame_host: bad argument"); return OFX_ERR_INVALID; }
  if (arena < 0 || arena >= h->cfg.n_arenas || tick < 0) { ofx_set_error("ofx_replay_frame_host: out of range"); return OFX_ERR_INVALID; }
  ofx_replay *r = h->replay;
  OFX_HIP(hipSetDevice(h->cfg.device));
  OFX_HIP(hipStreamSynchronize(h->stream));
  int32_t *ticks = (int32_t *)malloc(sizeof(int32_t) * r->frames);
  if (!ticks) { ofx_set_error("ofx_replay_frame_host: out of host memory"); return OFX_ERR_INVALID; }
  hipError_t e = hipMemcpy(ticks, r->frame_tick + (size_t)arena * r->frames, sizeof(int32_t) * r->frames, hipMemcpyDeviceToHost);
  int f = -1;
  for (int i = 0; e == hipSuccess && i < r->frames; i++) if (ticks[i] == tick) f = i;
  free(ticks);
  OFX_HIP(e);
  if (f < 0) { ofx_set_error("ofx_replay_frame_host: lock-step %d is not in the frame ring of arena %d", tick, arena); return OFX_ERR_STATE; }
  const size_t wb = (size_t)r->words * 4;
  const uint32_t *slot = r->frame_bits + ((size_t)arena * r->frames + f) * 2 * r->words;
  OFX_HIP(hipMemcpy(ship_bits_host, slot, wb, hipMemcpyDeviceToHost));
  OFX_HIP(hipMemcpy(laser_bits_host, slot + r->words, wb, hipMemcpyDeviceToHost));
  return OFX_OK;
}

// ---- minibatch: random.sample(memory, min(batch, len(memory))) per arena (qlearnIA_V2.py:241-243) ----------------
// Floyd's subset sampling (uniform over the subsets, no replacement) on Philox draws: counter (global arena, j,
// draw, stream 3).  Transitions whose `state` frame has already left the arena's frame ring are not eligible (only
// possible when the ring is shorter than the rows need: C rows in runs of consecutive plays use C + #runs frames).  slot[a][j] indexes the arena's rows oldest-first, -1 pads.
#define OFX_STREAM_REPLAY 3u
__global__ void k_replay_sample(int N, int C, int F, int batch, int arena_base, uint32_t k0, uint32_t k1, uint32_t draw,
                                ofx_replay r, int32_t *slot, int32_t *n_out) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= N) return;
  const int count = r.count[a], head = r.head[a];
  const int first = ((head - count) % C + C) % C;
  const ofx_transition *rows = r.rows + (size_t)a * C;
  int skip = 0;  // rows are chronological: the expired ones are the oldest
  while (skip < count) {
    const ofx_transition &o = rows[(first + skip) % C];
    if (r.frame_tick[(size_t)a * F + o.frame_prev] == o.tick_prev) break;
    skip++;
  }
  const int valid = count - skip, n = min(batch, valid);
  int32_t *out = slot + (size_t)a * batch;
  for (int j = 0; j < n; j++) {
    const int top = valid - n + j;  // draw t in [0, top]
    uint32_t rr[4];
    ofx_philox4x32_10((uint32_t)(arena_base + a), (uint32_t)j, draw, OFX_STREAM_REPLAY, k0, k1, rr);
    int tsel = ofx_draw_int(rr[0], top);
    for (int q = 0; q < j; q++) if (out[q] == skip + tsel) { tsel = top; break; }
    out[j] = skip + tsel;
  }
  for (int j = n; j < batch; j++) out[j] = -1;
  if (n_out) n_out[a] = n;
}

extern "C" int ofx_replay_sample(ofx_handle *h, uint64_t seed, uint32_t draw, int32_t batch, int32_t *slot,
                                 int32_t *n_sampled) {
  if (!h || !h->replay || !slot || batch <= 0) { ofx_set_error("ofx_replay_sample: bad argument"); return OFX_ERR_INVALID; }
  ofx_replay *r = h->replay;
  OFX_HIP(hipSetDevice(h->cfg.device));
  const int N = h->cfg.n_arenas;
  hipLaunchKernelGGL(k_replay_sample, dim3((N + 63) / 64), dim3(64), 0, h->stream, N, r->capacity, r->frames, batch,
                     h->cfg.arena_base, (uint32_t)seed, (uint32_t)(seed >> 32), draw, *r, slot, n_sampled);
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}

// ---- gather a sampled minibatch into dense tensors (the arrays Trainer.replay builds, qlearnIA_V2.py:246-283) -------
struct GatherParams {
  int N, C, F, batch, words;
  ofx_replay r;
  const int32_t *slot;
  ofx_transition *rows;      // [N][batch]
  uint32_t *bits_prev, *bits_next;  // [N][batch][2][words] or null
};

// one workgroup per (arena, j): row copy + the two frames' bit maps (16-byte loads/stores)
__global__ __launch_bounds__(256) void k_replay_gather(GatherParams p) {
  const int a = blockIdx.x / p.batch, j = blockIdx.x - a * p.batch;
  const int s = p.slot[(size_t)a * p.batch + j];
  ofx_transition *dst = p.rows + (size_t)a * p.batch + j;
  if (s < 0) {  // padding: an all-zero row with ship = -1 and empty maps (never uninitialised memory)
    if (threadIdx.x < sizeof(ofx_transition) / 4) ((int32_t *)dst)[threadIdx.x] = threadIdx.x == 4 ? -1 : 0;
    uint32_t *pads[2] = {p.bits_prev, p.bits_next};
    for (int w = 0; w < 2; w++) {
      if (!pads[w]) continue;
      uint4 *out = reinterpret_cast<uint4 *>(pads[w] + ((size_t)a * p.batch + j) * 2 * p.words);
      for (int k = threadIdx.x; k < 2 * p.words / 4; k += 256) out[k] = make_uint4(0u, 0u, 0u, 0u);
    }
    return;
  }
  const int count = p.r.count[a], head = p.r.head[a];
  const int first = ((head - count) % p.C + p.C) % p.C;
  const ofx_transition *src = p.r.rows + (size_t)a * p.C + (first + s) % p.C;
  if (threadIdx.x < sizeof(ofx_transition) / 4) ((int32_t *)dst)[threadIdx.x] = ((const int32_t *)src)[threadIdx.x];
  const int slots[2] = {src->frame_prev, src->frame_next};
  uint32_t *outs[2] = {p.bits_prev, p.bits_next};
  for (int w = 0; w < 2; w++) {
    if (!outs[w]) continue;
    const uint4 *in = reinterpret_cast<const uint4 *>(p.r.frame_bits + ((size_t)a * p.F + slots[w]) * 2 * p.words);
    uint4 *out = reinterpret_cast<uint4 *>(outs[w] + ((size_t)a * p.batch + j) * 2 * p.words);
    for (int k = threadIdx.x; k < 2 * p.words / 4; k += 256) out[k] = in[k];
  }
}

extern "C" int ofx_replay_gather(ofx_handle *h, const int32_t *slot, int32_t batch, ofx_transition *rows,
                                 void *bits_prev, void *bits_next) {
  if (!h || !h->replay || !slot || !rows || batch <= 0) { ofx_set_error("ofx_replay_gather: bad argument"); return OFX_ERR_INVALID; }
  ofx_replay *r = h->replay;
  OFX_HIP(hipSetDevice(h->cfg.device));
  GatherParams p;
  p.N = h->cfg.n_arenas; p.C = r->capacity; p.F = r->frames; p.batch = batch; p.words = r->words;
  p.r = *r; p.slot = slot; p.rows = rows; p.bits_prev = (uint32_t *)bits_prev; p.bits_next = (uint32_t *)bits_next;
  hipLaunchKernelGGL(k_replay_gather, dim3((unsigned)(p.N * batch)), dim3(256), 0, h->stream, p);
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}

// ---- the same minibatch without padding: only the rows that exist, packed (Trainer.replay never pads: its batch is
// min(batch_size, len(memory)) real transitions, qlearnIA_V2.py:241-243) ------------------------------------------------
// exclusive scan of n_sampled[N] by one workgroup -> off[N], total in off[N]
__global__ __launch_bounds__(1024) void k_replay_scan(int N, const int32_t *n_sampled, int32_t *off) {
  __shared__ int part[1024];
  const int tid = threadIdx.x, per = (N + 1023) / 1024, lo = min(tid * per, N), hi = min(lo + per, N);
  int sum = 0;
  for (int i = lo; i < hi; i++) sum += n_sampled[i];
  part[tid] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan of the partial sums
    const int v = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = tid ? part[tid - 1] : 0;
  for (int i = lo; i < hi; i++) { off[i] = run; run += n_sampled[i]; }
  if (tid == 1023) off[N] = part[1023];
}

// one workgroup per (arena, j): sampled entry j of arena a is packed row off[a] + j - first, when it falls into
// [0, max_rows)
__global__ __launch_bounds__(256) void k_replay_gather_valid(GatherParams p, const int32_t *n_sampled, const int32_t *off, int first,
                                                             int max_rows) {
  const int a = blockIdx.x / p.batch, j = blockIdx.x - a * p.batch;
  if (j >= n_sampled[a]) return;
  const int d = off[a] + j - first;
  if (d < 0 || d >= max_rows) return;
  const int s = p.slot[(size_t)a * p.batch + j];
  const int count = p.r.count[a], head = p.r.head[a];
  const int first_row = ((head - count) % p.C + p.C) % p.C;
  const ofx_transition *src = p.r.rows + (size_t)a * p.C + (first_row + s) % p.C;
  ofx_transition *dst = p.rows + d;
  if (threadIdx.x < sizeof(ofx_transition) / 4) ((int32_t *)dst)[threadIdx.x] = ((const int32_t *)src)[threadIdx.x];
  const int slots[2] = {src->frame_prev, src->frame_next};
  uint32_t *outs[2] = {p.bits_prev, p.bits_next};
  for (int w = 0; w < 2; w++) {
    if (!outs[w]) continue;
    const uint4 *in = reinterpret_cast<const uint4 *>(p.r.frame_bits + ((size_t)a * p.F + slots[w]) * 2 * p.words);
    uint4 *out = reinterpret_cast<uint4 *>(outs[w] + (size_t)d * 2 * p.words);
    for (int k = threadIdx.x; k < 2 * p.words / 4; k += 256) out[k] = in[k];
  }
}

extern "C" int ofx_replay_gather_valid(ofx_handle *h, const int32_t *slot, const int32_t *n_sampled, int32_t batch,
                                       int32_t first, int32_t max_rows, ofx_transition *rows, void *bits_prev,
                                       void *bits_next, int32_t *n_rows_host) {
  if (!h || !h->replay || !slot || !n_sampled || !rows || !n_rows_host || batch <= 0 || first < 0 || max_rows <= 0) {
    ofx_set_error("ofx_replay_gather_valid: bad argument");
    return OFX_ERR_INVALID;
  }
  ofx_replay *r = h->replay;
  OFX_HIP(hipSetDevice(h->cfg.device));
  const int N = h->cfg.n_arenas;
  int32_t *off = r->scan_off;
  hipLaunchKernelGGL(k_replay_scan, dim3(1), dim3(1024), 0, h->stream, N, n_sampled, off);
  GatherParams p;
  p.N = N; p.C = r->capacity; p.F = r->frames; p.batch = batch; p.words = r->words;
  p.r = *r; p.slot = slot; p.rows = rows; p.bits_prev = (uint32_t *)bits_prev; p.bits_next = (uint32_t *)bits_next;
  hipLaunchKernelGGL(k_replay_gather_valid, dim3((unsigned)(N * batch)), dim3(256), 0, h->stream, p, n_sampled, (const int32_t *)off,
                     first, max_rows);
  hipError_t e = hipGetLastError();
  int32_t total = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&total, off + N, sizeof(total), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) { ofx_set_error("ofx_replay_gather_valid: %s", hipGetErrorString(e)); return OFX_ERR_HIP; }
  *n_rows_host = total - first < 0 ? 0 : (total - first > max_rows ? max_rows : total - first);
  return OFX_OK;
}
