// ofx_internal.h - handle layout and helpers shared by the HIP translation
// units of libofx.so (gfx950 only; no CPU fallback, no CUDA shims).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ofx.h"

#define OFX_WAVE 64
#define OFX_ARENAS_PER_BLOCK 4 /* one 64-lane wave per arena, 256-thread blocks */

void ofx_set_error(const char *fmt, ...);

#define OFX_HIP(call)                                                                  \
  do {                                                                                 \
    hipError_t _e = (call);                                                            \
    if (_e != hipSuccess) {                                                            \
      ofx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__,   \
                    __LINE__);                                                         \
      return OFX_ERR_HIP;                                                              \
    }                                                                                  \
  } while (0)

// Structure-of-arrays arena state, all in HBM.  Ships: [N][M]; lasers [N][L];
// per-arena scalars [N].  A wave reads one arena's M ship words / 64 laser
// slots as one coalesced segment per array.
struct ofx_state {
  int32_t *ship_x, *ship_y, *ship_px, *ship_py, *hull;
  int32_t *reward, *score, *obs_reward, *last_score;
  uint8_t *alive;
  int16_t *killer;
  int32_t *time, *n_lasers;
  double *laser_x, *laser_y, *laser_dx, *laser_dy;
  uint8_t *laser_owner, *laser_dead;
  unsigned long long *overflow;    // [1]
  long long *episode_sums;         // [M+1]
};

struct ofx_replay;  // ofx_replay.hip

struct ofx_handle {
  ofx_config cfg;
  hipStream_t stream;
  bool own_stream;
  bool spawned;
  ofx_state st;
  double hit_thresh;               // largest d2 with rn(sqrt(d2)) <= r_laser + r_ship
  // observation maps (lazily allocated per type): [type][which]
  void *maps[5][2];
  int32_t *bot_behaviours;         // device [M]
  // scratch for nn / policy
  void *scratch;
  size_t scratch_bytes;
  hipEvent_t ev0, ev1;
  bool events;
  hipEvent_t *ring;                // numbered events for ofx_event_record
  int ring_n;
  int prof_base;                   // ofx_policy_profile: next event pair, -1 = off
  ofx_replay *replay;              // transition memory (ofx_replay_create), null = none
  void *aux;                       // temporaries of ofx_dqn_targets (grown on demand)
  size_t aux_bytes;
  void *fitws;                     // workspace of ofx_dqn_fit / ofx_dqn_fit_reference, kept between calls (a fresh 4 GB
  size_t fitws_bytes;              // hipMalloc per replay cost ~60 ms of mapping: r03), grown on demand
  void *fitws2;                    // the reference form's dense targets, kept likewise
  size_t fitws2_bytes;
  int32_t *counter;                // [4] small device counter of the fit's argument checks (allocated on first use)
  float *prep;                     // prepared policy weights (BN folded, phase weights, tables): ofx_policy.hip
  float *prep_tmp;                 // the same for a blob that is not the pinned one (rebuilt per forward)
  const float *prep_pinned;        // the blob `prep` was built from while it is pinned (ofx_policy_pin_weights)
  int opt_trunk_fuse;              // OFX_OPT_TRUNK_FUSE: 0 auto, 1 always, 2 never
  int n_cus;                       // compute units of the device (grid of the persistent trunk kernel)
  bool opt_trunk_plain, opt_frames_ref, opt_bilinear_legacy;  // ofx_set_option
  int opt_policy_lowp;             // OFX_OPT_POLICY_BF16: 0 fp32, 1 bf16 operands, 2 fp16 operands (opt-in)
  bool opt_fit_plain;              // OFX_OPT_FIT_PLAIN: the fit's layer-by-layer form (test reference)
  bool opt_trunk_sparse;           // OFX_OPT_TRUNK_SPARSE: the streaming trunk skips constant windows (exact, opt-in)
  unsigned long long *trunk_stat;  // [4] device counters of the sparse trunk (allocated with the option)
};
#define OFX_RING_MAX 65536         /* numbered events of ofx_event_record */

// kernels / launchers implemented in the other translation units
int ofx_launch_step(ofx_handle *h, const ofx_action *actions);
int ofx_launch_raster(ofx_handle *h, int map_type, void *ship_map, void *laser_map);
int ofx_ensure_scratch(ofx_handle *h, size_t bytes);
void ofx_replay_free(ofx_handle *h);
int ofx_replay_episode_reset(ofx_handle *h, const uint8_t *arena_mask);
// C[M][N] = act(A[M][K] (lda) x B[K][N] (ldb) + bias[N]) on the f32 MFMA (k_gemm_f32, ofx_policy.hip)
int ofx_launch_gemm(ofx_handle *h, const float *A, int lda, const float *B, int ldb, const float *bias, float *C, int ldc,
                    int M, int N, int K, int relu, const int32_t *live = nullptr);
int ofx_launch_conv1_lut(ofx_handle *h, const void *bits, int n, const float *lut, float *out);  // k_conv1_lut, caller's table
int ofx_policy_weights_updated(ofx_handle *h, const float *weights);  // the blob was changed in place (ofx_dqn_fit)
// model.predict on n stored observations: act_values [n][2], heatmap [n][H][W], ptr_max [n] (any may be null)
int ofx_policy_predict_obs(ofx_handle *h, const float *weights, int32_t n_obs, const void *bits, const float *vec8,
                           float *act_values, float *heatmap, float *ptr_max);

#define OFX_MAP_BITS_LSB 4 /* internal: 1 bit / cell, pixel p -> bit (p & 31) of word p >> 5 */

// Philox4x32-10 counter RNG (Salmon et al. SC'11); same constants / round
// structure as the oracle's restatement so both produce one stream.
__host__ __device__ inline void ofx_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                  uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
#define OFX_STREAM_BOT 0u
#define OFX_STREAM_RESET 1u

__host__ __device__ inline int32_t ofx_draw_int(uint32_t r, int32_t n_inclusive) {
  return (int32_t)(((uint64_t)r * (uint64_t)(n_inclusive + 1)) >> 32);
}
