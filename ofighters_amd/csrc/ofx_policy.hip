// ofx_policy.hip - bi-head "pointer_model" forward for every (arena, ship)
// (agents/qlearnIA_V2.py:123-190 graph, :206-220 inference glue, :447-454
// action packing).  fp32 like Keras.  Conventions (BN eps 1e-3, HWIO kernels,
// (h,w,c) flatten, vector-first concat, half-pixel bilinear x2) are the ones
// declared in oracle/policy_oracle.c; parity of this path is UNPINNED
// (keras/tensorflow and weights are absent), it is checked against that
// restatement with an fp32 tolerance.
//
// Work split (what is shared per arena):
//   trunk   4 x [conv3x3 + BN + ReLU + maxpool2]   once per ARENA  (image is the same for its ships)
//   dense1  [5008 -> 100]: the 5000 trunk features once per arena on MFMA
//           (v_mfma_f32_32x32x2_f32, exact fp32), the 8-scalar head per ship
//   head-1  dense2 + output1 per ship (VALU, tiny)
//   head-2  updense1 [100 -> 625] on MFMA, then 4 x [bilinear x2 + conv3x3]
//           per ship; the last conv (8 -> 1 at 400x400) is evaluated in the
//           4-phase low-resolution form with the arg-max fused, so the
//           (400,400) heat-map is only materialised on request.
//
// Direct convolutions are fp32 VALU kernels: an LDS-staged input tile (halo 1,
// bilinear upsampling fused into the staging), a 2x2 register tile of outputs
// x all output channels per thread, weights as wave-uniform scalar operands
// (s_load + v_fma with an SGPR source).  BatchNorm is folded into the conv
// weights by k_policy_prepare.
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <utility>

#ifndef OFX_HTB_UNROLL
#define OFX_HTB_UNROLL 1  // stage B of k_head_tail: unroll factor of the M-tile pair loop
#endif
#ifndef OFX_HTC_H64
#define OFX_HTC_H64 1     // stage C: second LDS read of a row as b64 instead of b128
#endif
#ifndef OFX_HTB_WEARLY
#define OFX_HTB_WEARLY 0  // stage B weights: 0 loaded at the top of stage B, 1 at the top of the tile, 2 behind stage A
#endif
#ifndef OFX_HTC_MINI
#define OFX_HTC_MINI 0    // stage C: 1 = the last 16 lane-tasks of a tile as 64 one-pixel lanes instead of a 7th wave-pass (measured neutral); 2 = all of pass 1 as nine one-pixel quarter passes (1.5 ms SLOWER: four times the LDS reads per MFMA)
#endif
#ifndef OFX_XCD_SWIZZLE
#define OFX_XCD_SWIZZLE 0  // k_convm: contiguous tile ranges per XCD (measured neutral: conv2 2.79 ms either way)
#endif
#ifndef OFX_CONV1_V4
#define OFX_CONV1_V4 1    // k_conv1_lut: four pooled pixels per thread, 16-byte stores
#endif
#ifndef OFX_HTA_FEWBAR
#define OFX_HTA_FEWBAR 1  // k_head_tail border tiles: two barriers fewer (frame lines built in stage A's phase)
#endif
#ifndef OFX_HT_PIPE
#define OFX_HT_PIPE 0     // k_head_tail: next tile patch committed at the end of the tile, two barriers per tile fewer (measured 0.15 ms SLOWER)
#endif
#ifndef OFX_CONVM_VOLA
#define OFX_CONVM_VOLA 0  // k_convm: single ds_read_b32 per A operand, no ds_read2 pairing (measured SLOWER: conv2 2.84 vs 2.74 ms)
#endif
#ifndef OFX_ABLATE_HOOKS
#define OFX_ABLATE_HOOKS 0  // 1: the diagnostic OFX_CONV_ABLATE / OFX_HT_ABLATE switches are compiled into the kernels
#endif
#if OFX_ABLATE_HOOKS
#define OFX_ABL(p) ((p).ablate)
#else
#define OFX_ABL(p) 0       // the hooks cost branches and, in k_convm, eight accumulator copies per M-tile pair
#endif
#ifndef OFX_CONV2_SHAPE
#define OFX_CONV2_SHAPE 0  // conv2 tile: 0 = 4 rows x 208, 1 = 4 rows x 112, 2 = 8 rows x 112
#endif
#ifndef OFX_HT_FACC_LIGHT
#define OFX_HT_FACC_LIGHT 1  // heat-map frame corrections of a half computed by the light wave
#endif
#ifndef OFX_HTA_ILP
#define OFX_HTA_ILP 0     // k_head_tail stage A: M-tiles of a wave as interleaved MFMA chains (measured neutral)
#endif
#ifndef OFX_HTB_TRIPLE
#define OFX_HTB_TRIPLE 0  // stage B: the 33rd M-tile as a third chain of the light wave last iteration (measured 0.15 ms slower)
#endif
#ifndef OFX_HTC_FENCE
#define OFX_HTC_FENCE 1   // stage C of k_head_tail: hard scheduling fences between the pipeline steps
#endif
#ifndef OFX_HTC_GROUP
#define OFX_HTC_GROUP 0   // ... and the read / MFMA order inside a step (measured slower with the fences)
#endif

#include "ofx_internal.h"
#include "ofx_head.h"

#define PS 400 /* the model's fixed input side: Input((DEFAULT_WIDTH, DEFAULT_HEIGHT, 2)) */

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int N, class F, int... I>
__device__ __forceinline__ void static_for_impl(F &f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &f) {
  static_for_impl<N>(f, std::make_integer_sequence<int, N>{});
}
typedef float f32x16 __attribute__((ext_vector_type(16)));

static const int kTrunkCin[4] = {2, 8, 8, 8};
static const int kUpCin[4] = {1, 2, 4, 8};
static const int kUpCout[4] = {2, 4, 8, 1};

// ---- user blob layout (identical to oracle/policy_oracle.c) -------------------
static int policy_layout(int32_t *offset, int32_t *count) {
  int n = 0, off = 0;
#define T(c) do { offset[n] = off; count[n] = (c); off += (c); n++; } while (0)
  for (int i = 0; i < 4; i++) { T(9 * kTrunkCin[i] * 8); T(8); T(8); T(8); T(8); T(8); }
  T(5008 * 100); T(100);
  T(100 * 50); T(50);
  T(50 * 2); T(2);
  T(100 * 625); T(625);
  for (int i = 0; i < 3; i++) { int co = kUpCout[i]; T(9 * kUpCin[i] * co); T(co); T(co); T(co); T(co); T(co); }
  T(9 * 8 * 1); T(1);
#undef T
  offset[n] = off;
  return n;
}

extern "C" int ofx_policy_layout(const ofx_handle *h, ofx_policy_desc *desc) {
  (void)h;
  if (!desc) { ofx_set_error("ofx_policy_layout: null desc"); return OFX_ERR_INVALID; }
  memset(desc, 0, sizeof(*desc));
  desc->n_tensors = policy_layout(desc->offset, desc->count);
  desc->n_floats = desc->offset[desc->n_tensors];
  return OFX_OK;
}

// ---- prepared (BN-folded) weights ------------------------------------------------
struct PrepLayout {
  int tw[4], tb[4];   // trunk folded kernels [9][cin][8], biases [8]
  int uw[3], ub[3];   // upconv1..3 folded
  int bg[4];          // background response after trunk layer i ([8] each): the value every output channel takes
                      // where the whole receptive window shows empty space
  int w3mf;           // [half 2][k = tap*4 + ci (36)][n = phase*4 + co_local (16)]  MFMA B operand
  int w2mf;           // upconv2 in phase form: [k = tap*2 + ci (18, padded to 20)][n = phase*4 + co (16)]
  int w2fr;           // [variant top|bottom|left|right][20][16]: w2mf with the taps that fall into the zero padding of the
                      // variant's frame line dropped (k_head_frames: exact frame bands of uprelu2)
  int w3fr;           // [variant][k = tap*4 + ci (36)][n = parity*8 + co (16)]: upconv3 phase weights of one frame line
                      // (top / bottom: row phase fixed, n's parity = column phase; left / right the other way round)
  int w4eff;          // [4 phases][9 low-res taps][8 ci] (v1 kernel)
  int w4eff_c;        // [8 ci][4 phases][9 taps] (fused kernel: one contiguous slice per input channel)
  int w4raw;          // [9][8]
  int efr;            // [line h|v][side first|last][parity 2][low-res offset 3][ci 8]: phase weights of the taps of
                      // upconv4 that fall into the zero padding of a frame pixel (k_head_tail border pass)
  int b4;             // [1]
  int wbm[3];         // conv2..4: the banded B operand of k_convm laid out per lane: [j 24][lane 64]
  int lut1;           // conv1 on the binary maps as a table: [ci 2][3x3 bit pattern 512][co 8] = sum of the folded
                      // weights of the set taps (k_conv1_lut)
  int total;
};

static PrepLayout prep_layout() {
  PrepLayout L;
  int off = 0;
  for (int i = 0; i < 4; i++) { L.tw[i] = off; off += 9 * kTrunkCin[i] * 8; L.tb[i] = off; off += 8; }
  for (int i = 0; i < 3; i++) { L.uw[i] = off; off += 9 * kUpCin[i] * kUpCout[i]; L.ub[i] = off; off += kUpCout[i]; }
  for (int i = 0; i < 4; i++) { L.bg[i] = off; off += 8; }
  L.w3mf = off; off += 36 * 32;
  L.w2mf = off; off += 20 * 16;
  L.w2fr = off; off += 4 * 20 * 16;
  L.w3fr = off; off += 4 * 36 * 16;
  L.w4eff = off; off += 4 * 9 * 8;
  L.w4eff_c = off; off += 8 * 4 * 9;
  L.w4raw = off; off += 72;
  L.efr = off; off += 2 * 2 * 2 * 3 * 8;
  L.b4 = off; off += 1;
  off += 8;                      // 8 zeros: the background of conv1's binary input
  off = (off + 3) & ~3;
  L.lut1 = off; off += 2 * 512 * 8;
  for (int i = 0; i < 3; i++) { L.wbm[i] = off; off += 24 * 64; }
  L.total = (off + 63) & ~63;
  return L;
}

struct PrepParams {
  const float *w;
  float *prep;
  int src_k[7], src_b[7], src_g[7], cin[7], cout[7], dst_w[7], dst_b[7];
  int src_k4, src_b4, dst_w4eff, dst_w4raw, dst_b4, dst_efr;
  int dst_w4eff_c, dst_w3mf, dst_w2mf, dst_w2fr, dst_w3fr, dst_lut1, dst_wbm[3];
  int phase;
  int dst_bg[4];
};

// interpolation coefficients of the x2 half-pixel bilinear: output row 2i+a, conv
// tap dy in {-1,0,1} touches low-res rows i-1, i, i+1 with these weights
__device__ inline float up_coef(int a, int dy, int t) {
  // a=0: dy-1 -> row 2i-1 = .75 L[i-1] + .25 L[i]; dy0 -> .25 L[i-1] + .75 L[i]; dy+1 -> .75 L[i] + .25 L[i+1]
  // a=1: dy-1 -> row 2i   = .25 L[i-1] + .75 L[i]; dy0 -> .75 L[i] + .25 L[i+1]; dy+1 -> .25 L[i] + .75 L[i+1]
  const float c0[3][3] = {{0.75f, 0.25f, 0.f}, {0.25f, 0.75f, 0.f}, {0.f, 0.75f, 0.25f}};
  const float c1[3][3] = {{0.25f, 0.75f, 0.f}, {0.f, 0.75f, 0.25f}, {0.f, 0.25f, 0.75f}};
  return a ? c1[dy][t] : c0[dy][t];
}

// Two launches: phase 0 (many workgroups) folds and builds every table that depends on the raw weights only; phase 1
// builds what needs the folded kernels (k_convm's per-lane B operands, the background chain).
__global__ void k_policy_prepare(PrepParams p) {
  const int ltid = threadIdx.x;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nthr = gridDim.x * blockDim.x;
  if (p.phase == 0) {
  for (int l = 0; l < 7; l++) {  // BN fold: y = (conv + b) * inv + (beta - mean * inv)
    const int cin = p.cin[l], cout = p.cout[l];
    const float *g = p.w + p.src_g[l];  // gamma, beta, mean, var consecutive, each [cout]
    for (int e = tid; e < 9 * cin * cout; e += nthr) {
      const int co = e % cout;
      const float inv = g[co] / sqrtf(g[3 * cout + co] + 1e-3f);
      p.prep[p.dst_w[l] + e] = p.w[p.src_k[l] + e] * inv;
    }
    for (int co = tid; co < cout; co += nthr) {
      const float inv = g[co] / sqrtf(g[3 * cout + co] + 1e-3f);
      p.prep[p.dst_b[l] + co] = p.w[p.src_b[l] + co] * inv + (g[cout + co] - g[2 * cout + co] * inv);
    }
  }
  // upconv4 (linear, no BN): effective weights of the 4 output phases on the low-res grid
  for (int e = tid; e < 4 * 9 * 8; e += nthr) {
    const int ci = e % 8, tap = (e / 8) % 9, ph = e / 72;
    const int a = ph >> 1, b = ph & 1, ty = tap / 3, tx = tap % 3;
    float acc = 0.f;
    for (int dy = 0; dy < 3; dy++)
      for (int dx = 0; dx < 3; dx++)
        acc += p.w[p.src_k4 + (dy * 3 + dx) * 8 + ci] * (up_coef(a, dy, ty) * up_coef(b, dx, tx));
    p.prep[p.dst_w4eff + e] = acc;
    p.prep[p.dst_w4eff_c + (ci * 4 + ph) * 9 + tap] = acc;
  }
  for (int e = tid; e < 72; e += nthr) p.prep[p.dst_w4raw + e] = p.w[p.src_k4 + e];
  // frame pixels of the heat map: the conv taps of the row (column) outside the image, in phase form along the
  // line: pixel 2j + b of the line gets sum_o E[b][o] L[j + o - 1] of the low-res frame row (column) L
  for (int e = tid; e < 2 * 2 * 2 * 3 * 8; e += nthr) {
    const int ci = e % 8, o = (e / 8) % 3, b = (e / 24) % 2, side = (e / 48) % 2, isv = e / 96;
    float acc = 0.f;
    for (int d = 0; d < 3; d++) {
      const int tap = isv ? d * 3 + (side ? 2 : 0) : (side ? 2 : 0) * 3 + d;
      acc += p.w[p.src_k4 + tap * 8 + ci] * up_coef(b, d, o);
    }
    p.prep[p.dst_efr + e] = acc;
  }
  // background chain of the trunk: an all-empty window (input 0) gives relu(b1') after layer 1, a window of that
  // constant gives a constant after layer 2, ... -- same fma order (ci outer, tap inner) as the conv kernels, so
  // the skipped waves write bit-identical values
  // upconv3 (layer index 6): phase weights from the BN-folded kernel (folded in place, same thread order
  // would race with the fold above: recompute the fold here)
  {
    const int cout = 8, cin = 4;
    const float *g = p.w + p.src_g[6];
    for (int e = tid; e < 4 * 9 * cin * cout; e += nthr) {
      const int co = e % cout, ci = (e / cout) % cin, tap = (e / (cout * cin)) % 9, ph = e / (cout * cin * 9);
      const int a = ph >> 1, b = ph & 1, ty = tap / 3, tx = tap % 3;
      const float inv = g[co] / sqrtf(g[3 * cout + co] + 1e-3f);
      float acc = 0.f;
      for (int dy = 0; dy < 3; dy++)
        for (int dx = 0; dx < 3; dx++)
          acc += (p.w[p.src_k[6] + ((dy * 3 + dx) * cin + ci) * cout + co] * inv) * (up_coef(a, dy, ty) * up_coef(b, dx, tx));
      p.prep[p.dst_w3mf + ((co >> 2) * 36 + tap * 4 + ci) * 16 + ph * 4 + (co & 3)] = acc;
    }
  }
  // conv1 (layer 0) reads two BINARY maps: the response of a 3x3 window of one map is one of 512 values per output
  // channel.  Same fold as above (recomputed: the in-place fold may still be running in other threads); bit
  // (dy*3 + dx) of the pattern <-> tap (dy, dx), taps summed in tap order.
  {
    const float *g = p.w + p.src_g[0];
    for (int e = tid; e < 2 * 512 * 8; e += nthr) {
      const int co = e & 7, pat = (e >> 3) & 511, ci = e >> 12;
      const float inv = g[co] / sqrtf(g[3 * 8 + co] + 1e-3f);
      // the folded bias rides in the table of channel 0: out = LUT[0][pattern0] + LUT[1][pattern1]
      float acc = ci == 0 ? p.w[p.src_b[0] + co] * inv + (g[8 + co] - g[2 * 8 + co] * inv) : 0.f;
      for (int tap = 0; tap < 9; tap++)
        if ((pat >> tap) & 1) acc += p.w[p.src_k[0] + (tap * 2 + ci) * 8 + co] * inv;
      p.prep[p.dst_lut1 + e] = acc;
    }
  }
  // upconv2 (layer index 5): the same for the 2 -> 4 layer, K padded from 18 to 20 with zero rows
  {
    const int cout = 4, cin = 2;
    const float *g = p.w + p.src_g[5];
    for (int e = tid; e < 20 * 16; e += nthr) {
      const int n = e % 16, k = e / 16, co = n & 3, ph = n >> 2, ci = k & 1, tap = k >> 1;
      float acc = 0.f;
      if (k < 18) {
        const int a = ph >> 1, b = ph & 1, ty = tap / 3, tx = tap % 3;
        const float inv = g[co] / sqrtf(g[3 * cout + co] + 1e-3f);
        for (int dy = 0; dy < 3; dy++)
          for (int dx = 0; dx < 3; dx++)
            acc += (p.w[p.src_k[5] + ((dy * 3 + dx) * cin + ci) * cout + co] * inv) * (up_coef(a, dy, ty) * up_coef(b, dx, tx));
      }
      p.prep[p.dst_w2mf + e] = acc;
    }
  }
  // frame-line variants of the phase weights (k_head_frames): the conv taps that fall into the zero padding of the
  // variant's frame line are left out.  v = 0 top (row phase 0 loses dy = 0), 1 bottom (row phase 1 loses dy = 2),
  // 2 left (column phase 0 loses dx = 0), 3 right (column phase 1 loses dx = 2)
  {
    const float *g2 = p.w + p.src_g[5], *g3 = p.w + p.src_g[6];
    for (int e = tid; e < 4 * 20 * 16; e += nthr) {
      const int n = e % 16, k = (e / 16) % 20, v = e / 320, co = n & 3, ph = n >> 2, ci = k & 1, tap = k >> 1;
      float acc = 0.f;
      if (k < 18) {
        const int a = ph >> 1, b = ph & 1, ty = tap / 3, tx = tap % 3;
        const float inv = g2[co] / sqrtf(g2[3 * 4 + co] + 1e-3f);
        for (int dy = 0; dy < 3; dy++)
          for (int dx = 0; dx < 3; dx++) {
            const bool drop = (v == 0 && a == 0 && dy == 0) || (v == 1 && a == 1 && dy == 2) || (v == 2 && b == 0 && dx == 0) ||
                              (v == 3 && b == 1 && dx == 2);
            if (!drop) acc += (p.w[p.src_k[5] + ((dy * 3 + dx) * 2 + ci) * 4 + co] * inv) * (up_coef(a, dy, ty) * up_coef(b, dx, tx));
          }
      }
      p.prep[p.dst_w2fr + e] = acc;
    }
    for (int e = tid; e < 4 * 36 * 16; e += nthr) {
      const int n = e % 16, k = (e / 16) % 36, v = e / 576, co = n & 7, q = n >> 3, ci = k & 3, tap = k >> 2;
      const int a = v == 0 ? 0 : v == 1 ? 1 : q, b = v == 2 ? 0 : v == 3 ? 1 : q, ty = tap / 3, tx = tap % 3;
      const float inv = g3[co] / sqrtf(g3[3 * 8 + co] + 1e-3f);
      float acc = 0.f;
      for (int dy = 0; dy < 3; dy++)
        for (int dx = 0; dx < 3; dx++) {
          const bool drop = (v == 0 && dy == 0) || (v == 1 && dy == 2) || (v == 2 && dx == 0) || (v == 3 && dx == 2);
          if (!drop) acc += (p.w[p.src_k[6] + ((dy * 3 + dx) * 4 + ci) * 8 + co] * inv) * (up_coef(a, dy, ty) * up_coef(b, dx, tx));
        }
      p.prep[p.dst_w3fr + e] = acc;
    }
  }
  if (tid == 0) p.prep[p.dst_b4] = p.w[p.src_b4];
  } else {
  // k_convm's B operand for the 8 -> 8 layers, exactly as lane (n = (co, r), kq) of MFMA step j wants it:
  // B[k = 4 j + kq][(co, r)] = w[row - r][dx][ci][co] for k = (row * 3 + dx) * 8 + ci inside the 3-row window, else 0
  for (int l = 1; l < 4; l++)
    for (int e = tid; e < 24 * 64; e += nthr) {
      const int lane = e & 63, j = e >> 6, n16 = lane & 15, kq = lane >> 4, co = n16 >> 1, r = n16 & 1;
      const int k = 4 * j + kq, rd = k >> 3, ci = k & 7, row = rd / 3, dx = rd - row * 3, tr = row - r;
      p.prep[p.dst_wbm[l - 1] + e] = (tr >= 0 && tr < 3) ? p.prep[p.dst_w[l] + ((tr * 3 + dx) * 8 + ci) * 8 + co] : 0.f;
    }
  if (blockIdx.x == 0 && ltid < 8) {
    const int tid = ltid;
    float bgv[8];
    for (int ci = 0; ci < 8; ci++) bgv[ci] = 0.f;
    for (int l = 0; l < 4; l++) {
      const int cin = p.cin[l];
      float acc = 0.f;
      for (int ci = 0; ci < cin; ci++)
        for (int tap = 0; tap < 9; tap++) acc = __builtin_fmaf(bgv[ci], p.prep[p.dst_w[l] + (tap * cin + ci) * 8 + tid], acc);
      const float o = fmaxf(acc + p.prep[p.dst_b[l] + tid], 0.f);
      p.prep[p.dst_bg[l] + tid] = o;
      for (int ci = 0; ci < 8; ci++) bgv[ci] = __shfl(o, ci, 8);
    }
  }
  }
}

// ---- generic direct 3x3 convolution ------------------------------------------------
struct ConvParams {
  const float *in;                 // MODE 0 / 2: planar [img][CIN][Hin][Win]
  const unsigned *bits[2];         // MODE 1: word bits[ci][img * bits_stride + w], LSB-first (ch0 ship, ch1 laser)
  size_t bits_stride;              // words between consecutive images (PS*PS/32, or twice that for interleaved maps)
  const float *w, *b;              // folded [9][CIN][COUT], [COUT]
  float *out;                      // planar [img][COUT][Ho][Wo] or HWC [img][Ho][Wo][COUT]
  const uint8_t *mask;             // per image, may be null
  const float *wbm;                // k_convm, CIN = 8: per-lane B operand [24][64] (PrepLayout::wbm)
  int ablate;                      // diagnostics (OFX_CONV_ABLATE): 1 no global loads, 2 no FMAs, 4 no stores
  const float *bg_in, *bg_out;     // background value per input / output channel (null = no background skip)
  int H, W;                        // conv domain (input after any upsampling) = conv output size
  int tiles_x, tiles;              // tiles per row / per image
};

// MODE: 0 planar f32 input, 1 two 1-bit maps, 2 planar f32 input upsampled x2 (bilinear, half-pixel)
template <int CIN, int COUT, int TH, int TW, int MODE, bool POOL, bool OUT_HWC>
__global__ __launch_bounds__(((TH / 2) * (TW / 2) + 63) / 64 * 64) void k_conv(ConvParams p) {
  constexpr int NT = (TH / 2) * (TW / 2);
  constexpr int NTB = (NT + 63) / 64 * 64;
  constexpr int TWP = TW + 2;
  __shared__ __align__(16) float tile[CIN][TH + 2][TWP];
  const int img = blockIdx.x / p.tiles, t = blockIdx.x - img * p.tiles;
  if (p.mask && !p.mask[img]) return;  // block-uniform
  const int ty0 = (t / p.tiles_x) * TH, tx0 = (t % p.tiles_x) * TW;
  const int tid = threadIdx.x;
  const int H = p.H, W = p.W;

  // ---- stage the (TH+2) x (TW+2) x CIN input patch (zero outside the image: padding 'same').
  // Loads are issued in batches of SU before any LDS store so one memory latency covers SU elements.
  constexpr int TOTAL = CIN * (TH + 2) * TWP;
  constexpr int SU = (MODE == 2) ? 4 : 8;
  for (int base = 0; base < TOTAL; base += NTB * SU) {
    float vals[SU];
#pragma unroll
    for (int u = 0; u < SU; u++) {
      const int e = base + u * NTB + tid;
      float v = 0.f;
      if (e < TOTAL) {
        const int c = e % TWP, r = (e / TWP) % (TH + 2), ci = e / (TWP * (TH + 2));
        const int gy = ty0 - 1 + r, gx = tx0 - 1 + c;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
          if (MODE == 0) {
            v = (OFX_ABL(p) & 1) ? 1.f : p.in[(((size_t)img * CIN + ci) * H + gy) * W + gx];
          } else if (MODE == 1) {
            const int cell = gy * W + gx;
            v = (float)((p.bits[ci][(size_t)img * p.bits_stride + (cell >> 5)] >> (cell & 31)) & 1u);
          } else {
            const int Hs = H >> 1, Ws = W >> 1;
            const float sy = ((float)gy + 0.5f) * 0.5f - 0.5f, sx = ((float)gx + 0.5f) * 0.5f - 0.5f;
            const float fy = floorf(sy), fx = floorf(sx);
            const float ly = sy - fy, lx = sx - fx;
            int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
            y0 = max(y0, 0); x0 = max(x0, 0); y1 = min(y1, Hs - 1); x1 = min(x1, Ws - 1);
            const float *sp = p.in + ((size_t)img * CIN + ci) * Hs * Ws;
            const float a = sp[y0 * Ws + x0], b = sp[y0 * Ws + x1], d = sp[y1 * Ws + x0], g = sp[y1 * Ws + x1];
            const float top = a + (b - a) * lx, bot = d + (g - d) * lx;
            v = top + (bot - top) * ly;
          }
        }
      }
      vals[u] = v;
    }
#pragma unroll
    for (int u = 0; u < SU; u++) {
      const int e = base + u * NTB + tid;
      if (e < TOTAL) (&tile[0][0][0])[e] = vals[u];
    }
  }
  __syncthreads();
  if (tid >= NT) return;
  const int tr = tid / (TW / 2), tc = tid - tr * (TW / 2);

  float acc[2][2][COUT];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int co = 0; co < COUT; co++) acc[i][j][co] = 0.f;

  // Background skip (wave-uniform): when every window of this wave lies inside the image and shows only the
  // background value of each input channel, the outputs are the precomputed background response.
  bool skip = false;
  if (p.bg_in != nullptr) {
    bool flat = ty0 + 2 * tr >= 1 && ty0 + 2 * tr + 2 < H && tx0 + 2 * tc >= 1 && tx0 + 2 * tc + 2 < W;
    for (int ci = 0; ci < CIN && flat; ci++) {
      const float bgv = p.bg_in[ci];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float2 lo = *reinterpret_cast<const float2 *>(&tile[ci][2 * tr + r][2 * tc]);
        const float2 hi = *reinterpret_cast<const float2 *>(&tile[ci][2 * tr + r][2 * tc + 2]);
        flat = flat && lo.x == bgv && lo.y == bgv && hi.x == bgv && hi.y == bgv;
      }
    }
    skip = __all(flat);
  }
  if (!skip && !(OFX_ABL(p) & 2)) {
#pragma unroll
  for (int ci = 0; ci < CIN; ci++) {
    float v[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const float2 lo = *reinterpret_cast<const float2 *>(&tile[ci][2 * tr + r][2 * tc]);
      const float2 hi = *reinterpret_cast<const float2 *>(&tile[ci][2 * tr + r][2 * tc + 2]);
      v[r][0] = lo.x; v[r][1] = lo.y; v[r][2] = hi.x; v[r][3] = hi.y;
    }
#pragma unroll
    for (int dy = 0; dy < 3; dy++)
#pragma unroll
      for (int dx = 0; dx < 3; dx++)
#pragma unroll
        for (int co = 0; co < COUT; co++) {
          const float wv = p.w[((dy * 3 + dx) * CIN + ci) * COUT + co];  // wave-uniform -> scalar load
#pragma unroll
          for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) acc[i][j][co] = __builtin_fmaf(v[i + dy][j + dx], wv, acc[i][j][co]);
        }
  }
  }

  // ---- epilogue: folded bias, ReLU, optional 2x2 max-pool ----
  const int oy = ty0 + 2 * tr, ox = tx0 + 2 * tc;
#pragma unroll
  for (int co = 0; co < COUT; co++) {
    const float bias = p.b[co];
    float o00 = fmaxf(acc[0][0][co] + bias, 0.f), o01 = fmaxf(acc[0][1][co] + bias, 0.f);
    float o10 = fmaxf(acc[1][0][co] + bias, 0.f), o11 = fmaxf(acc[1][1][co] + bias, 0.f);
    if (skip) o00 = o01 = o10 = o11 = p.bg_out[co];
    if ((OFX_ABL(p) & 4) && o00 != 12345.f) continue;
    if (POOL) {
      const float m = fmaxf(fmaxf(o00, o01), fmaxf(o10, o11));
      const int Ho = H >> 1, Wo = W >> 1, py = oy >> 1, px = ox >> 1;
      if (OUT_HWC) p.out[(((size_t)img * Ho + py) * Wo + px) * COUT + co] = m;
      else p.out[(((size_t)img * COUT + co) * Ho + py) * Wo + px] = m;
    } else {
      float *o = p.out + (((size_t)img * COUT + co) * H + oy) * W + ox;
      *reinterpret_cast<float2 *>(o) = make_float2(o00, o01);
      *reinterpret_cast<float2 *>(o + W) = make_float2(o10, o11);
    }
  }
}

// ---- direct 3x3 convolution, wide register tile, VGPR-resident weights ---------------------------------
// Same contract as k_conv for the COUT = 8 layers, tuned to what tools/ubench_fma.hip measured on gfx950:
// v_fmac_f32 with an SGPR weight operand issues at HALF rate (75 TFLOP/s), with VGPR operands at 115.  Here each
// thread owns 2 x 4 output pixels x 8 channels (64 accumulators), and the 72 weights of one input channel are
// fetched from an LDS copy with broadcast ds_read_b128 into VGPRs, each feeding 8 FMAs.
template <int CIN, int TH, int TW, int MODE, bool POOL, bool OUT_HWC>
__global__ __launch_bounds__(((TH / 2) * (TW / 4) + 63) / 64 * 64) void k_conv8(ConvParams p) {
  constexpr int COUT = 8;
  constexpr int NT = (TH / 2) * (TW / 4);
  constexpr int NTB = (NT + 63) / 64 * 64;
  constexpr int TWP = TW + 4;  // halo 1 each side + 2 pad: row stride multiple of 4 floats (16-byte aligned reads)
  __shared__ __align__(16) float tile[CIN][TH + 2][TWP];
  __shared__ __align__(16) float wl[CIN][9 * COUT];  // [ci][tap][co]
  const int img = blockIdx.x / p.tiles, t = blockIdx.x - img * p.tiles;
  if (p.mask && !p.mask[img]) return;
  const int ty0 = (t / p.tiles_x) * TH, tx0 = (t % p.tiles_x) * TW;
  const int tid = threadIdx.x;
  const int H = p.H, W = p.W;

  for (int e = tid; e < CIN * 9 * COUT; e += NTB) {  // folded weights [tap][ci][co] -> [ci][tap][co]
    const int co = e % COUT, tap = (e / COUT) % 9, ci = e / (9 * COUT);
    wl[ci][tap * COUT + co] = p.w[(tap * CIN + ci) * COUT + co];
  }
  // stage the (TH+2) x (TW+2) x CIN input patch: tile column c <-> image column tx0 - 1 + c
  constexpr int TOTAL = CIN * (TH + 2) * TWP;
  if (MODE == 1) {
    // 1-bit input: one thread per 32-bit WORD of a tile row (a row of TW+2 cells touches at most WPR words) instead
    // of one global load per cell: 2 x (TH+2) x WPR word loads, each expanded into up to 32 LDS floats.
    // (stage ablation: per-cell loads were 1.8 of the kernel's 5.0 ms)
    constexpr int WPR = (TW + 2 + 31) / 32 + 1;
    for (int e = tid; e < CIN * (TH + 2) * WPR; e += NTB) {
      const int k = e % WPR, r = (e / WPR) % (TH + 2), ci = e / (WPR * (TH + 2));
      const int gy = ty0 - 1 + r;
      float *trow = &tile[ci][r][0];
      if (gy < 0 || gy >= H) {
        for (int c = k; c < TWP; c += WPR) trow[c] = 0.f;  // padding row: zeros (strided split of the row)
        continue;
      }
      const int cell0 = gy * W + tx0 - 1;            // cell index of tile column 0 (may be -1 at the left image edge)
      const int w0 = (cell0 >= 0 ? cell0 : 0) >> 5;  // first word of the row segment
      const int wi = w0 + k;
      const unsigned word = (wi < (PS * PS) >> 5) ? p.bits[ci][(size_t)img * p.bits_stride + wi] : 0u;
      // cells of this word: wi*32 .. wi*32+31 -> tile columns c = cell - cell0
      const int cbeg = max(wi * 32 - cell0, 0), cend = min(wi * 32 + 32 - cell0, TWP);
      if (k == 0)
        for (int c = 0; c < cbeg; c++) trow[c] = 0.f;  // cell0 = -1 at the left image edge: column 0 is padding
      for (int c = cbeg; c < cend; c++) {
        const int gx = tx0 - 1 + c;
        const unsigned bit = (word >> ((cell0 + c) & 31)) & 1u;
        trow[c] = (c < TW + 2 && gx >= 0 && gx < W && bit) ? 1.f : 0.f;
      }
    }
  } else {
  constexpr int SU = 8;
  for (int base = 0; base < TOTAL; base += NTB * SU) {
    float vals[SU];
#pragma unroll
    for (int u = 0; u < SU; u++) {
      const int e = base + u * NTB + tid;
      float v = 0.f;
      if (e < TOTAL) {
        const int c = e % TWP, r = (e / TWP) % (TH + 2), ci = e / (TWP * (TH + 2));
        const int gy = ty0 - 1 + r, gx = tx0 - 1 + c;
        if (c < TW + 2 && gy >= 0 && gy < H && gx >= 0 && gx < W)
          v = p.in[(((size_t)img * CIN + ci) * H + gy) * W + gx];
      }
      vals[u] = v;
    }
#pragma unroll
    for (int u = 0; u < SU; u++) {
      const int e = base + u * NTB + tid;
      if (e < TOTAL) (&tile[0][0][0])[e] = vals[u];
    }
  }
  }
  __syncthreads();
  if (tid >= NT) return;
  const int tr = tid / (TW / 4), tc = tid - tr * (TW / 4);

  float acc[2][4][COUT];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int co = 0; co < COUT; co++) acc[i][j][co] = 0.f;

  bool skip = false;  // background skip, see k_conv
  if (p.bg_in != nullptr) {
    bool flat = ty0 + 2 * tr >= 1 && ty0 + 2 * tr + 2 < H && tx0 + 4 * tc >= 1 && tx0 + 4 * tc + 4 < W;
    for (int ci = 0; ci < CIN && flat; ci++) {
      const float bgv = p.bg_in[ci];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float4 lo = *reinterpret_cast<const float4 *>(&tile[ci][2 * tr + r][4 * tc]);
        const float4 hi = *reinterpret_cast<const float4 *>(&tile[ci][2 * tr + r][4 * tc + 4]);
        flat = flat && lo.x == bgv && lo.y == bgv && lo.z == bgv && lo.w == bgv && hi.x == bgv && hi.y == bgv;
      }
    }
    skip = __all(flat);
  }
#pragma unroll 1
  for (int ci = 0; ci < ((skip || (OFX_ABL(p) & 2)) ? 0 : CIN); ci++) {
    float v[4][8];
#pragma unroll
    for (int r = 0; r < 4; r++) {  // cols 4tc .. 4tc+5 of the tile: two aligned b128 reads (conflict free)
      const float4 lo = *reinterpret_cast<const float4 *>(&tile[ci][2 * tr + r][4 * tc]);
      const float4 hi = *reinterpret_cast<const float4 *>(&tile[ci][2 * tr + r][4 * tc + 4]);
      v[r][0] = lo.x; v[r][1] = lo.y; v[r][2] = lo.z; v[r][3] = lo.w; v[r][4] = hi.x; v[r][5] = hi.y;
    }
#pragma unroll
    for (int dy = 0; dy < 3; dy++) {
      float wv[3 * COUT];  // one tap row: 24 weights, broadcast reads
#pragma unroll
      for (int q = 0; q < 6; q++) {
        const float4 t4 = *reinterpret_cast<const float4 *>(&wl[ci][dy * 3 * COUT + 4 * q]);
        wv[4 * q] = t4.x; wv[4 * q + 1] = t4.y; wv[4 * q + 2] = t4.z; wv[4 * q + 3] = t4.w;
      }
#pragma unroll
      for (int dx = 0; dx < 3; dx++)
#pragma unroll
        for (int co = 0; co < COUT; co++)
#pragma unroll
          for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
              acc[i][j][co] = __builtin_fmaf(v[i + dy][j + dx], wv[dx * COUT + co], acc[i][j][co]);
    }
  }

  const int oy = ty0 + 2 * tr, ox = tx0 + 4 * tc;
#pragma unroll
  for (int co = 0; co < COUT; co++) {
    const float bias = p.b[co];
    float o[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) o[i][j] = skip ? p.bg_out[co] : fmaxf(acc[i][j][co] + bias, 0.f);
    if ((OFX_ABL(p) & 4) && o[0][0] != 12345.f) continue;
    if (POOL) {
      const float m0 = fmaxf(fmaxf(o[0][0], o[0][1]), fmaxf(o[1][0], o[1][1]));
      const float m1 = fmaxf(fmaxf(o[0][2], o[0][3]), fmaxf(o[1][2], o[1][3]));
      const int Ho = H >> 1, Wo = W >> 1, py = oy >> 1, px = ox >> 1;
      if (OUT_HWC) {
        p.out[(((size_t)img * Ho + py) * Wo + px) * COUT + co] = m0;
        p.out[(((size_t)img * Ho + py) * Wo + px + 1) * COUT + co] = m1;
      } else {
        *reinterpret_cast<float2 *>(&p.out[(((size_t)img * COUT + co) * Ho + py) * Wo + px]) = make_float2(m0, m1);
      }
    } else {
      float *op = p.out + (((size_t)img * COUT + co) * H + oy) * W + ox;
      *reinterpret_cast<float4 *>(op) = make_float4(o[0][0], o[0][1], o[0][2], o[0][3]);
      *reinterpret_cast<float4 *>(op + W) = make_float4(o[1][0], o[1][1], o[1][2], o[1][3]);
    }
  }
}

// max without the canonicalising v_max(x, x) the compiler puts in front of fmaxf() in IEEE mode (x is an MFMA
// result here, never a signalling NaN): med3(x, floor, +inf).  NOT inline assembly: the compiler's hazard recogniser
// does not look inside an asm statement, so an asm v_max placed right behind the MFMA that produces x reads the
// register before the matrix pipe has written it (seen as wrong cells under one scheduling variant).
// The +inf comes out of an opaque scalar move: with a literal the optimiser folds med3 back into the canonicalising max.
__device__ __forceinline__ float max_raw(float x, float floor) {
  float pinf;
  asm("s_mov_b32 %0, 0x7f800000" : "=s"(pinf));
  return __builtin_amdgcn_fmed3f(x, floor, pinf);
}

// ---- trunk convolution on the matrix cores ---------------------------------------------------------------------
// conv3x3 (zero padding) + folded BN + ReLU + 2x2 max-pool as a GEMM whose N dimension is 8 output channels x 2
// adjacent output rows:  D[pixel x][(co, r)] = sum_k A[x][k] B[k][(co, r)],  k = (input row 0..3, dx, ci),
// B[k][(co, r)] = w[row - r][dx][ci][co] when 0 <= row - r <= 2, else 0  -> K = 12 CIN, 3/4 of the MACs useful, but a
// v_mfma_f32_16x16x4_f32 retires 32 MAC/cycle against 16 for v_fmac_f32 (both share the SIMD's issue slots on gfx950,
// tools/ubench_mix.hip), and the whole epilogue of an M-tile (2x2 pool, ReLU, store) is ~10 VALU instructions.
// A[x][k] is gathered from an LDS copy of the input tile (one ds_read_b32 per lane per MFMA, immediate offsets);
// the row pair of a column group shares one accumulator quad: lane (n = (co, r), kq) holds pixels 4 kq .. 4 kq + 3,
// so the x-pool is in-lane and the y-pool is one DPP quad swap.
// MODE 0: planar f32 input [img][CIN][H][W]; MODE 1: two 1-bit maps (CIN = 2).  Output: planar [img][8][H/2][W/2] or
// (OUT_HWC) [img][H/2][W/2][8].  TH rows x 16 NG columns per workgroup, TH even, H % TH == 0; W is masked.
// A workgroup walks TPW consecutive tiles of one image: the weights are fetched once, the global loads of tile i+1 are
// in flight (in registers) while tile i computes, and the grid stays small (the dispatcher needs ~5 ns per workgroup:
// one workgroup per tile cost 1.6 ms of launch floor for conv2 alone).
template <int CIN, int TH, int NG, int MODE, bool OUT_HWC, int TPW>
__global__ __launch_bounds__(256) void k_convm(ConvParams p) {
  constexpr int TW = 16 * NG, LS = TW + 8;  // LDS row: image column tx0 + c sits at index c + 4 (16-byte aligned interior),
                                            // the left / right halo columns at 3 and TW + 4
  constexpr int PLS = ((TH + 2) * LS + 63) / 64 * 64 + 16;  // plane stride = 16 mod 64: the 4 k-quarters hit different banks
  constexpr int NK = 3 * CIN;                                // MFMAs per M-tile (K = 12 CIN)
  constexpr int JOBS = NG * (TH / 2);
  constexpr int ROWS = CIN * (TH + 2);
  constexpr bool VEC = MODE == 0 && !OUT_HWC;                // rows of W floats are 16-byte aligned (W % 4 == 0)
  __shared__ __align__(16) float tile[CIN * PLS];
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (each XCD has its own L2), so the eight XCDs take contiguous
  // eighths of the tile list and neighbouring row tiles -- which share their halo rows -- meet in one L2
  unsigned bid = blockIdx.x;
#if OFX_XCD_SWIZZLE
  if ((gridDim.x & 7u) == 0) bid = (bid & 7u) * (gridDim.x >> 3) + (bid >> 3);
#endif
  const int first = (int)bid * TPW;                          // p.tiles % TPW == 0: all tiles of a workgroup share the image
  const int img = first / p.tiles, t_first = first - img * p.tiles;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, W = p.W;
  const int n16 = lane & 15, kq = lane >> 4, co = n16 >> 1, r = n16 & 1;

  // B operand: the lane's column (co, r) of the banded weight matrix, rows k = 4 j + kq
  float bw[NK];
  int aoff[NK];  // LDS offset of A[.][4 j + kq] relative to the M-tile origin (compile-time + kq * PLS when CIN == 8)
#pragma unroll
  for (int j = 0; j < NK; j++) {
    const int k = 4 * j + kq, rd = k / CIN, ci = k - rd * CIN, row = rd / 3, dx = rd - row * 3, tr = row - r;
    if constexpr (CIN == 8) bw[j] = p.wbm[j * 64 + lane];  // pre-arranged by k_policy_prepare: one coalesced load
    else bw[j] = (tr >= 0 && tr < 3) ? p.w[((tr * 3 + dx) * CIN + ci) * 8 + co] : 0.f;
    aoff[j] = ci * PLS + row * LS + dx;
  }
  const float bias = p.b[co];
  const f32x4 binit = {bias, bias, bias, bias};
  const int H2 = H >> 1, W2 = W >> 1;

  // ---- staging, split into fetch (global -> registers) and commit (registers -> LDS) ----
  constexpr int WPR = (TW + 2 + 31) / 32 + 1;                      // MODE 1: words a tile row can touch
  constexpr int NWI = MODE == 1 ? (ROWS * WPR + 255) / 256 : 1;    // word items per thread
  constexpr int V4 = TW / 4, RPW = VEC ? (ROWS + 3) / 4 : 1;       // VEC: float4 per row, rows per wave
  static_assert(!VEC || V4 + 2 <= 64, "tile row wider than one wave");
  unsigned wpre[NWI];
  f32x4 vpre[RPW];  // native vectors: a float4 select is lowered to a pointer select + flat loads
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

  auto fetch = [&](int t) {
    const int ty0 = (t / p.tiles_x) * TH, tx0 = (t % p.tiles_x) * TW;
    if constexpr (MODE == 1) {
#pragma unroll
      for (int u = 0; u < NWI; u++) {
        const int e = tid + 256 * u;
        const int k = e % WPR, rr = (e / WPR) % (TH + 2), ci = e / (WPR * (TH + 2));
        const int gy = ty0 - 1 + rr;
        unsigned word = 0u;
        if (e < ROWS * WPR && gy >= 0 && gy < H) {
          const int cell0 = gy * W + tx0 - 1;
          const int wi = ((cell0 >= 0 ? cell0 : 0) >> 5) + k;
          if (wi < (PS * PS) >> 5) word = p.bits[ci][(size_t)img * p.bits_stride + wi];
        }
        wpre[u] = word;
      }
    } else if constexpr (VEC) {
      // a wave moves one tile row per step: lane i < V4 the i-th float4 of the interior, lanes V4 / V4+1 the float4
      // that holds the left / right halo column
      const int gxl = lane < V4 ? tx0 + 4 * lane : (lane == V4 ? tx0 - 4 : tx0 + TW);
      const bool colok = lane < V4 + 2 && gxl >= 0 && gxl < W && !(OFX_ABL(p) & 1);
      const float *imgbase = p.in + (size_t)img * CIN * H * W;
#pragma unroll
      for (int u = 0; u < RPW; u++) {
        const int rr = wv + 4 * u;
        const int ci = rr / (TH + 2), r_ = rr - ci * (TH + 2), gy = ty0 - 1 + r_;
        const bool rowok = rr < ROWS && gy >= 0 && gy < H;  // wave-uniform
        // 32-bit offset from the image's base (scalar base + vector offset addressing; an image is < 4 GB)
        const unsigned off = (unsigned)((rowok ? ci * H + gy : 0) * W + gxl);
        vpre[u] = z4;
        if (rowok && colok) vpre[u] = *reinterpret_cast<const f32x4 *>(imgbase + off);  // exec-masked global load
      }
    }
  };

  auto commit = [&](int t) {
    const int ty0 = (t / p.tiles_x) * TH, tx0 = (t % p.tiles_x) * TW;
    if constexpr (MODE == 1) {
#pragma unroll
      for (int u = 0; u < NWI; u++) {
        const int e = tid + 256 * u;
        if (e >= ROWS * WPR) break;
        const int k = e % WPR, rr = (e / WPR) % (TH + 2), ci = e / (WPR * (TH + 2));
        const int gy = ty0 - 1 + rr;
        float *trow = &tile[ci * PLS + rr * LS + 3];  // trow[c] <-> image column tx0 - 1 + c
        if (gy < 0 || gy >= H) {
          for (int c = k; c < TW + 2; c += WPR) trow[c] = 0.f;
          continue;
        }
        const int cell0 = gy * W + tx0 - 1;            // cell index of column c = 0 (-1 at the left image edge)
        const int wi = ((cell0 >= 0 ? cell0 : 0) >> 5) + k;
        const int cbeg = max(wi * 32 - cell0, 0), cend = min(wi * 32 + 32 - cell0, TW + 2);
        if (k == 0)
          for (int c = 0; c < cbeg; c++) trow[c] = 0.f;
        const unsigned sh = wpre[u] >> ((cell0 + cbeg) & 31);
        for (int c = cbeg; c < cend; c++) {
          const int gx = tx0 - 1 + c;
          trow[c] = (gx >= 0 && gx < W && ((sh >> (c - cbeg)) & 1u)) ? 1.f : 0.f;
        }
      }
    } else if constexpr (VEC) {
#pragma unroll
      for (int u = 0; u < RPW; u++) {
        const int rr = wv + 4 * u;
        if (rr >= ROWS) break;  // wave-uniform
        const int ci = rr / (TH + 2), r_ = rr - ci * (TH + 2);
        float *trow = &tile[ci * PLS + r_ * LS];
        if (lane < V4) *reinterpret_cast<f32x4 *>(trow + 4 + 4 * lane) = vpre[u];
        else if (lane == V4) trow[3] = vpre[u][3];
        else if (lane == V4 + 1) trow[TW + 4] = vpre[u][0];
      }
    } else {  // unaligned rows (W % 4 != 0): element-wise, no prefetch (the 50x50 layer only)
      constexpr int RW = TW + 2, TOTAL = CIN * (TH + 2) * RW, SU = 8;
      for (int base = 0; base < TOTAL; base += 256 * SU) {
        float vals[SU];
#pragma unroll
        for (int u = 0; u < SU; u++) {
          const int e = base + u * 256 + tid;
          float v = 0.f;
          if (e < TOTAL) {
            const int c = e % RW, rr = (e / RW) % (TH + 2), ci = e / (RW * (TH + 2));
            const int gy = ty0 - 1 + rr, gx = tx0 - 1 + c;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = p.in[(((size_t)img * CIN + ci) * H + gy) * W + gx];
          }
          vals[u] = v;
        }
#pragma unroll
        for (int u = 0; u < SU; u++) {
          const int e = base + u * 256 + tid;
          if (e < TOTAL) {
            const int c = e % RW, rr = (e / RW) % (TH + 2), ci = e / (RW * (TH + 2));
            tile[ci * PLS + rr * LS + 3 + c] = vals[u];
          }
        }
      }
    }
  };

  // A[.][4 j + kq]: with 8 input channels the (row, dx) of step j is a compile-time constant and the channel is
  // 4 (j & 1) + kq, so every LDS read is base + immediate; with 2 channels the per-lane offsets live in registers
  const float *abase = &tile[(CIN == 8 ? kq * PLS : 0) + n16 + 3];
  auto aof = [&](int j) -> int {
    if (CIN == 8) return (4 * (j & 1)) * PLS + ((j >> 1) / 3) * LS + ((j >> 1) % 3);
    return aoff[j];
  };

  // One ds_read_b32 with a 16-bit immediate per operand: left alone the compiler pairs the reads into ds_read2_b32,
  // whose 8-bit offsets cost a v_add per pair -- and VALU instructions are what this kernel is short of (they share
  // the issue port with the MFMAs; the LDS port is idle).  volatile keeps the reads single.
  auto lda = [&](const float *a, int j) -> float {
#if OFX_CONVM_VOLA
    typedef const volatile __attribute__((address_space(3))) float lds_cvf;  // stay in the LDS address space
    return *(lds_cvf *)(a + aof(j));
#else
    return a[aof(j)];
#endif
  };
  fetch(t_first);
#pragma unroll 1
  for (int i = 0; i < TPW; i++) {
    const int t = t_first + i;
    const int ty0 = (t / p.tiles_x) * TH, tx0 = (t % p.tiles_x) * TW;
    commit(t);
    __syncthreads();
    if (i + 1 < TPW) fetch(t + 1);

    // epilogue of an M-tile: x-pool + ReLU (two v_med3), y-pool = max with the DPP quad swap [1,0,3,2] (rows r = 0 / 1
    // sit in lanes n, n ^ 1), one 8-byte store from the r = 0 lanes
    float *const obase = OUT_HWC ? p.out + (((size_t)img * H2 + (ty0 >> 1)) * W2 + (tx0 >> 1) + 2 * kq) * 8 + co
                                 : p.out + (((size_t)img * 8 + co) * H2 + (ty0 >> 1)) * W2 + (tx0 >> 1) + 2 * kq;
    auto finish = [&](const f32x4 d, int g, int tt) {
      float q0, q1;
      q0 = max_raw(max_raw(d[0], 0.f), d[1]);  // compiler-visible reads of the MFMA result (see max_raw)
      q1 = max_raw(max_raw(d[2], 0.f), d[3]);
      q0 = max_raw(q0, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, q0), 0xB1, 0xF, 0xF, true)));
      q1 = max_raw(q1, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, q1), 0xB1, 0xF, 0xF, true)));
      const int px = ((tx0 + 16 * g) >> 1) + 2 * kq;
      if (r == 0 && px < W2 && !(OFX_ABL(p) & 4)) {
        if (OUT_HWC) {
          float *op = obase + ((size_t)tt * W2 + 8 * g) * 8;
          op[0] = q0;
          if (px + 1 < W2) op[8] = q1;
        } else {  // W2 is even here: px < W2 implies px + 1 < W2
          *reinterpret_cast<float2 *>(obase + tt * W2 + 8 * g) = make_float2(q0, q1);
        }
      }
    };
    // two M-tiles per iteration (independent accumulator chains keep the matrix pipe busy); a trailing odd one alone
#pragma unroll 1
    for (int job = wv; job < JOBS; job += 8) {
      const int job1 = job + 4;
      const int g0 = job % NG, t0 = job / NG;
      const float *a0 = abase + (2 * t0) * LS + 16 * g0;
      if (job1 < JOBS) {  // wave-uniform
        const int g1 = job1 % NG, t1 = job1 / NG;
        const float *a1 = abase + (2 * t1) * LS + 16 * g1;
        f32x4 d0 = binit, d1 = binit;
        if (!(OFX_ABL(p) & 2)) {
#if OFX_CONVM_VOLA
          // volatile reads stay in program order: three batches, the reads of batch b + 2 behind the MFMAs of batch b
          constexpr int NB = NK / 3;
          static_assert(NK % 3 == 0, "three batches");
          float A0[3][NB], A1[3][NB];
          auto ldb = [&](int b) {
#pragma unroll
            for (int j = 0; j < NB; j++) { A0[b][j] = lda(a0, b * NB + j); A1[b][j] = lda(a1, b * NB + j); }
          };
          auto mmb = [&](int b) {
#pragma unroll
            for (int j = 0; j < NB; j++) {
              d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[b][j], bw[b * NB + j], d0, 0, 0, 0);
              d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[b][j], bw[b * NB + j], d1, 0, 0, 0);
            }
          };
          ldb(0); ldb(1);
          __builtin_amdgcn_sched_barrier(0);  // or the scheduler sinks every read next to its MFMA again
          mmb(0);
          ldb(2);
          __builtin_amdgcn_sched_barrier(0);
          mmb(1);
          mmb(2);
#else
#pragma unroll
          for (int j = 0; j < NK; j++) {
            d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(lda(a0, j), bw[j], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(lda(a1, j), bw[j], d1, 0, 0, 0);
          }
#endif
        }
        finish(d0, g0, t0);
        finish(d1, g1, t1);
      } else {
        f32x4 d0 = binit;
        if (!(OFX_ABL(p) & 2)) {
#pragma unroll
          for (int j = 0; j < NK; j++) d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(lda(a0, j), bw[j], d0, 0, 0, 0);
        }
        finish(d0, g0, t0);
      }
    }
    if (i + 1 < TPW) __syncthreads();  // the next commit overwrites the tile
  }
}

template <int CIN, int TH, int NG, int MODE, bool OUT_HWC, int TPW>
static int launch_convm(ofx_handle *h, ConvParams p, int images, int H) {
  p.H = H; p.W = H;
  p.tiles_x = (H + 16 * NG - 1) / (16 * NG);
  p.tiles = p.tiles_x * (H / TH);
  if (p.tiles % TPW) { ofx_set_error("launch_convm: %d tiles per image not divisible by %d", p.tiles, TPW); return OFX_ERR_INVALID; }
  hipLaunchKernelGGL((k_convm<CIN, TH, NG, MODE, OUT_HWC, TPW>), dim3((unsigned)(images * (p.tiles / TPW))), dim3(256), 0,
                     h->stream, p);
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}

// ---- conv1 on the binary observation maps as a table lookup ------------------------------------------------------
// The two input channels are 1-bit maps, so conv3x3 of one channel at one pixel takes one of 512 values per output
// channel: out[co] = b[co] + LUT[0][pattern0][co] + LUT[1][pattern1][co] (PrepLayout::lut1, 32 KB, staged in LDS).
// Every pixel is evaluated (no sparsity shortcut); a thread owns one pooled output pixel = 2x2 conv outputs = a
// 4x4 bit window per channel: 8 aligned word pairs + funnel shifts give the windows, 16 ds_read_b128 the table rows,
// then 2x2 max-pool + ReLU and 8 coalesced stores (planar [img][8][200][200]).  ~170 VALU instructions per pooled
// pixel against 6 MFMAs + epilogue per 16 in the GEMM form (3.4 ms): the kernel is bound by its 5.2 GB of output.
// Input rows are re-aligned while staging: bit x + 1 of LDS row r <-> image column x of row ty0 - 1 + r (bit 0 and
// the bits past column W-1 are the zero padding).
template <int TH>
__global__ __launch_bounds__(256) void k_conv1_lut(ConvParams p, const float *lut) {
  constexpr int W = PS, H = PS, WR = 14;                       // words per staged row (402 bits)
  constexpr int W2 = W / 2, NPX = (TH / 2) * W2;
  __shared__ __align__(16) float slut[2 * 512 * 8];
  __shared__ unsigned rows[2][TH + 2][WR];
  const int tiles = H / TH;
  const int img = blockIdx.x / tiles, ty0 = (blockIdx.x - img * tiles) * TH;
  const int tid = threadIdx.x;
  for (int e = tid; e < 2 * 512 * 8 / 4; e += 256)
    reinterpret_cast<float4 *>(slut)[e] = reinterpret_cast<const float4 *>(lut)[e];
  for (int e = tid; e < 2 * (TH + 2) * WR; e += 256) {
    const int w = e % WR, r = (e / WR) % (TH + 2), ci = e / (WR * (TH + 2));
    const int gy = ty0 - 1 + r;
    unsigned out = 0u;
    if (gy >= 0 && gy < H) {
      // output bits b = 0..31 <-> column x = 32 w - 1 + b <-> cell gy * W + x
      const long long s0 = (long long)gy * W + 32 * w - 1;        // cell of output bit 0 (-1 only for gy = 0, w = 0)
      const unsigned *bits = p.bits[ci] + (size_t)img * p.bits_stride;
      const long long sw = s0 >> 5;                               // arithmetic shift: -1 -> word -1
      const unsigned lo = (sw >= 0 && sw < (PS * PS) >> 5) ? bits[sw] : 0u;
      const unsigned hi = (sw + 1 < (PS * PS) >> 5) ? bits[sw + 1] : 0u;
      out = __funnelshift_r(lo, hi, (unsigned)(s0 & 31));
      // keep only columns 0 <= x < W of THIS row
      const int xlo = 32 * w - 1;
      if (xlo < 0) out &= ~1u;
      const int over = xlo + 32 - W;                              // bits past the last column
      if (over > 0) out = over >= 32 ? 0u : (out & (0xFFFFFFFFu >> over));
    }
    rows[ci][r][w] = out;
  }
  __syncthreads();
  float *const obase = p.out + ((size_t)img * 8 * (H / 2) + (ty0 >> 1)) * W2;  // wave-uniform: scalar base + 32-bit offsets
#if OFX_CONV1_V4
  // a thread owns FOUR horizontally adjacent pooled pixels: the kernel is bound by its 5.2 GB of output, and 16-byte
  // stores (1 KB contiguous per wave and channel plane) use the write path better than 4-byte ones (256 B)
  constexpr int NQ = NPX / 4, QR = W2 / 4;                       // quads of the tile, quads per pooled row
  for (int qd = tid; qd < NQ; qd += 256) {
    const int py = qd / QR, pq = qd - py * QR;
    const int x0 = 8 * pq;                                        // window = staged bits x0 .. x0 + 9 of rows 2 py .. 2 py + 3
    unsigned f[2][4];
#pragma unroll
    for (int ci = 0; ci < 2; ci++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const unsigned *rw = &rows[ci][2 * py + r][x0 >> 5];
        f[ci][r] = __funnelshift_r(rw[0], rw[1], (unsigned)(x0 & 31)) & 1023u;
      }
    f32x4 m4[8];                                                  // [channel] = the 4 pixels
#pragma unroll
    for (int j = 0; j < 4; j++) {
      f32x4 acc[4][2];                                            // [2x2 pixel][channels 0-3 | 4-7]
#pragma unroll
      for (int ci = 0; ci < 2; ci++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int dy = q >> 1, sh = 2 * j + (q & 1);
          const unsigned pat = ((f[ci][dy] >> sh) & 7u) | (((f[ci][dy + 1] >> sh) & 7u) << 3) | (((f[ci][dy + 2] >> sh) & 7u) << 6);
          const f32x4 *e = reinterpret_cast<const f32x4 *>(&slut[(ci * 512 + pat) * 8]);
          if (ci == 0) { acc[q][0] = e[0]; acc[q][1] = e[1]; }    // the table of channel 0 carries the bias
          else { acc[q][0] += e[0]; acc[q][1] += e[1]; }
        }
#pragma unroll
      for (int co = 0; co < 8; co++) {
        float m;  // the operands are ordinary VALU results (interlocked), not MFMA results: asm is safe here
        asm("v_max3_f32 %0, %1, %2, 0" : "=v"(m) : "v"(acc[0][co >> 2][co & 3]), "v"(acc[1][co >> 2][co & 3]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(acc[2][co >> 2][co & 3]), "v"(acc[3][co >> 2][co & 3]), "v"(m));
        m4[co][j] = m;
      }
    }
    const unsigned off = (unsigned)(py * W2 + 4 * pq);
    if (!(OFX_ABL(p) & 4)) {
#pragma unroll
      for (int co = 0; co < 8; co++) *reinterpret_cast<f32x4 *>(obase + (size_t)co * (H / 2) * W2 + off) = m4[co];
    }
  }
#else
  for (int px = tid; px < NPX; px += 256) {
    const int py = px / W2, pxx = px - py * W2;                   // pooled pixel of the tile
    const int x0 = 2 * pxx;                                       // window = staged bits x0 .. x0 + 3 of rows 2 py .. 2 py + 3
    f32x4 acc[4][2];                                              // [2x2 pixel][channels 0-3 | 4-7]
#pragma unroll
    for (int ci = 0; ci < 2; ci++) {
      unsigned f[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const unsigned *rw = &rows[ci][2 * py + r][x0 >> 5];
        f[r] = __funnelshift_r(rw[0], rw[1], (unsigned)(x0 & 31)) & 15u;
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int dy = q >> 1, dx = q & 1;
        const unsigned pat = ((f[dy] >> dx) & 7u) | (((f[dy + 1] >> dx) & 7u) << 3) | (((f[dy + 2] >> dx) & 7u) << 6);
        const f32x4 *e = reinterpret_cast<const f32x4 *>(&slut[(ci * 512 + pat) * 8]);
        if (ci == 0) { acc[q][0] = e[0]; acc[q][1] = e[1]; }      // the table of channel 0 carries the bias
        else { acc[q][0] += e[0]; acc[q][1] += e[1]; }
      }
    }
    const unsigned off = (unsigned)(py * W2 + pxx);
#pragma unroll
    for (int co = 0; co < 8; co++) {
      float m;  // the operands are ordinary VALU results (interlocked), not MFMA results: asm is safe here
      asm("v_max3_f32 %0, %1, %2, 0" : "=v"(m) : "v"(acc[0][co >> 2][co & 3]), "v"(acc[1][co >> 2][co & 3]));
      asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(acc[2][co >> 2][co & 3]), "v"(acc[3][co >> 2][co & 3]), "v"(m));
      if (!(OFX_ABL(p) & 4)) (obase + (size_t)co * (H / 2) * W2)[off] = m;
    }
  }
#endif
}

// ---- fp32 MFMA GEMM for the dense layers -------------------------------------------
// C[M][N] = act(A[M][K] (lda) x B[K][N] (ldb) + bias[N])   one wave per 32x32 tile,
// v_mfma_f32_32x32x2_f32: lane l holds A[row l&31][k l>>5], B[k l>>5][col l&31];
// C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).

__global__ __launch_bounds__(256) void k_gemm_f32(const float *A, int lda, const float *B, int ldb, const float *bias,
                                                  float *C, int ldc, int M, int N, int K, int relu) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int tiles_n = (N + 31) / 32;
  const int tile = blockIdx.x * 4 + wv;
  if (tile >= ((M + 31) / 32) * tiles_n) return;
  const int m0 = (tile / tiles_n) * 32, n0 = (tile % tiles_n) * 32;
  const int r = m0 + (lane & 31), c = n0 + (lane & 31), kh = lane >> 5;
  const bool rv = r < M, cv = c < N;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = 0.f;
  const float *ap = A + (size_t)(rv ? r : 0) * lda + kh;
  const float *bp = B + (size_t)kh * ldb + (cv ? c : 0);
  for (int k0 = 0; k0 < K; k0 += 2) {
    const bool kv = k0 + kh < K;
    const float a = (rv && kv) ? ap[k0] : 0.f;
    const float b = (cv && kv) ? bp[(size_t)k0 * ldb] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  if (cv) {
    const float bv = bias ? bias[c] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int row = m0 + (i & 3) + 8 * (i >> 2) + 4 * kh;
      if (row < M) {
        float v = acc[i] + bv;
        if (relu) v = fmaxf(v, 0.f);
        C[(size_t)row * ldc + c] = v;
      }
    }
  }
}

// Split-K form for the tall-K dense1 (M = arenas, N = 100, K = 5000): one wave per (32x32 tile, K chunk), partial
// sums P[chunk][M][N] reduced in a fixed order by the consumer (k_head_dense) - no atomics, bit-reproducible.
// K is walked 8 at a time with the k-slots permuted (slot kh of step j <-> k = k0 + 4 kh + j) so that a lane
// fetches its A operands with one 16-byte load per 4 MFMAs; Kc % 8 == 0, lda % 4 == 0, A 16-byte aligned.
__global__ __launch_bounds__(256) void k_gemm_f32_splitk(const float *A, int lda, const float *B, int ldb, float *P,
                                                         int M, int N, int Kc, int tiles, int jobs) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int tiles_n = (N + 31) / 32;
  const int job = blockIdx.x * 4 + wv;
  const int chunk = job / tiles, tile = job - chunk * tiles;
  if (job >= jobs) return;  // wave-uniform
  const int m0 = (tile / tiles_n) * 32, n0 = (tile % tiles_n) * 32;
  const int r = m0 + (lane & 31), c = n0 + (lane & 31), kh = lane >> 5;
  const bool rv = r < M, cv = c < N;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = 0.f;
  const float *ap = A + (size_t)(rv ? r : 0) * lda + (size_t)chunk * Kc + 4 * kh;
  const float *bp = B + ((size_t)chunk * Kc + 4 * kh) * ldb + (cv ? c : 0);
  for (int k0 = 0; k0 < Kc; k0 += 8) {
    const float4 a4 = *reinterpret_cast<const float4 *>(ap + k0);
    const float b0 = bp[(size_t)(k0 + 0) * ldb], b1 = bp[(size_t)(k0 + 1) * ldb], b2 = bp[(size_t)(k0 + 2) * ldb],
                b3 = bp[(size_t)(k0 + 3) * ldb];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(rv ? a4.x : 0.f, cv ? b0 : 0.f, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(rv ? a4.y : 0.f, cv ? b1 : 0.f, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(rv ? a4.z : 0.f, cv ? b2 : 0.f, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(rv ? a4.w : 0.f, cv ? b3 : 0.f, acc, 0, 0, 0);
  }
  if (cv) {
    float *out = P + (size_t)chunk * M * N;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int row = m0 + (i & 3) + 8 * (i >> 2) + 4 * kh;
      if (row < M) out[(size_t)row * N + c] = acc[i];
    }
  }
}

// ---- per-ship dense1 finish + head-1 ---------------------------------------------
// d1 = relu(G1[arena] + vec8 . K1[0:8] + b1) ; d2 = relu(d1 K2 + b2) ; act = d2 K3 + b3
struct HeadParams {
  int N, M;
  ofx_state st;
  const float *g1;            // [g1_chunks][N][100]  trunk part of dense1 (no bias), split-K partial sums
  int g1_chunks;
  const float *k1, *b1, *k2, *b2, *k3, *b3;
  const uint8_t *mask;
  const float *vec8;          // [S][8] explicit observation heads (ofx_policy_forward_obs) or null = the live state
  float *d1;                  // [S][100]
  float *act;                 // [S][2] or null
  int32_t *iaction;           // [S] or null
};

__global__ __launch_bounds__(256) void k_head_dense(HeadParams p) {
  __shared__ float sd1[4][100];
  __shared__ float sd2[4][50];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int s = blockIdx.x * 4 + wv;
  const bool on = s < p.N * p.M && (!p.mask || p.mask[s]);
  const int a = on ? s / p.M : 0;
  float vec[8];
  if (on) {  // obs.vector[:8] (observation.py:119-123): reward, can_shoot, pointing, dim, pos
    if (p.vec8) {
#pragma unroll
      for (int k = 0; k < 8; k++) vec[k] = p.vec8[(size_t)s * 8 + k];
    } else {
      vec[0] = (float)p.st.reward[s]; vec[1] = 1.f;
      vec[2] = (float)p.st.ship_px[s]; vec[3] = (float)p.st.ship_py[s];
      vec[4] = (float)PS; vec[5] = (float)PS;
      vec[6] = (float)p.st.ship_x[s]; vec[7] = (float)p.st.ship_y[s];
    }
    for (int o = lane; o < 100; o += 64) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 8; k++) acc += vec[k] * p.k1[k * 100 + o];
      float g = 0.f;
      for (int ch = 0; ch < p.g1_chunks; ch++) g += p.g1[((size_t)ch * p.N + a) * 100 + o];  // fixed order
      acc += g;
      acc += p.b1[o];
      acc = fmaxf(acc, 0.f);
      sd1[wv][o] = acc;
      p.d1[(size_t)s * 100 + o] = acc;
    }
  }
  __syncthreads();
  if (on && lane < 50) {
    float acc = 0.f;
    for (int k = 0; k < 100; k++) acc += sd1[wv][k] * p.k2[k * 50 + lane];
    sd2[wv][lane] = fmaxf(acc + p.b2[lane], 0.f);
  }
  __syncthreads();
  if (on && lane < 2) {
    float acc = 0.f;
    for (int k = 0; k < 50; k++) acc += sd2[wv][k] * p.k3[k * 2 + lane];
    acc += p.b3[lane];
    if (p.act) p.act[(size_t)s * 2 + lane] = acc;
    const float other = __shfl_xor(acc, 1);
    if (lane == 0 && p.iaction) p.iaction[s] = other > acc ? 1 : 0;  // np.argmax: first maximum
  }
}

// ---- last layer: bilinear x2 + conv3x3 (8 -> 1, linear) in 4-phase low-res form + arg-max ----
// Output (2i+a, 2j+b) = sum over the 3x3 low-res neighbourhood of (i,j) and 8 channels of
// Weff[a][b][ty][tx][ci] * L[i+ty-1][j+tx-1][ci], L clamp-extended.  The conv's zero padding
// differs from the clamp extension only for the 1-pixel frame of the output; those outputs
// subtract the taps that fall outside:  T = G - sum_{outside taps} w[dy][dx][ci] * U[clamp].
struct Up4Params {
  const float *in;            // planar [S][8][200][200]
  const float *weff, *wraw, *b4;
  const uint8_t *mask;
  unsigned long long *best;   // [S] packed (ordered value << 32) | ~index
  float *heat;                // [S][400][400] or null
};

constexpr int U4_TH = 10, U4_TW = 50, U4_LS = PS / 2;  // low-res tile, 1 x 2 low-res px per thread

// Taps of a frame output (y, x) that fall into the conv's zero padding, evaluated on the
// clamp-extended low-res tile: sum w[dy][dx][ci] * U[clamp(y+dy)][clamp(x+dx)][ci].
// Rare (1 % of the outputs): kept out of line and rolled so the hot path stays small.
__device__ __forceinline__ float frame_correction(const float *tile, const float *wraw, int y, int x, int i0, int j0) {
  constexpr int TWP = U4_TW + 2, PLANE = (U4_TH + 2) * TWP;
  float corr = 0.f;
#pragma unroll 1
  for (int tap = 0; tap < 9; tap++) {
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    const int uy = y + dy, ux = x + dx;
    if (uy >= 0 && uy < PS && ux >= 0 && ux < PS) continue;
    const int cy = min(max(uy, 0), PS - 1), cx = min(max(ux, 0), PS - 1);
    // U[cy][cx] (bilinear x2, half-pixel): rows ya, ya+1 with weight wy on the second
    const int ky = cy >> 1, kx = cx >> 1;
    const int ya = (cy & 1) ? ky : ky - 1, xa = (cx & 1) ? kx : kx - 1;
    const float wy = (cy & 1) ? 0.25f : 0.75f, wx = (cx & 1) ? 0.25f : 0.75f;
    const float *t0 = tile + (ya - (i0 - 1)) * TWP + (xa - (j0 - 1));
#pragma unroll 1
    for (int ci = 0; ci < 8; ci++) {
      const float *t = t0 + ci * PLANE;
      const float l00 = t[0], l01 = t[1], l10 = t[TWP], l11 = t[TWP + 1];
      const float top = l00 + (l01 - l00) * wx, bot = l10 + (l11 - l10) * wx;
      corr += wraw[tap * 8 + ci] * (top + (bot - top) * wy);
    }
  }
  return corr;
}

__device__ inline unsigned ordered_f32(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void k_upconv4(Up4Params p) {
  constexpr int TWP = U4_TW + 2;
  __shared__ __align__(16) float tile[8][U4_TH + 2][TWP];
  constexpr int tiles_x = U4_LS / U4_TW, tiles_y = U4_LS / U4_TH, tiles = tiles_x * tiles_y;
  const int s = blockIdx.x / tiles, t = blockIdx.x - s * tiles;
  if (p.mask && !p.mask[s]) return;
  const int i0 = (t / tiles_x) * U4_TH, j0 = (t % tiles_x) * U4_TW;
  const int tid = threadIdx.x;
  for (int e = tid; e < 8 * (U4_TH + 2) * TWP; e += 256) {
    const int c = e % TWP, r = (e / TWP) % (U4_TH + 2), ci = e / (TWP * (U4_TH + 2));
    const int gi = min(max(i0 - 1 + r, 0), U4_LS - 1), gj = min(max(j0 - 1 + c, 0), U4_LS - 1);  // clamp extension
    tile[ci][r][c] = p.in[(((size_t)s * 8 + ci) * U4_LS + gi) * U4_LS + gj];
  }
  __syncthreads();
  constexpr int NT = U4_TH * (U4_TW / 2);
  float bestv = -INFINITY;
  unsigned bestk = 0xFFFFFFFFu;
  if (tid < NT) {
    const int tr = tid / (U4_TW / 2), tc = tid - tr * (U4_TW / 2);
    float acc[2][4];  // [low-res px q][phase]
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
      for (int ph = 0; ph < 4; ph++) acc[q][ph] = 0.f;
#pragma unroll
    for (int ci = 0; ci < 8; ci++) {
      float v[3][4];
#pragma unroll
      for (int r = 0; r < 3; r++) {
        const float2 lo = *reinterpret_cast<const float2 *>(&tile[ci][tr + r][2 * tc]);
        const float2 hi = *reinterpret_cast<const float2 *>(&tile[ci][tr + r][2 * tc + 2]);
        v[r][0] = lo.x; v[r][1] = lo.y; v[r][2] = hi.x; v[r][3] = hi.y;
      }
#pragma unroll
      for (int ph = 0; ph < 4; ph++)
#pragma unroll
        for (int ty = 0; ty < 3; ty++)
#pragma unroll
          for (int tx = 0; tx < 3; tx++) {
            const float wv = p.weff[(ph * 9 + ty * 3 + tx) * 8 + ci];
            acc[0][ph] = __builtin_fmaf(v[ty][tx], wv, acc[0][ph]);
            acc[1][ph] = __builtin_fmaf(v[ty][tx + 1], wv, acc[1][ph]);
          }
    }
    const float bias = p.b4[0];
    const int li = i0 + tr;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int lj = j0 + 2 * tc + q;
#pragma unroll
      for (int ph = 0; ph < 4; ph++) {
        const int a = ph >> 1, b = ph & 1;
        const int y = 2 * li + a, x = 2 * lj + b;
        float val = acc[q][ph] + bias;
        if (y == 0 || y == PS - 1 || x == 0 || x == PS - 1)
          val -= frame_correction(&tile[0][0][0], p.wraw, y, x, i0, j0);
        const unsigned k = (unsigned)(y * PS + x);
        if (p.heat) p.heat[(size_t)s * PS * PS + k] = val;
        if (val > bestv || (val == bestv && k < bestk)) { bestv = val; bestk = k; }
      }
    }
  }
  // wave arg-max (first maximum in C order), then one 64-bit atomicMax per wave
  unsigned long long key = ((unsigned long long)ordered_f32(bestv) << 32) | (unsigned long long)(~bestk);
  if (bestk == 0xFFFFFFFFu) key = 0ull;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = __shfl_xor(key, o);
    key = other > key ? other : key;
  }
  if ((tid & 63) == 0 && key) atomicMax(&p.best[s], key);
}

// ---- fused head tail: [bilinear x2 + conv 4->8 + BN + ReLU] -> [bilinear x2 + conv 8->1] -> arg-max ----
// One workgroup per (ship, 40x40 tile of the 200x200 uprelu3 plane).  Both layers run in the 4-phase
// low-resolution form (a x2 bilinear upsample followed by a 3x3 conv is, per output parity, a 3x3 conv on the
// low-res grid with pre-combined weights), so no upsampled tensor is ever formed, and the 42 GB uprelu3 tensor
// (N=4096, M=8) with its 84 GB of HBM traffic never exists: it lives as a 42x42 LDS tile, 4 channels at a time.
//   stage A   24x24x4 clamp-extended patch of uprelu2 (100x100x4) -> LDS
//   stage B   uprelu3 on the MATRIX CORES: in phase form the layer is a GEMM out[quad][ph*4+co] =
//             sum_k in[quad][k] W[k][ph*4+co] with k = (tap, ci), K = 36 and a natural N = 4 phases x 4
//             channels = 16 per half: v_mfma_f32_16x16x4_f32 (exact fp32, an fma chain in k order), A gathered
//             from the LDS patch (one ds_read_b32 per lane per MFMA), B in 9 VGPRs per half.
//   stage C   heat-map quads on the VALU: each thread owns 2x4 uprelu3 pixels = 8 quads of the 400x400 map;
//             the 36 phase weights of a channel are broadcast LDS reads into VGPRs (an SGPR operand halves the
//             FMA rate on gfx950: 75 vs 115 TFLOP/s measured, tools/ubench_fma.hip), each feeds 8 FMAs.
//   The two channel halves run B(0) C(0) B(1) C(1) with the stage-C accumulators kept in registers, which
//   halves the LDS tile (30 KB) so three workgroups share a CU; the matrix pipe is separate from the VALU, so
//   one workgroup's stage B overlaps another's stage C.
//   Frame outputs (first/last row/column of either layer) are the only place where the conv's zero padding
//   differs from the clamp-extended phase form: border tiles recompute those cells from the definition
//   (upsample, then conv with zero padding) in a small rolled pass.
struct HeadTailParams {
  const float *up1;            // planar [S][2][50][50]: k_head_tail computes its uprelu2 patches itself
  const float *w2mf;           // [20][16] phase weights of upconv2 (MFMA B operand), PrepLayout::w2mf
  const float *w2raw, *b2;     // BN-folded [9][2][4], folded bias [4]
  const float *w3mf;           // [half 2][k = tap*4 + ci (36)][n = phase*4 + co_local (16)]  MFMA B operand
  const float *w3raw, *b3;     // BN-folded [9][4][8], folded bias [8]
  const float *w4eff;          // [ci 8][phase 4][tap 9]
  const float *w4raw, *b4;     // [9][8], [1]
  const float *efr;            // [2][2][2][3][8] frame phase weights of upconv4 (PrepLayout::efr)
  const uint8_t *mask;
  unsigned long long *best;
  float *heat;
  const int32_t *probe;        // [S][2] (x, y) or null: ptr_probe[s] = heat-map value at that pointer
  float *ptr_probe;
  int ablate;                  // diagnostics (OFX_HT_ABLATE): 1 no stage-A loads, 2 no stage B, 4 no stage C, 8 no border passes,
                               // 16 / 32 drop the barrier after stage C / stage B (timing only: the results are wrong)
};

constexpr int HT_T = 40;              // uprelu3 tile side
constexpr int HT_Q = HT_T / 2 + 2;    // quads per side (22)
constexpr int HT_QW = 24;             // padded quad row: 22 real quads + 2 duplicates of the last one, so that the
                                      // 4 accumulator registers of a lane are 4 consecutive quads of ONE row
constexpr int HT_NQP = HT_Q * HT_QW;  // 528
constexpr int HT_MT = HT_NQP / 16;    // 16-quad M-tiles (33)
constexpr int HT_L2 = HT_T / 2 + 4;   // uprelu2 patch side (24)
constexpr int HT_U3 = HT_T + 2;       // uprelu3 tile side incl. halo (42)
constexpr int HT_U3S = 48;            // storage row stride: halo columns 0..41, then the landing zone of the cells that
                                      // stage B computes but nobody reads (column 42 and the padded quads 43..46)
constexpr int HT_U3PL = (HT_U3 + 2) * HT_U3S + 4;  // plane: halo rows -1..42 (the outer rows are a landing zone too)
                                                   // + 4 floats in front for cell (-1, -1); multiple of 4 floats
constexpr int HT_S2 = 100, HT_S3 = 200;

// offset of the uprelu3 cell (channel, halo row, halo column) inside the LDS tile; halo (0, 0) = pixel (r0-1, c0-1)
__device__ __forceinline__ constexpr int u3o(int cl, int row, int col) {
  return cl * HT_U3PL + 4 + (row + 1) * HT_U3S + col;
}

// bilinear x2 (half-pixel) sample at up-res (uy, ux) of a clamp-extended low-res LDS plane whose element
// [0][0] has low-res coordinates (o_r, o_c)
__device__ __forceinline__ float up2d(const float *plane, int stride, int o_r, int o_c, int uy, int ux) {
  const int ya = (uy & 1) ? (uy >> 1) : (uy >> 1) - 1, xa = (ux & 1) ? (ux >> 1) : (ux >> 1) - 1;
  const float wy = (uy & 1) ? 0.25f : 0.75f, wx = (ux & 1) ? 0.25f : 0.75f;
  const float *q = plane + (ya - o_r) * stride + (xa - o_c);
  const float l00 = q[0], l01 = q[1], l10 = q[stride], l11 = q[stride + 1];
  const float top = l00 + (l01 - l00) * wx, bot = l10 + (l11 - l10) * wx;
  return top + (bot - top) * wy;
}

// x2 half-pixel bilinear along one axis of a clamp-extended line: value at up-res index u from the low-res
// samples line[(k - base) * stride]: even u = 2k -> .25 L[k-1] + .75 L[k]; odd -> .75 L[k] + .25 L[k+1]
__device__ __forceinline__ float up1d(const float *line, int stride, int base, int u) {
  const int k = u >> 1;
  const int ka = (u & 1) ? k : k - 1;
  const float w = (u & 1) ? 0.25f : 0.75f;
  const float l0 = line[(ka - base) * stride], l1 = line[(ka + 1 - base) * stride];
  return l0 + (l1 - l0) * w;
}

constexpr int HT_L2P = HT_L2 * HT_L2 + 16;  // plane stride of the patch: +16 floats so the 4 channel planes of
                                            // the MFMA A-gather land on different LDS banks
constexpr int HT_LB2 = HT_T + 4;            // frame-line length of stage B (x' in [c0-2, c0+T+1])
constexpr int HT_S1 = 50;                   // uprelu1 plane side
constexpr int HT_L1 = HT_L2 / 2 + 2;        // uprelu1 patch side (14)
constexpr int HT_L1P = HT_L1 * HT_L1 + 4;   // plane stride
constexpr int HT_LB1 = HT_L2 + 2;           // level-2 frame-line length (x' in [jb-1, jb+24])

// The f32 MFMAs and the VALU share the SIMD's issue slots on gfx950 (tools/ubench_mix.hip: their rates do not add
// up), so every VALU instruction costs matrix throughput.  The kernel is written to keep the VALU count per tile
// low: stage B writes its accumulators with one address computation per M-tile (immediate offsets for the 4
// registers, bias pre-loaded into the accumulator, no per-cell predicates), stage C has no VALU at all, and the
// arg-max is a 3-instruction compare/select per value on a compile-time slot id.
__global__ __launch_bounds__(256, 3) void k_head_tail(HeadTailParams p) {
  __shared__ __align__(16) float l2[4 * HT_L2P];
  __shared__ __align__(16) float u3f[4 * HT_U3PL];
  __shared__ __align__(16) float w4s[8][4][12];       // [channel][phase][tap, padded to 12: three aligned b128 per lane]
  __shared__ float wfr[192 + 288 + 72 + 72];        // border passes: efr | w3raw [9][4][8] | w4raw [9][8] | w2raw [9][2][4]
  __shared__ float l1[2 * HT_L1P];                  // uprelu1 patch 14 x 14 x 2, clamp-extended
  __shared__ float hb1[2][HT_LB1], vb1[2][HT_LB1];  // U1[0|99][x'] , U1[y'][0|99]: level-2 frame lines (border tiles)
  __shared__ float hb2[4][HT_LB2], vb2[4][HT_LB2];  // U2[0|199][x'] , U2[y'][0|199]  (border tiles)
  __shared__ float facc[2][2 * HT_T];               // zero-padding corrections of the heat-map frame pixels
  __shared__ unsigned gtab[HT_NQP / 4];             // quad group g = 4i -> byte offset of its first cell | qi << 16 | qj0 << 24
  __shared__ unsigned short atab[HT_NQP];           // padded quad -> qi*L2 + qj (A-operand gather base)
  // one workgroup walks the 5 tiles of one tile row of one ship: tables, weights and launch cost are paid once
  constexpr int tiles_x = HT_S3 / HT_T;
  const int s = blockIdx.x / tiles_x, trow = blockIdx.x - s * tiles_x;
  if (p.mask && !p.mask[s]) return;  // block-uniform
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction: keep it in an SGPR
  const int r0 = trow * HT_T, ib = r0 / 2 - 2;   // uprelu3 row of the tiles, uprelu2 row of the patch origin
  const bool top = r0 == 0, bot = r0 + HT_T == HT_S3, hline = top || bot;
  {
    const float wa = p.w4eff[tid], wb = tid < 32 ? p.w4eff[256 + tid] : 0.f;
    w4s[tid / 36][(tid % 36) / 9][tid % 9] = wa;
    if (tid < 32) w4s[(256 + tid) / 36][((256 + tid) % 36) / 9][(256 + tid) % 9] = wb;
    if (tid < 192) wfr[tid] = p.efr[tid];
    for (int e = tid; e < 288 + 72 + 72; e += 256)
      wfr[192 + e] = e < 288 ? p.w3raw[e] : e < 360 ? p.w4raw[e - 288] : p.w2raw[e - 360];
    for (int m = tid; m < HT_NQP; m += 256) {
      const int qi = m / HT_QW, qj = min(m - qi * HT_QW, HT_Q - 1);
      atab[m] = (unsigned short)(qi * HT_L2 + qj);
    }
    if (tid < HT_NQP / 4) {
      const int g = 4 * tid, qi = g / HT_QW, qj0 = g - qi * HT_QW;
      // quad (qi, qj), phase (pa, pb) -> halo cell (2 qi - 1 + pa, 2 qj - 1 + pb); the lane adds its (pa, pb, channel)
      gtab[tid] = (unsigned)(((2 * qi) * HT_U3S + 2 * qj0) * 4) | ((unsigned)qi << 16) | ((unsigned)qj0 << 24);
    }
  }
  float bestv = -INFINITY;
  unsigned bestk = 0xFFFFFFFFu;
  const float bias4 = p.b4[0];

  // software pipeline over the row: the uprelu1 patch of tile t+1 (14 x 14 x 2, clamp-extended) is fetched from HBM
  // while tile t computes; low-res origin of the patch = (r0/4 - 2, c0/4 - 2)
  constexpr int PATCH1 = 2 * HT_L1 * HT_L1;  // 392 = 2 per thread
  const int ib1 = r0 / 4 - 2;
  int poff[2], prow[2], pcol[2];
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int e = min(u * 256 + tid, PATCH1 - 1);
    const int ci = e / (HT_L1 * HT_L1), rem = e - ci * (HT_L1 * HT_L1), r = rem / HT_L1, c = rem - r * HT_L1;
    poff[u] = ci * HT_L1P + rem;
    prow[u] = (ci * HT_S1 + min(max(ib1 + r, 0), HT_S1 - 1)) * HT_S1;
    pcol[u] = c;
  }
  const float *up1s = p.up1 + (size_t)s * 2 * HT_S1 * HT_S1;
  float vals[2];
#pragma unroll
  for (int u = 0; u < 2; u++) vals[u] = up1s[prow[u] + min(max(pcol[u] - 2, 0), HT_S1 - 1)];

  // per-lane constants of the MFMA stage: n = lane & 15 -> (phase, local channel); kq = lane >> 4 -> input channel
  const int n16 = lane & 15, kq = lane >> 4;
  const int ph3 = n16 >> 2, cl3 = n16 & 3, pa3 = ph3 >> 1, pb3 = ph3 & 1;
  // stage-B output address: lane part (channel plane, phase row / column) + gtab part (quad group)
  char *const wbase = (char *)u3f + (cl3 * HT_U3PL + 4 + pa3 * HT_U3S + pb3 - 1) * 4;
  const float *const arow = &l2[kq * HT_L2P];
  // frame cells of the plane (y or x in {0,199}) keep G + bias WITHOUT ReLU: the border pass subtracts the
  // zero-padding taps first.  y = 0 <=> top tile, quad row 1, phase row 0 ; y = 199 <=> bottom tile, row Q-2, phase row 1;
  // x = 0 <=> left tile, quad column 1 (register 1 of group 0), phase column 0 ; x = 199 <=> right tile, quad column
  // Q-2 (register 0 of group 20), phase column 1
  const int fr_qi = (top && pa3 == 0) ? 1 : (bot && pa3 == 1) ? HT_Q - 2 : -1;
  // level 2 (uprelu2 patch from uprelu1, 12 x 12 low-res pixels = 9 M-tiles, K = 9 taps x 2 channels padded to 20):
  // A[pixel][k = 4 j + kq] = L1[ci = kq & 1][qi + ty][qj + tx], tap = 2 j + (kq >> 1); the two padding rows read cell 0
  int a2off[5];
  float bw2[5];
#pragma unroll
  for (int j = 0; j < 5; j++) {
    const int tap = 2 * j + (kq >> 1);
    a2off[j] = tap < 9 ? (kq & 1) * HT_L1P + (tap / 3) * HT_L1 + tap % 3 : 0;
    bw2[j] = p.w2mf[(4 * j + kq) * 16 + n16];
  }
  const float bias2 = p.b2[cl3];
  // frame cells of the uprelu2 plane inside the patch: y2 = 0 <=> top tile, low-res row 1, phase row 0; y2 = 99 <=>
  // bottom tile, row 10, phase row 1; x2 = 0 <=> left tile, column 1 (register 1 of group 0), phase column 0;
  // x2 = 99 <=> right tile, column 10 (register 2 of group 8), phase column 1
  const int fr2_qi = (top && pa3 == 0) ? 1 : (bot && pa3 == 1) ? 10 : -1;

  auto commit_patch = [&]() {  // the prefetched uprelu1 patch -> LDS
#pragma unroll
    for (int u = 0; u < 2; u++)
      if (u * 256 + tid < PATCH1) l1[poff[u]] = vals[u];
  };
  auto fetch_patch = [&](int tc) {  // request the patch of tile column tc
    const int jn = (tc * HT_T) / 4 - 2;
#pragma unroll
    for (int u = 0; u < 2; u++) vals[u] = up1s[prow[u] + min(max(jn + pcol[u], 0), HT_S1 - 1)];
  };
#if OFX_HT_PIPE
  // The patch of tile t + 1 is committed at the end of tile t (l1 is dead behind stage A), in front of the barrier that
  // closes stage C: that barrier doubles as the one stage A of the next tile needs, and the barrier at the end of a
  // tile goes too (facc is assigned in the first half instead of being zeroed at the top of the tile).
  commit_patch();
  if (1 < tiles_x) fetch_patch(1);
  __syncthreads();
#endif
#pragma unroll 1
  for (int tcol = 0; tcol < tiles_x; tcol++) {
  const int c0 = tcol * HT_T, jb = c0 / 2 - 2;
  const bool lef = c0 == 0, rig = c0 + HT_T == HT_S3;
  const bool vline = lef || rig, border = hline || vline;
  const int fc_q0 = (lef && pb3 == 0) ? 0 : (rig && pb3 == 1) ? HT_Q - 2 : -1, fc_i = lef ? 1 : 0;
  // 400 stage-C lane-tasks = 6.25 waves: three waves run two passes, one runs a single pass and takes the 33rd
  // M-tile of stage B instead; the light wave rotates so that the four SIMDs of the CU see the same load
  const int light = (tcol + (int)blockIdx.x) & 3;

  // ---- stage A: the prefetched uprelu1 patch goes to LDS (the next tile's loads are issued right away), then the
  // uprelu2 patch of the tile is computed in place: upconv2 in phase form on the matrix cores, so the 5 GB uprelu2
  // tensor never exists either ----
  const int jb1 = c0 / 4 - 2;
  const int fc2_q0 = (lef && pb3 == 0) ? 0 : (rig && pb3 == 1) ? 8 : -1, fc2_i = lef ? 1 : 2;
  // stage-B weights of the first channel half: requested here, a whole stage A ahead of their use (they come from
  // global memory / L2; loaded at the top of stage B the first MFMAs of every half waited for them)
  float bw[9], bias3;
  auto load_bw = [&](int half) {
#pragma unroll
    for (int j = 0; j < 9; j++) bw[j] = p.w3mf[(half * 36 + 4 * j + kq) * 16 + n16];
    bias3 = p.b3[4 * half + cl3];
  };
#if OFX_HTB_WEARLY == 1
  load_bw(0);
#endif
#if !OFX_HT_PIPE
  commit_patch();
  if (tcol + 1 < tiles_x) fetch_patch(tcol + 1);
  __syncthreads();
#endif
  // level-2 frame lines (border tiles): rows / columns of the upsampled uprelu1 plane next to the frame of uprelu2; they
  // depend on the l1 patch only, so with OFX_HTA_FEWBAR they are built in stage A's phase and need no barrier of their own
  auto lines1 = [&]() {
    const int nl = (hline ? 1 : 0) + (vline ? 1 : 0);
    const int ib = r0 / 2 - 2, jb = c0 / 2 - 2;
    for (int e = tid; e < nl * 2 * HT_LB1; e += 256) {
      const int li = e / (2 * HT_LB1), rem = e - li * 2 * HT_LB1, ci = rem / HT_LB1, k = rem - ci * HT_LB1;
      if (hline && li == 0) {
        const int R = top ? 0 : HT_S1 - 1, xc = min(max(jb - 1 + k, 0), HT_S2 - 1);
        hb1[ci][k] = up1d(&l1[ci * HT_L1P + (R - ib1) * HT_L1], 1, jb1, xc);
      } else {
        const int Cc = lef ? 0 : HT_S1 - 1, yc = min(max(ib - 1 + k, 0), HT_S2 - 1);
        vb1[ci][k] = up1d(&l1[ci * HT_L1P + (Cc - jb1)], HT_L1, ib1, yc);
      }
    }
  };
#if OFX_HTA_FEWBAR
  if (border) lines1();
#endif
  {
    const f32x4 binit2 = {bias2, bias2, bias2, bias2};
    auto arow2 = [&](int mt) -> const float * {
      const int m = 16 * mt + n16, qi = m / 12, qj = m - qi * 12;
      return &l1[qi * HT_L1 + qj];
    };
    auto epi2 = [&](int mt, const f32x4 d) {
      // D: the 4 consecutive low-res pixels of group 16 mt + 4 kq (one row: 12 = 3 groups), column n = (phase, co)
      const int g = 16 * mt + 4 * kq, gi = g / 12, gj0 = g - gi * 12;
      const float rf = gi == fr2_qi ? -INFINITY : 0.f;  // -inf keeps a frame cell raw for the border pass
      const bool cf = gj0 == fc2_q0;
      float *w = &l2[cl3 * HT_L2P + (2 * gi + pa3) * HT_L2 + 2 * gj0 + pb3];
#pragma unroll
      for (int i = 0; i < 4; i++) w[2 * i] = max_raw(d[i], (cf && fc2_i == i) ? -INFINITY : rf);
    };
#if OFX_HTA_ILP
    {  // M-tiles wv and wv + 4 as two interleaved chains (a lone chain of 5 dependent MFMAs is latency bound), then wave
       // 0's third one: the wave with three M-tiles sets the length of the stage
      const float *a0 = arow2(wv), *a1 = arow2(wv + 4);
      f32x4 d0 = binit2, d1 = binit2;
#pragma unroll
      for (int j = 0; j < 5; j++) {
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[a2off[j]], bw2[j], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[a2off[j]], bw2[j], d1, 0, 0, 0);
      }
      if (wv == 0) {
        const float *a2 = arow2(8);
        f32x4 d2 = binit2;
#pragma unroll
        for (int j = 0; j < 5; j++) d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[a2off[j]], bw2[j], d2, 0, 0, 0);
        epi2(0, d0);
        epi2(4, d1);
        epi2(8, d2);
      } else {
        epi2(wv, d0);
        epi2(wv + 4, d1);
      }
    }
#else
#pragma unroll 1
    for (int mt = wv; mt < 9; mt += 4) {
      const float *a = arow2(mt);
      f32x4 d = binit2;
#pragma unroll
      for (int j = 0; j < 5; j++) d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[a2off[j]], bw2[j], d, 0, 0, 0);
      epi2(mt, d);
    }
#endif
  }
#if OFX_HTB_WEARLY == 2
  load_bw(0);  // in flight across the barrier (and the level-2 border passes)
#endif
  __syncthreads();
  if (border) {
    // level-2 frame: lines of the upsampled uprelu1 plane, then the zero-padding correction of the frame cells of
    // uprelu2 (wave = output channel, lane = cell of the line) and the clamp copies into the cells outside the plane
    const int ib = r0 / 2 - 2, jb = c0 / 2 - 2;
#if !OFX_HTA_FEWBAR
    lines1();
    __syncthreads();
#endif
    if (lane < HT_L2) {
      const float *w = &wfr[192 + 288 + 72 + wv];  // w2raw[(tap * 2 + ci) * 4 + co], co = wave
      float *pl = &l2[wv * HT_L2P];
      auto frame_cell2 = [&](int y, int x) {
        const bool fy = y == 0 || y == HT_S2 - 1, fx = x == 0 || x == HT_S2 - 1;
        float corr = 0.f;
        if (fy) {
          const int trow = (y == 0) ? 0 : 2;
#pragma unroll
          for (int dx = -1; dx <= 1; dx++) {
            const int xx = min(max(x + dx, 0), HT_S2 - 1) - (jb - 1);
#pragma unroll
            for (int ci = 0; ci < 2; ci++) corr += w[((trow * 3 + dx + 1) * 2 + ci) * 4] * hb1[ci][xx];
          }
        }
        if (fx) {
          const int tcol = (x == 0) ? 0 : 2;
#pragma unroll
          for (int dy = -1; dy <= 1; dy++) {
            const int uy = y + dy;
            if (uy < 0 || uy >= HT_S2) continue;  // counted with the row
#pragma unroll
            for (int ci = 0; ci < 2; ci++) corr += w[(((dy + 1) * 3 + tcol) * 2 + ci) * 4] * vb1[ci][uy - (ib - 1)];
          }
        }
        const int pr = y - ib, pc = x - jb;
        const float v = fmaxf(pl[pr * HT_L2 + pc] - corr, 0.f);
        // the patch reaches 2 cells beyond the plane: they are clamp copies of the frame cell
        const int oy = (y == 0) ? -1 : (y == HT_S2 - 1) ? 1 : 0, ox = (x == 0) ? -1 : (x == HT_S2 - 1) ? 1 : 0;
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
          for (int b = 0; b < 3; b++)
            if ((a == 0 || oy) && (b == 0 || ox)) pl[(pr + a * oy) * HT_L2 + pc + b * ox] = v;
      };
      if (hline) {
        const int x = jb + lane;
        if (x >= 0 && x < HT_S2) frame_cell2(top ? 0 : HT_S2 - 1, x);
      }
      if (vline) {  // the corner cells belong to the horizontal line
        const int y = ib + lane;
        if (y > 0 && y < HT_S2 - 1) frame_cell2(y, lef ? 0 : HT_S2 - 1);
      }
    }
    __syncthreads();
  }
  if (border) {  // frame lines of the upsampled uprelu2 plane (block-uniform); only the lines this tile has
    const int nl = (hline ? 1 : 0) + (vline ? 1 : 0);
    for (int e = tid; e < nl * 4 * HT_LB2; e += 256) {
      const int li = e / (4 * HT_LB2), rem = e - li * 4 * HT_LB2, ci = rem / HT_LB2, k = rem - ci * HT_LB2;
      if (hline && li == 0) {  // U2[0] = L[0], U2[199] = L[99] (row clamp): an x-lerp of one patch row
        const int R = top ? 0 : HT_S2 - 1, xc = min(max(c0 - 2 + k, 0), HT_S3 - 1);
        hb2[ci][k] = up1d(&l2[ci * HT_L2P + (R - ib) * HT_L2], 1, jb, xc);
      } else {
        const int Cc = lef ? 0 : HT_S2 - 1, yc = min(max(r0 - 2 + k, 0), HT_S3 - 1);
        vb2[ci][k] = up1d(&l2[ci * HT_L2P + (Cc - jb)], HT_L2, ib, yc);
      }
    }
#if !OFX_HTA_FEWBAR
    __syncthreads();  // not needed: hb2 / vb2 are read behind stage B's barrier, and stage B does not write l2
#endif
  }

  // stage-C ownership: a lane owns 4 horizontally adjacent uprelu3 pixels (one aligned b128 LDS read serves four
  // MFMA B operands); 400 such lane-tasks per tile = pass 0 (all 256 lanes) + pass 1 (144 lanes of the three
  // non-light waves).  cacc[pass][pixel] = the 4 output phases of that pixel's heat-map quad, bias pre-loaded.
  constexpr int CT = (HT_T * HT_T) / 4;          // 400 lane-tasks
  const int hrank = wv - (wv > light ? 1 : 0);   // rank of a non-light wave: 0, 1, 2
#if OFX_HTC_MINI == 2
  // no second wave-pass at all: tasks 256..399 are nine "quarter passes" of 16 tasks = 64 pixels, one pixel per lane
  // (36 MFMAs per half each).  Ranks 0 / 1 take three, rank 2 two, the light wave one next to its 33rd M-tile and the
  // frame lines: the longest wave runs 1.75 pass-times of stage C instead of 2
  const bool pass1 = false;
  const int nq = wv == light ? 1 : (hrank < 2 ? 3 : 2), k0 = wv == light ? 8 : 3 * hrank;  // wave-uniform
#elif OFX_HTC_MINI
  // pass 1 = tasks 256..383 on the non-light waves of rank 0 and 1 (all lanes busy); the last 16 tasks (64 pixels, tile
  // rows 38 / 39) go to the rank-2 wave as one pixel per lane: 36 MFMAs per half instead of the 144 of a wave-pass
  // with 48 idle lanes
  const bool pass1 = wv != light && hrank < 2;   // wave-uniform
  const int nq = (wv != light && hrank == 2) ? 1 : 0, k0 = 8;
#else
  const bool pass1 = wv != light;                // wave-uniform
#endif
#if OFX_HTC_MINI
  int mrow[3], mcol[3], moff[3];                 // quarter-pass slot i < nq: tasks 256 + 16 (k0 + i) .. + 15
  f32x4 macc[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const int mt = min(256 + 16 * (k0 + i) + (lane >> 2), CT - 1);
    mrow[i] = mt / (HT_T / 4);
    mcol[i] = 4 * (mt % (HT_T / 4)) + (lane & 3);
    moff[i] = u3o(0, mrow[i], mcol[i]);
    macc[i] = (f32x4){bias4, bias4, bias4, bias4};
  }
#endif
  int ctask[2], coff[2];                         // task, LDS offset of its window origin inside a channel plane
  f32x4 cacc[2][4];
  ctask[0] = tid;
  ctask[1] = min(256 + hrank * 64 + lane, CT - 1);
#pragma unroll
  for (int q = 0; q < 2; q++) {
    coff[q] = u3o(0, ctask[q] / (HT_T / 4), 4 * (ctask[q] % (HT_T / 4)));
#pragma unroll
    for (int g = 0; g < 4; g++) cacc[q][g] = (f32x4){bias4, bias4, bias4, bias4};
  }

#pragma unroll 1
  for (int half = 0; half < 2; half++) {
    // ---- stage B: 4 channels of the uprelu3 tile on the matrix cores ----
    if (!(OFX_ABL(p) & 2)) {
#if OFX_HTB_WEARLY == 0
      load_bw(half);
#endif
      const f32x4 binit = {bias3, bias3, bias3, bias3};
      // two M-tiles per iteration, their MFMA chains interleaved (16x16x4: 32-cycle issue, 40-cycle dependent
      // latency -> two independent accumulators keep the matrix pipe full)
      auto run = [&](auto BT) {
        constexpr bool BORDER = decltype(BT)::value;
        auto floors = [&](unsigned g, float *f) {  // ReLU floor of the 4 cells of a group; -inf keeps a frame cell raw
          f[0] = f[1] = f[2] = f[3] = 0.f;
          if (BORDER) {
            const float rf = (int)((g >> 16) & 0xFFu) == fr_qi ? -INFINITY : 0.f;
            const bool cf = (int)(g >> 24) == fc_q0;
            f[0] = (cf && fc_i == 0) ? -INFINITY : rf; f[1] = (cf && fc_i == 1) ? -INFINITY : rf; f[2] = f[3] = rf;
          }
        };
        auto epiB = [&](int mt, const f32x4 d) {
          // D: col = lane & 15, row = 4 (lane >> 4) + reg -> the 4 consecutive quads of group mt * 4 + kq
          const unsigned g = gtab[mt * 4 + kq];
          float f[4];
          floors(g, f);
          float *w = (float *)(wbase + (g & 0xFFFFu));
#pragma unroll
          for (int i = 0; i < 4; i++) w[2 * i] = max_raw(d[i], f[i]);
        };
        auto pairB = [&](int it) {
          const int mt0 = wv + 8 * it, mt1 = mt0 + 4;
          const float *a0 = arow + atab[mt0 * 16 + n16];  // A row = quad m + (lane & 15), k = 4 j + kq
          const float *a1 = arow + atab[mt1 * 16 + n16];
          f32x4 d0 = binit, d1 = binit;
#pragma unroll
          for (int j = 0; j < 9; j++) {  // tap j, channel kq
            d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[(j / 3) * HT_L2 + (j % 3)], bw[j], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[(j / 3) * HT_L2 + (j % 3)], bw[j], d1, 0, 0, 0);
          }
          epiB(mt0, d0);
          epiB(mt1, d1);
        };
#if OFX_HTB_TRIPLE
        // the light wave's odd 33rd M-tile rides as a third chain in its last iteration instead of a lone chain of
        // nine dependent MFMAs behind the loop: the wave with nine M-tiles sets the length of stage B
#pragma unroll 1
        for (int it = 0; it < 3; it++) pairB(it);
        if (wv == light) {
          const int mt0 = wv + 24, mt1 = mt0 + 4, mt2 = HT_MT - 1;
          const float *a0 = arow + atab[mt0 * 16 + n16];
          const float *a1 = arow + atab[mt1 * 16 + n16];
          const float *a2 = arow + atab[mt2 * 16 + n16];
          f32x4 d0 = binit, d1 = binit, d2 = binit;
#pragma unroll
          for (int j = 0; j < 9; j++) {
            d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[(j / 3) * HT_L2 + (j % 3)], bw[j], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[(j / 3) * HT_L2 + (j % 3)], bw[j], d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[(j / 3) * HT_L2 + (j % 3)], bw[j], d2, 0, 0, 0);
          }
          epiB(mt0, d0);
          epiB(mt1, d1);
          epiB(mt2, d2);
        } else {
          pairB(3);
        }
#else
#pragma unroll OFX_HTB_UNROLL
        for (int it = 0; it < 4; it++) pairB(it);
        if (wv == light) {  // the odd 33rd M-tile: one chain
          const float *a0 = arow + atab[(HT_MT - 1) * 16 + n16];
          f32x4 d0 = binit;
#pragma unroll
          for (int j = 0; j < 9; j++)
            d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[(j / 3) * HT_L2 + (j % 3)], bw[j], d0, 0, 0, 0);
          epiB(HT_MT - 1, d0);
        }
#endif
      };
      if (border) run(std::true_type{}); else run(std::false_type{});
    }
    if (!(OFX_ABL(p) & 32)) __syncthreads();
    if (border && !(OFX_ABL(p) & 8)) {
      // Frame cells (row/col 0 or 199 of the plane) hold G + bias without ReLU: subtract the taps that fall
      // into the conv's zero padding, sum w[tap][ci] U2[clamp] from the frame lines, then apply the ReLU.
      // Wave = local channel, lane = cell of the line: the weights are wave-uniform LDS broadcasts.
      if (lane < HT_U3) {
        const float *w = &wfr[192 + 4 * half + wv];
        auto frame_cell = [&](int y, int x) {
          const bool fy = y == 0 || y == HT_S3 - 1, fx = x == 0 || x == HT_S3 - 1;
          float corr = 0.f;
          if (fy) {
            const int trow = (y == 0) ? 0 : 2;
#pragma unroll
            for (int dx = -1; dx <= 1; dx++) {
              const int xx = min(max(x + dx, 0), HT_S3 - 1) - (c0 - 2);
#pragma unroll
              for (int ci = 0; ci < 4; ci++) corr += w[((trow * 3 + dx + 1) * 4 + ci) * 8] * hb2[ci][xx];
            }
          }
          if (fx) {
            const int tcol = (x == 0) ? 0 : 2;
#pragma unroll
            for (int dy = -1; dy <= 1; dy++) {
              const int uy = y + dy;
              if (uy < 0 || uy >= HT_S3) continue;  // counted with the row
#pragma unroll
              for (int ci = 0; ci < 4; ci++) corr += w[(((dy + 1) * 3 + tcol) * 4 + ci) * 8] * vb2[ci][uy - (r0 - 2)];
            }
          }
          const int ty = y - (r0 - 1), tx = x - (c0 - 1);
          const float v = fmaxf(u3f[u3o(wv, ty, tx)] - corr, 0.f);
          u3f[u3o(wv, ty, tx)] = v;
          // the halo cells outside the image are clamp copies of exactly these frame cells
          const int oy = (y == 0) ? -1 : (y == HT_S3 - 1) ? 1 : 0, ox = (x == 0) ? -1 : (x == HT_S3 - 1) ? 1 : 0;
          if (oy) u3f[u3o(wv, ty + oy, tx)] = v;
          if (ox) u3f[u3o(wv, ty, tx + ox)] = v;
          if (oy && ox) u3f[u3o(wv, ty + oy, tx + ox)] = v;
        };
        if (hline) {
          const int x = c0 - 1 + lane;
          if (x >= 0 && x < HT_S3) frame_cell(top ? 0 : HT_S3 - 1, x);
        }
        if (vline) {  // the corner cells belong to the horizontal line
          const int y = r0 - 1 + lane;
          if (y > 0 && y < HT_S3 - 1) frame_cell(y, lef ? 0 : HT_S3 - 1);
        }
      }
      __syncthreads();
      // Zero-padding corrections of the heat-map frame pixels of this tile, accumulated over the two halves:
      // facc[0][k] for (y in {0,399}, x = 2c0 + k) ; facc[1][k] for (y = 2r0 + k, x in {0,399}), rows counted once.
      // In phase form along the line (PrepLayout::efr): pixel 2j + b gets sum_o E[b][o] L[j + o - 1] of the
      // clamp-extended low-res frame row / column L of the tile.  Wave 0 = horizontal line, wave 1 = vertical line,
      // lane = j; the other two waves go straight to stage C.
#if OFX_HT_FACC_LIGHT
      // both lines on the LIGHT wave: it runs one stage-C pass where the others run two, so the lines ride in its slack
      // instead of lengthening the chain of waves 0 and 1
      for (int ln = 0; ln < 2; ln++) {
        if (!(wv == light && lane < HT_T && (ln ? vline : hline))) continue;
#else
      for (int ln = 0; ln < 2; ln++) {  // wave 0 = horizontal line, wave 1 = vertical line
        if (!(wv == ln && lane < HT_T && (ln ? vline : hline))) continue;
#endif
        const int side = ln ? (lef ? 0 : 1) : (top ? 0 : 1);
        const float *E = &wfr[(ln * 2 + side) * 48 + 4 * half];  // [b][o][ci]
        const int R = (top ? 0 : HT_S3 - 1) - (r0 - 1), Cc = (lef ? 0 : HT_S3 - 1) - (c0 - 1);
        const float *L = u3f + (ln ? u3o(0, lane, Cc) : u3o(0, R, lane));  // sample j - 1
        const int st = ln ? HT_U3S : 1;
        float e0 = 0.f, e1 = 0.f;
#pragma unroll
        for (int cl = 0; cl < 4; cl++)
#pragma unroll
          for (int o = 0; o < 3; o++) {
            const float v = L[cl * HT_U3PL + o * st];
            e0 += E[o * 8 + cl] * v;
            e1 += E[(3 + o) * 8 + cl] * v;
          }
        if (ln == 1) {  // corner pixels: the conv row outside the image is counted with the horizontal line
          const float *w4 = &wfr[192 + 288 + 4 * half];
          const int tcol = lef ? 0 : 2;
          if (top && lane == 0)
#pragma unroll
            for (int cl = 0; cl < 4; cl++) e0 -= w4[(0 * 3 + tcol) * 8 + cl] * u3f[u3o(cl, 1, Cc)];
          if (bot && lane == HT_T - 1)
#pragma unroll
            for (int cl = 0; cl < 4; cl++) e1 -= w4[(2 * 3 + tcol) * 8 + cl] * u3f[u3o(cl, HT_T, Cc)];
        }
        if (half == 0) {  // every entry a tile reads is written in both halves: no zeroing pass
          facc[ln][2 * lane] = e0;
          facc[ln][2 * lane + 1] = e1;
        } else {
          facc[ln][2 * lane] += e0;
          facc[ln][2 * lane + 1] += e1;
        }
      }
    }

    // ---- stage C: 4 channels of the heat-map quads, also on the matrix cores.  Per output pixel the layer is
    // out[phase] = sum_k in[k] W[k][phase], k = (channel, tap): v_mfma_f32_4x4x1_16B_f32 with CBSZ = 4 broadcasts the
    // A block of lanes 0-3 (the 4 phase weights of tap k) to all 16 blocks, B = one input value per lane (pixel):
    // D[phase][pixel] += W[k][phase] * in[pixel][k] -- an M = 4, N = 64, K = 1 step with N = 4 phases exactly (no
    // padding) and not a single VALU instruction.  36 steps per half and group.
    if (!(OFX_ABL(p) & 4)) {
      // Software pipeline: the two LDS rows of step t + 2 (step = channel, tap row: 12 MFMAs = 96 matrix-pipe cycles)
      // and the 3 x b128 weights of the next channel are requested before the MFMAs of step t; the scheduling groups
      // pin that order (left alone the compiler issues each read right before its first use and waits for it).
      const f32x4 *wrow = reinterpret_cast<const f32x4 *>(&w4s[4 * half][lane & 3][0]);  // A: lane i < 4 = phase i
#pragma unroll
      for (int q = 0; q < 2; q++) {
        if (q == 1 && !pass1) break;
        const float *base = u3f + coff[q];
        f32x4 L[3], H[3];
        f32x4 W[2][3];
        auto ldrow = [&](int t) {  // two b128 (of the second only .xy is used: a b64 would get merged across steps)
          const float *row = base + (t / 3) * HT_U3PL + (t % 3) * HT_U3S;
          L[t % 3] = *reinterpret_cast<const f32x4 *>(row);
#if OFX_HTC_H64
          const f32x2 h2 = *reinterpret_cast<const f32x2 *>(row + 4);
          H[t % 3][0] = h2[0];
          H[t % 3][1] = h2[1];
#else
          H[t % 3] = *reinterpret_cast<const f32x4 *>(row + 4);
#endif
        };
        auto ldw = [&](int cl) {
#pragma unroll
          for (int i = 0; i < 3; i++) W[cl & 1][i] = wrow[cl * 12 + i];
        };
        ldw(0);
        ldrow(0);
        ldrow(1);
#if OFX_HTC_FENCE
        __builtin_amdgcn_sched_barrier(0);
#endif
        auto step = [&](auto TT) {
          constexpr int t = decltype(TT)::value, cl = t / 3, a = t % 3;
          if (t + 2 < 12) ldrow(t + 2);
          if (a == 0 && cl + 1 < 4) ldw(cl + 1);
          const float v[6] = {L[a][0], L[a][1], L[a][2], L[a][3], H[a][0], H[a][1]};
#pragma unroll
          for (int b = 0; b < 3; b++) {
            const int k = a * 3 + b;
            const float w = W[cl & 1][k >> 2][k & 3];
#pragma unroll
            for (int g = 0; g < 4; g++)  // 4 independent accumulator chains
              cacc[q][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(w, v[g + b], cacc[q][g], 4, 0, 0);
          }
#if OFX_HTC_GROUP
          constexpr int nread = (t + 2 < 12 ? 2 : 0) + (a == 0 && cl + 1 < 4 ? 3 : 0);
          if constexpr (nread > 0) __builtin_amdgcn_sched_group_barrier(0x100, nread, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
#endif
#if OFX_HTC_FENCE == 1
          __builtin_amdgcn_sched_barrier(0);
#elif OFX_HTC_FENCE == 2
          if constexpr (a == 2) __builtin_amdgcn_sched_barrier(0);  // one fence per channel
#endif
        };
        static_for<12>(step);
      }
#if OFX_HTC_MINI
      // one pixel per lane: three taps of a row are three consecutive floats (unaligned: b32 reads); the slots of a
      // wave are independent MFMA chains and share the weights
      auto quarters = [&](auto NQ) {
        constexpr int nqc = decltype(NQ)::value;
#pragma unroll
        for (int cl = 0; cl < 4; cl++) {
          f32x4 Wm[3];
#pragma unroll
          for (int i = 0; i < 3; i++) Wm[i] = wrow[cl * 12 + i];
#pragma unroll
          for (int a = 0; a < 3; a++) {
            float v[nqc][3];
#pragma unroll
            for (int i = 0; i < nqc; i++)
#pragma unroll
              for (int b = 0; b < 3; b++) v[i][b] = u3f[moff[i] + cl * HT_U3PL + a * HT_U3S + b];
#pragma unroll
            for (int b = 0; b < 3; b++)
#pragma unroll
              for (int i = 0; i < nqc; i++)
                macc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(Wm[(3 * a + b) >> 2][(3 * a + b) & 3], v[i][b], macc[i], 4, 0, 0);
          }
        }
      };
      if (nq == 3) quarters(std::integral_constant<int, 3>{});
      else if (nq == 2) quarters(std::integral_constant<int, 2>{});
      else if (nq == 1) quarters(std::integral_constant<int, 1>{});
#endif
    }
#if OFX_HTB_WEARLY
    if (half == 0) load_bw(1);  // in flight across the barrier
#endif
#if OFX_HT_PIPE
    if (half == 1 && tcol + 1 < tiles_x) {
      commit_patch();
      if (tcol + 2 < tiles_x) fetch_patch(tcol + 2);
    }
#endif
    if (!(OFX_ABL(p) & 16)) __syncthreads();  // the tile is overwritten by the next half
  }

  // ---- outputs + arg-max (first maximum in C order) ----
  if (border) {  // zero-padding corrections of the frame pixels: only the lanes that own them
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int lt = ctask[q] / (HT_T / 4), c4 = ctask[q] - lt * (HT_T / 4);
      if (hline && lt == (top ? 0 : HT_T - 1)) {  // y = 0 (phase row 0) or y = 399 (phase row 1)
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
          for (int b = 0; b < 2; b++) {
            const float c = facc[0][2 * (4 * c4 + g) + b];
            if (top) cacc[q][g][b] -= c; else cacc[q][g][2 + b] -= c;
          }
      }
      if (vline && c4 == (lef ? 0 : HT_T / 4 - 1)) {  // x = 0 (pixel 0, phase column 0) or x = 399
#pragma unroll
        for (int r = 0; r < 2; r++) {
          const float c = facc[1][2 * lt + r];
          if (lef) cacc[q][0][2 * r] -= c; else cacc[q][3][2 * r + 1] -= c;
        }
      }
    }
  }
#if OFX_HTC_MINI
  if (border) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
      if (i >= nq) break;  // wave-uniform
      if (hline && mrow[i] == (top ? 0 : HT_T - 1)) {
#pragma unroll
        for (int b = 0; b < 2; b++) {
          const float c = facc[0][2 * mcol[i] + b];
          if (top) macc[i][b] -= c; else macc[i][2 + b] -= c;
        }
      }
      if (vline && mcol[i] == (lef ? 0 : HT_T - 1)) {
#pragma unroll
        for (int r = 0; r < 2; r++) {
          const float c = facc[1][2 * mrow[i] + r];
          if (lef) macc[i][2 * r] -= c; else macc[i][2 * r + 1] -= c;
        }
      }
    }
  }
#endif
  if (p.heat) {
#if OFX_HTC_MINI
#pragma unroll
    for (int i = 0; i < 3; i++) {
      if (i >= nq) break;
#pragma unroll
      for (int ph = 0; ph < 4; ph++)
        p.heat[(size_t)s * PS * PS + (size_t)(2 * (r0 + mrow[i]) + (ph >> 1)) * PS + 2 * (c0 + mcol[i]) + (ph & 1)] = macc[i][ph];
    }
#endif
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int task = q * 256 + tid;  // a light wave's q = 1 registers hold nothing
      const bool mine = q == 0 ? true : (pass1 && 256 + hrank * 64 + lane < CT);
      if (!mine) continue;
      (void)task;
      const int li = r0 + ctask[q] / (HT_T / 4), lj0 = c0 + 4 * (ctask[q] % (HT_T / 4));
#pragma unroll
      for (int g = 0; g < 4; g++)
#pragma unroll
        for (int ph = 0; ph < 4; ph++)
          p.heat[(size_t)s * PS * PS + (size_t)(2 * li + (ph >> 1)) * PS + 2 * (lj0 + g) + (ph & 1)] = cacc[q][g][ph];
    }
  }
  if (p.ptr_probe) {  // block-uniform; one pixel per ship: the lane that owns it stores it
    const int pk = p.probe[2 * s + 1] * PS + p.probe[2 * s];
#if OFX_HTC_MINI
#pragma unroll
    for (int i = 0; i < 3; i++) {
      if (i >= nq) break;
#pragma unroll
      for (int ph = 0; ph < 4; ph++)
        if ((2 * (r0 + mrow[i]) + (ph >> 1)) * PS + 2 * (c0 + mcol[i]) + (ph & 1) == pk) p.ptr_probe[s] = macc[i][ph];
    }
#endif
#pragma unroll
    for (int q = 0; q < 2; q++) {
      if (q == 1 && !pass1) break;
      const int li = r0 + ctask[q] / (HT_T / 4), lj0 = c0 + 4 * (ctask[q] % (HT_T / 4));
#pragma unroll
      for (int g = 0; g < 4; g++)
#pragma unroll
        for (int ph = 0; ph < 4; ph++)
          if ((2 * li + (ph >> 1)) * PS + 2 * (lj0 + g) + (ph & 1) == pk) p.ptr_probe[s] = cacc[q][g][ph];
    }
  }
  {
    // per thread the 32 values are visited in increasing flat index, so a strict > keeps the first maximum; the slot
    // is a compile-time constant (3 VALU instructions per value); lanes past the 400th task hold copies of task 399
    float tv = -INFINITY;
    int ts = 0;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      if (q == 1 && !pass1) break;
#pragma unroll
      for (int r = 0; r < 2; r++)
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
          for (int b = 0; b < 2; b++) {
            const float val = cacc[q][g][2 * r + b];
            const bool gt = val > tv;
            tv = gt ? val : tv;
            ts = gt ? (q * 16 + r * 8 + g * 2 + b) : ts;
          }
    }
#if OFX_HTC_MINI
    // the one-pixel tasks (>= 256, in increasing order) lie behind every pass-0 / pass-1 task of this lane in C order
#pragma unroll
    for (int i = 0; i < 3; i++) {
      if (i >= nq) break;
#pragma unroll
      for (int ph = 0; ph < 4; ph++) {
        const float val = macc[i][ph];
        const bool gt = val > tv;
        tv = gt ? val : tv;
        ts = gt ? 32 + 4 * i + ph : ts;
      }
    }
#endif
    const int task = (ts & 16) ? ctask[1] : ctask[0];
    const int lt = task / (HT_T / 4), c4 = task - lt * (HT_T / 4);
    int y = 2 * (r0 + lt) + ((ts >> 3) & 1), x = 2 * (c0 + 4 * c4 + ((ts >> 1) & 3)) + (ts & 1);
#if OFX_HTC_MINI
    if (ts & 32) {
      const int i = (ts >> 2) & 3;
      const int mr = i == 0 ? mrow[0] : i == 1 ? mrow[1] : mrow[2], mc = i == 0 ? mcol[0] : i == 1 ? mcol[1] : mcol[2];
      y = 2 * (r0 + mr) + ((ts >> 1) & 1);
      x = 2 * (c0 + mc) + (ts & 1);
    }
#endif
    const unsigned k = (unsigned)(y * PS + x);
    if (tv > bestv || (tv == bestv && k < bestk)) { bestv = tv; bestk = k; }
  }
#if !OFX_HT_PIPE
  __syncthreads();  // facc / l2 are rewritten by the next tile
#endif
  }  // tile loop

  unsigned long long key = ((unsigned long long)ordered_f32(bestv) << 32) | (unsigned long long)(~bestk);
  if (bestk == 0xFFFFFFFFu) key = 0ull;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = __shfl_xor(key, o);
    key = other > key ? other : key;
  }
  if (lane == 0 && key) atomicMax(&p.best[s], key);
}

__device__ inline float unordered_f32(unsigned o) {  // inverse of ordered_f32
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}

__global__ void k_policy_finish(int S, const uint8_t *mask, const unsigned long long *best, int32_t *ipointer,
                                float *ptr_max) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S || (mask && !mask[s])) return;
  const unsigned k = ~(unsigned)(best[s] & 0xFFFFFFFFull);
  ipointer[2 * s] = (int)(k % PS);      // unravel_index(order='F') of a C-order flat index = (x, y)
  ipointer[2 * s + 1] = (int)(k / PS);  // qlearnIA_V2.py:218-220
  if (ptr_max) ptr_max[s] = unordered_f32((unsigned)(best[s] >> 32));  // np.max(ptr_prediction)
}

// QlearnIA.play packing (qlearnIA_V2.py:447-454): exactly one of shoot / thrust, pointer always set
__global__ void k_policy_actions(int S, const ofx_state st, const int32_t *iaction, const int32_t *ipointer,
                                 const uint8_t *mask, ofx_action *out) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S || (mask && !mask[s])) return;
  ofx_action a;
  a.valid = st.alive[s] ? 1 : 0;  // a dead ship's action is None (ship.py:260-262)
  a.shoot = iaction[s] == 0;
  a.thrust = iaction[s] == 1;
  a.px = ipointer[2 * s];
  a.py = ipointer[2 * s + 1];
  a._pad = 0;
  out[s] = a;
}

// ---- workspace -----------------------------------------------------------------------
struct PolicyWs {
  float *prep, *p1, *p2, *p3, *p4, *g1, *d1, *u0, *up1, *up2, *up3, *u2fr, *u3fr, *c4;
  unsigned long long *best;
  int32_t *iaction, *ipointer;
};

// v1 (layer-by-layer tail through a [S][8][200][200] HBM tensor) is kept for A/B checks
static bool policy_unfused() {
  const char *e = getenv("OFX_POLICY_UNFUSED");
  return e && e[0] == '1';
}

static size_t al(size_t b) { return (b + 255) & ~(size_t)255; }
constexpr int kDense1Chunks = 25;  // split-K of dense1: 5000 = 25 x 200

// N images (trunk runs), S policy samples (heads); the layout depends on both, so a forward with other sizes
// (ofx_policy_forward_obs) invalidates the results a previous forward left in the workspace
static int policy_workspace(ofx_handle *h, PolicyWs *ws, size_t N, size_t S) {
  const PrepLayout L = prep_layout();
  size_t f2, f3, f4;
  ofx_head_frame_bytes(S, &f2, &f3, &f4);
  const size_t sz[] = {al(4ull * L.total),          al(4ull * N * 8 * 200 * 200), al(4ull * N * 8 * 100 * 100),
                       al(4ull * N * 8 * 50 * 50),   al(4ull * N * 5000),          al(4ull * N * 100 * kDense1Chunks),
                       al(4ull * S * 100),           al(4ull * S * 625),           al(4ull * S * 2 * 50 * 50),
                       al(4ull * S * 4 * 100 * 100), al(policy_unfused() ? 4ull * S * 8 * 200 * 200 : 256), al(8ull * S),
                       al(4ull * S),                 al(8ull * S),                 al(f2), al(f3), al(f4)};
  size_t total = 0;
  for (size_t b : sz) total += b;
  int rc = ofx_ensure_scratch(h, total);
  if (rc) return rc;
  char *base = (char *)h->scratch;
  void **dst[] = {(void **)&ws->prep, (void **)&ws->p1, (void **)&ws->p2, (void **)&ws->p3, (void **)&ws->p4,
                  (void **)&ws->g1,   (void **)&ws->d1, (void **)&ws->u0, (void **)&ws->up1, (void **)&ws->up2,
                  (void **)&ws->up3,  (void **)&ws->best, (void **)&ws->iaction, (void **)&ws->ipointer,
                  (void **)&ws->u2fr, (void **)&ws->u3fr, (void **)&ws->c4};
  for (int i = 0; i < 17; i++) { *dst[i] = base; base += sz[i]; }
  return OFX_OK;
}

template <int CIN, int COUT, int TH, int TW, int MODE, bool POOL, bool OUT_HWC>
static int launch_conv(ofx_handle *h, ConvParams p, int images, int H) {
  p.H = H; p.W = H;
  p.tiles_x = H / TW;
  p.tiles = p.tiles_x * (H / TH);
  constexpr int NTB = ((TH / 2) * (TW / 2) + 63) / 64 * 64;
  hipLaunchKernelGGL((k_conv<CIN, COUT, TH, TW, MODE, POOL, OUT_HWC>), dim3((unsigned)(images * p.tiles)), dim3(NTB), 0,
                     h->stream, p);
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}

template <int CIN, int TH, int TW, int MODE, bool POOL, bool OUT_HWC>
static int launch_conv8(ofx_handle *h, ConvParams p, int images, int H) {
  p.H = H; p.W = H;
  p.tiles_x = H / TW;
  p.tiles = p.tiles_x * (H / TH);
  constexpr int NTB = ((TH / 2) * (TW / 4) + 63) / 64 * 64;
  hipLaunchKernelGGL((k_conv8<CIN, TH, TW, MODE, POOL, OUT_HWC>), dim3((unsigned)(images * p.tiles)), dim3(NTB), 0,
                     h->stream, p);
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}

static int launch_gemm(ofx_handle *h, const float *A, int lda, const float *B, int ldb, const float *bias, float *C,
                       int ldc, int M, int N, int K, int relu) {
  const int tiles = ((M + 31) / 32) * ((N + 31) / 32);
  hipLaunchKernelGGL(k_gemm_f32, dim3((tiles + 3) / 4), dim3(256), 0, h->stream, A, lda, B, ldb, bias, C, ldc, M, N, K,
                     relu);
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}

// The forward proper: N images (two 1-bit maps each: word bits?[img * bits_stride + w]) with M policy samples per
// image; vec8 = explicit observation heads [N*M][8] or null (the live state of the handle's arenas).
static int policy_forward_impl(ofx_handle *h, const float *weights, int N, int M, const unsigned *bits0,
                               const unsigned *bits1, size_t bits_stride, const float *vec8, const uint8_t *ship_mask,
                               float *act_values, int32_t *iaction, int32_t *ipointer, float *heatmap, float *ptr_max,
                               const int32_t *probe, float *ptr_probe) {
  const ofx_config &c = h->cfg;
  const int S = N * M;
  PolicyWs ws;
  int rc = policy_workspace(h, &ws, N, S);
  if (rc) return rc;
  int32_t off[64], cnt[64];
  policy_layout(off, cnt);
  const PrepLayout L = prep_layout();

  // 0. BN folding, phase weights, tables
  PrepParams pp;
  pp.w = weights; pp.prep = ws.prep;
  OFX_HIP(hipMemsetAsync(ws.prep + L.total - 64, 0, 64 * sizeof(float), h->stream));
  for (int i = 0; i < 4; i++) {
    pp.src_k[i] = off[6 * i]; pp.src_b[i] = off[6 * i + 1]; pp.src_g[i] = off[6 * i + 2];
    pp.cin[i] = kTrunkCin[i]; pp.cout[i] = 8; pp.dst_w[i] = L.tw[i]; pp.dst_b[i] = L.tb[i];
  }
  const int t_d1 = 24, t_d2 = 26, t_o1 = 28, t_ud = 30, t_up = 32, t_u4 = 50;
  for (int i = 0; i < 3; i++) {
    pp.src_k[4 + i] = off[t_up + 6 * i]; pp.src_b[4 + i] = off[t_up + 6 * i + 1]; pp.src_g[4 + i] = off[t_up + 6 * i + 2];
    pp.cin[4 + i] = kUpCin[i]; pp.cout[4 + i] = kUpCout[i]; pp.dst_w[4 + i] = L.uw[i]; pp.dst_b[4 + i] = L.ub[i];
  }
  pp.src_k4 = off[t_u4]; pp.src_b4 = off[t_u4 + 1];
  pp.dst_w4eff = L.w4eff; pp.dst_w4raw = L.w4raw; pp.dst_b4 = L.b4; pp.dst_efr = L.efr;
  pp.dst_w4eff_c = L.w4eff_c; pp.dst_w3mf = L.w3mf; pp.dst_w2mf = L.w2mf; pp.dst_w2fr = L.w2fr; pp.dst_w3fr = L.w3fr; pp.dst_lut1 = L.lut1;
  for (int i = 0; i < 3; i++) pp.dst_wbm[i] = L.wbm[i];
  for (int i = 0; i < 4; i++) pp.dst_bg[i] = L.bg[i];
  pp.phase = 0;
  hipLaunchKernelGGL(k_policy_prepare, dim3(32), dim3(256), 0, h->stream, pp);
  pp.phase = 1;
  hipLaunchKernelGGL(k_policy_prepare, dim3(4), dim3(256), 0, h->stream, pp);
  OFX_HIP(hipGetLastError());

  // 1. trunk, once per arena
  ConvParams cp;
  memset(&cp, 0, sizeof(cp));
  { const char *e = getenv("OFX_CONV_ABLATE"); cp.ablate = e ? atoi(e) : 0; }
  cp.bits[0] = bits0;
  cp.bits[1] = bits1;
  cp.bits_stride = bits_stride;
  // opt-in: pays off on sparse scenes only (measured: with policy-driven play the maps are full of lasers
  // and the window checks cost more than they save)
  const bool bgskip = getenv("OFX_POLICY_BG_SKIP") != nullptr;
  cp.w = ws.prep + L.tw[0]; cp.b = ws.prep + L.tb[0]; cp.out = ws.p1;
  if (bgskip) { cp.bg_in = ws.prep + L.total - 8; cp.bg_out = ws.prep + L.bg[0]; }  // the 8 pad floats are zero
  // TPW (tiles walked per workgroup with register prefetch) = 1: walking 5 or 25 tiles measured the same time (2.95-2.97
  // ms for conv2): the kernels are bound by their instruction streams, not by launch or staging latency; tile shapes
  // from an A/B on the chip (tools/ab_convm.sh)
  const bool trunk_valu = getenv("OFX_TRUNK_VALU") != nullptr;  // A/B: the pre-MFMA trunk kernels
  if (trunk_valu) rc = launch_conv8<2, 40, 100, 1, true, false>(h, cp, N, 400);
  else if (getenv("OFX_CONV1_MFMA")) rc = launch_convm<2, 8, 13, 1, false, 1>(h, cp, N, 400);  // A/B: the GEMM form
  else {
    cp.H = 400; cp.W = 400;
    hipLaunchKernelGGL(k_conv1_lut<40>, dim3((unsigned)(N * (400 / 40))), dim3(256), 0, h->stream, cp,
                       (const float *)(ws.prep + L.lut1));
    OFX_HIP(hipGetLastError());
  }
  if (rc) return rc;
  cp.in = ws.p1; cp.w = ws.prep + L.tw[1]; cp.b = ws.prep + L.tb[1]; cp.out = ws.p2; cp.wbm = ws.prep + L.wbm[0];
  if (bgskip) { cp.bg_in = ws.prep + L.bg[0]; cp.bg_out = ws.prep + L.bg[1]; }
  if (trunk_valu) rc = launch_conv<8, 8, 10, 100, 0, true, false>(h, cp, N, 200);
#if OFX_CONV2_SHAPE == 1
  else rc = launch_convm<8, 2, 7, 0, false, 1>(h, cp, N, 200);   // half-width tiles: more, shorter workgroups
#elif OFX_CONV2_SHAPE == 2
  else rc = launch_convm<8, 4, 7, 0, false, 1>(h, cp, N, 200);
#else
  else rc = launch_convm<8, 2, 13, 0, false, 1>(h, cp, N, 200);  // 29 KB of LDS: five workgroups per CU
#endif
  if (rc) return rc;
  cp.in = ws.p2; cp.w = ws.prep + L.tw[2]; cp.b = ws.prep + L.tb[2]; cp.out = ws.p3; cp.wbm = ws.prep + L.wbm[1];
  if (bgskip) { cp.bg_in = ws.prep + L.bg[1]; cp.bg_out = ws.prep + L.bg[2]; }
  if (trunk_valu) rc = launch_conv<8, 8, 10, 100, 0, true, false>(h, cp, N, 100);
  else rc = launch_convm<8, 4, 7, 0, false, 1>(h, cp, N, 100);
  if (rc) return rc;
  cp.in = ws.p3; cp.w = ws.prep + L.tw[3]; cp.b = ws.prep + L.tb[3]; cp.out = ws.p4; cp.wbm = ws.prep + L.wbm[2];
  cp.bg_in = nullptr; cp.bg_out = nullptr;
  if (trunk_valu) rc = launch_conv<8, 8, 10, 50, 0, true, true>(h, cp, N, 50);  // (h,w,c) = Flatten order
  else rc = launch_convm<8, 10, 4, 0, true, 1>(h, cp, N, 50);
  if (rc) return rc;

  // 2. dense1: trunk features on MFMA once per arena; head + head-1 per ship
  const float *k1 = weights + off[t_d1];
  {
    const int tiles = ((N + 31) / 32) * 4, jobs = tiles * kDense1Chunks;
    hipLaunchKernelGGL(k_gemm_f32_splitk, dim3((jobs + 3) / 4), dim3(256), 0, h->stream, ws.p4, 5000, k1 + 8 * 100, 100,
                       ws.g1, N, 100, 5000 / kDense1Chunks, tiles, jobs);
    OFX_HIP(hipGetLastError());
  }
  HeadParams hp;
  hp.N = N; hp.M = M; hp.st = h->st; hp.vec8 = vec8; hp.g1 = ws.g1; hp.g1_chunks = kDense1Chunks;
  hp.k1 = k1; hp.b1 = weights + off[t_d1 + 1];
  hp.k2 = weights + off[t_d2]; hp.b2 = weights + off[t_d2 + 1];
  hp.k3 = weights + off[t_o1]; hp.b3 = weights + off[t_o1 + 1];
  hp.mask = ship_mask; hp.d1 = ws.d1; hp.act = act_values; hp.iaction = iaction ? iaction : ws.iaction;
  hipLaunchKernelGGL(k_head_dense, dim3((S + 3) / 4), dim3(256), 0, h->stream, hp);
  OFX_HIP(hipGetLastError());

  // 3. head-2: updense1 on MFMA, then the up-convolutions per ship
  if ((rc = launch_gemm(h, ws.d1, 100, weights + off[t_ud], 625, weights + off[t_ud + 1], ws.u0, 625, S, 625, 100, 1)))
    return rc;
  ConvParams up;
  memset(&up, 0, sizeof(up));
  up.ablate = cp.ablate;
  up.mask = ship_mask;
  up.in = ws.u0; up.w = ws.prep + L.uw[0]; up.b = ws.prep + L.ub[0]; up.out = ws.up1;
  if ((rc = launch_conv<1, 2, 10, 50, 2, false, false>(h, up, S, 50))) return rc;
  up.in = ws.up1; up.w = ws.prep + L.uw[1]; up.b = ws.prep + L.ub[1]; up.out = ws.up2;
  // upconv2 is computed inside k_head_tail (its 24x24x4 patches are a 9 M-tile phase GEMM from uprelu1); the
  // stand-alone kernel only feeds the layer-by-layer A/B path
  if (policy_unfused())
    if ((rc = launch_conv<2, 4, 10, 100, 2, false, false>(h, up, S, 100))) return rc;
  OFX_HIP(hipMemsetAsync(ws.best, 0, sizeof(unsigned long long) * S, h->stream));
  if (!policy_unfused() && !getenv("OFX_HEAD_OLD")) {
    HeadParams2 hp2;
    hp2.S = S; hp2.up1 = ws.up1;
    hp2.w2mf = ws.prep + L.w2mf; hp2.b2 = ws.prep + L.ub[1]; hp2.w2raw = ws.prep + L.uw[1];
    hp2.w3mf = ws.prep + L.w3mf; hp2.b3 = ws.prep + L.ub[2]; hp2.w3raw = ws.prep + L.uw[2];
    hp2.w4eff_c = ws.prep + L.w4eff_c; hp2.b4 = ws.prep + L.b4; hp2.w4raw = ws.prep + L.w4raw;
    hp2.w2fr = ws.prep + L.w2fr; hp2.w3fr = ws.prep + L.w3fr; hp2.efr = ws.prep + L.efr;
    hp2.u2fr = ws.u2fr; hp2.u3fr = ws.u3fr; hp2.c4 = ws.c4;
    hp2.ablate = 0; hp2.dbg = nullptr; hp2.mask = ship_mask; hp2.best = ws.best; hp2.heat = heatmap; hp2.probe = probe; hp2.ptr_probe = probe ? ptr_probe : nullptr;
    const int pb = h->prof_base;
    if (pb >= 0 && (rc = ofx_event_record(h, pb))) return rc;
    if ((rc = ofx_launch_head(h, hp2))) return rc;
    if (pb >= 0) {
      if ((rc = ofx_event_record(h, pb + 1))) return rc;
      h->prof_base = pb + 2;
    }
  } else if (!policy_unfused()) {
    HeadTailParams ht;
    ht.up1 = ws.up1;
    ht.w2mf = ws.prep + L.w2mf; ht.w2raw = ws.prep + L.uw[1]; ht.b2 = ws.prep + L.ub[1];
    ht.w3mf = ws.prep + L.w3mf; ht.w3raw = ws.prep + L.uw[2]; ht.b3 = ws.prep + L.ub[2];
    ht.w4eff = ws.prep + L.w4eff_c; ht.w4raw = ws.prep + L.w4raw; ht.b4 = ws.prep + L.b4; ht.efr = ws.prep + L.efr;
    ht.mask = ship_mask; ht.best = ws.best; ht.heat = heatmap; ht.probe = probe; ht.ptr_probe = probe ? ptr_probe : nullptr;
    { const char *e = getenv("OFX_HT_ABLATE"); ht.ablate = e ? atoi(e) : 0; }
    const int pb = h->prof_base;
    if (pb >= 0 && (rc = ofx_event_record(h, pb))) return rc;
    hipLaunchKernelGGL(k_head_tail, dim3((unsigned)(S * (HT_S3 / HT_T))), dim3(256), 0, h->stream, ht);
    OFX_HIP(hipGetLastError());
    if (pb >= 0) {
      if ((rc = ofx_event_record(h, pb + 1))) return rc;
      h->prof_base = pb + 2;
    }
  } else {
    up.in = ws.up2; up.w = ws.prep + L.uw[2]; up.b = ws.prep + L.ub[2]; up.out = ws.up3;
    if ((rc = launch_conv<4, 8, 10, 100, 2, false, false>(h, up, S, 200))) return rc;
    Up4Params u4;
    u4.in = ws.up3; u4.weff = ws.prep + L.w4eff; u4.wraw = ws.prep + L.w4raw; u4.b4 = ws.prep + L.b4;
    u4.mask = ship_mask; u4.best = ws.best; u4.heat = heatmap;
    constexpr int tiles4 = (U4_LS / U4_TW) * (U4_LS / U4_TH);
    hipLaunchKernelGGL(k_upconv4, dim3((unsigned)(S * tiles4)), dim3(256), 0, h->stream, u4);
    OFX_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(k_policy_finish, dim3((S + 255) / 256), dim3(256), 0, h->stream, S, ship_mask, ws.best,
                     ipointer ? ipointer : ws.ipointer, ptr_max);
  OFX_HIP(hipGetLastError());
  (void)c;
  return OFX_OK;
}

extern "C" int ofx_policy_forward(ofx_handle *h, const float *weights, const uint8_t *ship_mask, float *act_values,
                                  int32_t *iaction, int32_t *ipointer, float *heatmap) {
  if (!h || !weights) { ofx_set_error("ofx_policy_forward: null argument"); return OFX_ERR_INVALID; }
  if (!h->spawned) { ofx_set_error("You must execute analyse_battleground first."); return OFX_ERR_STATE; }
  const ofx_config &c = h->cfg;
  if (c.width != PS || c.height != PS) {
    // Input((DEFAULT_WIDTH, DEFAULT_HEIGHT, 2)) is fixed at 400x400 (qlearnIA_V2.py:125)
    ofx_set_error("ofx_policy_forward: the pointer_model takes 400x400 maps (got %d x %d)", c.width, c.height);
    return OFX_ERR_INVALID;
  }
  OFX_HIP(hipSetDevice(c.device));
  int rc = ofx_launch_raster(h, OFX_MAP_BITS_LSB, nullptr, nullptr);  // the observation as 1-bit maps
  if (rc) return rc;
  return policy_forward_impl(h, weights, c.n_arenas, c.n_ships, (const unsigned *)h->maps[OFX_MAP_BITS_LSB][0],
                             (const unsigned *)h->maps[OFX_MAP_BITS_LSB][1], (size_t)(PS * PS) >> 5, nullptr, ship_mask,
                             act_values, iaction, ipointer, heatmap, nullptr, nullptr, nullptr);
}

extern "C" int ofx_policy_forward_obs(ofx_handle *h, const float *weights, int32_t n_obs, const void *bits,
                                      const float *vec8, float *act_values, int32_t *iaction, int32_t *ipointer,
                                      float *ptr_max, const int32_t *probe, float *ptr_probe) {
  if (!h || !weights || !bits || !vec8 || n_obs < 1) { ofx_set_error("ofx_policy_forward_obs: bad argument"); return OFX_ERR_INVALID; }
  if ((probe == nullptr) != (ptr_probe == nullptr)) { ofx_set_error("ofx_policy_forward_obs: pass probe and ptr_probe together"); return OFX_ERR_INVALID; }
  if (policy_unfused()) { ofx_set_error("ofx_policy_forward_obs: not available with OFX_POLICY_UNFUSED"); return OFX_ERR_STATE; }
  OFX_HIP(hipSetDevice(h->cfg.device));
  const size_t words = (size_t)(PS * PS) >> 5;
  return policy_forward_impl(h, weights, n_obs, 1, (const unsigned *)bits, (const unsigned *)bits + words, 2 * words, vec8,
                             nullptr, act_values, iaction, ipointer, nullptr, ptr_max, probe, ptr_probe);
}

// ---- TD targets of Trainer.replay (agents/qlearnIA_V2.py:251-270) -------------------------------------------------
__global__ void k_dqn_unpack(int n, const ofx_transition *rows, float *vec_prev, float *vec_next, int32_t *probe) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const ofx_transition r = rows[i];
  const bool pad = r.ship < 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    vec_prev[(size_t)i * 8 + k] = pad ? 0.f : r.head_prev[k];
    vec_next[(size_t)i * 8 + k] = pad ? 0.f : r.head_next[k];
  }
  probe[2 * i] = pad ? 0 : min(max(r.px, 0), PS - 1);
  probe[2 * i + 1] = pad ? 0 : min(max(r.py, 0), PS - 1);
}

__global__ void k_dqn_targets(int n, const ofx_transition *rows, float gamma, const float *act_prev, const float *probe_prev,
                              const float *act_next, const float *max_next, float *q_sa, float *p_sp, float *y_act,
                              float *y_ptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const ofx_transition r = rows[i];
  if (r.ship < 0) { q_sa[i] = p_sp[i] = y_act[i] = y_ptr[i] = 0.f; return; }
  const float live = r.done ? 0.f : 1.f;  // int(not done)
  q_sa[i] = act_prev[2 * i + (r.iaction ? 1 : 0)];
  p_sp[i] = probe_prev[i];
  y_act[i] = (float)r.reward + gamma * fmaxf(act_next[2 * i], act_next[2 * i + 1]) * live;  // np.max(prediction)
  y_ptr[i] = (float)r.reward + gamma * max_next[i] * live;                                    // np.max(ptr_prediction)
}

static int ensure_aux(ofx_handle *h, size_t bytes) {
  if (h->aux_bytes >= bytes) return OFX_OK;
  OFX_HIP(hipStreamSynchronize(h->stream));
  if (h->aux) (void)hipFree(h->aux);
  h->aux = nullptr; h->aux_bytes = 0;
  OFX_HIP(hipMalloc(&h->aux, bytes));
  h->aux_bytes = bytes;
  return OFX_OK;
}

extern "C" int ofx_policy_forward_obs(ofx_handle *h, const float *weights, int32_t n_obs, const void *bits,
                                      const float *vec8, float *act_values, int32_t *iaction, int32_t *ipointer,
                                      float *ptr_max, const int32_t *probe, float *ptr_probe);

extern "C" int ofx_dqn_targets(ofx_handle *h, const float *weights, int32_t n, const ofx_transition *rows,
                               const void *bits_prev, const void *bits_next, float gamma, float *q_sa, float *p_sp,
                               float *y_act, float *y_ptr) {
  if (!h || !weights || !rows || !bits_prev || !bits_next || !q_sa || !p_sp || !y_act || !y_ptr || n < 1) {
    ofx_set_error("ofx_dqn_targets: bad argument");
    return OFX_ERR_INVALID;
  }
  OFX_HIP(hipSetDevice(h->cfg.device));
  const size_t nn = (size_t)n;
  int rc = ensure_aux(h, al(32 * nn) * 2 + al(8 * nn) * 3 + al(4 * nn) * 2);
  if (rc) return rc;
  char *b = (char *)h->aux;
  float *vec_prev = (float *)b; b += al(32 * nn);
  float *vec_next = (float *)b; b += al(32 * nn);
  int32_t *probe = (int32_t *)b; b += al(8 * nn);
  float *act_prev = (float *)b; b += al(8 * nn);
  float *act_next = (float *)b; b += al(8 * nn);
  float *probe_prev = (float *)b; b += al(4 * nn);
  float *max_next = (float *)b;
  hipLaunchKernelGGL(k_dqn_unpack, dim3((n + 255) / 256), dim3(256), 0, h->stream, n, rows, vec_prev, vec_next, probe);
  OFX_HIP(hipGetLastError());
  if ((rc = ofx_policy_forward_obs(h, weights, n, bits_prev, vec_prev, act_prev, nullptr, nullptr, nullptr, probe, probe_prev)))
    return rc;
  if ((rc = ofx_policy_forward_obs(h, weights, n, bits_next, vec_next, act_next, nullptr, nullptr, max_next, nullptr, nullptr)))
    return rc;
  hipLaunchKernelGGL(k_dqn_targets, dim3((n + 255) / 256), dim3(256), 0, h->stream, n, rows, gamma, act_prev, probe_prev,
                     act_next, max_next, q_sa, p_sp, y_act, y_ptr);
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}

// (iaction, ipointer) of the last forward / explore, for the other translation units (ofx_replay.hip)
int ofx_policy_results(ofx_handle *h, int32_t **iaction, int32_t **ipointer) {
  PolicyWs ws;
  int rc = policy_workspace(h, &ws, h->cfg.n_arenas, (size_t)h->cfg.n_arenas * h->cfg.n_ships);
  if (rc) return rc;
  *iaction = ws.iaction;
  *ipointer = ws.ipointer;
  return OFX_OK;
}

#define OFX_STREAM_EXPLORE 2u
__global__ void k_policy_explore(int N, int M, int W, int H, int arena_base, double eps, uint32_t k0, uint32_t k1,
                                 uint32_t tick, int collecting, const uint8_t *mask, int32_t *iaction, int32_t *ipointer) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= N * M || (mask && !mask[s])) return;
  const int a = s / M, i = s - a * M;
  uint32_t r[4];
  ofx_philox4x32_10((uint32_t)(arena_base + a), (uint32_t)i, tick, OFX_STREAM_EXPLORE, k0, k1, r);
  const double u = (double)r[0] * (1.0 / 4294967296.0);
  if (collecting || u <= eps) {  // np.random.rand() <= epsilon (qlearnIA_V2.py:201)
    iaction[s] = ofx_draw_int(r[1], 1);
    ipointer[2 * s] = ofx_draw_int(r[2], W - 1);
    ipointer[2 * s + 1] = ofx_draw_int(r[3], H - 1);
  }
}

extern "C" int ofx_policy_explore(ofx_handle *h, double epsilon, uint64_t seed, uint32_t tick, int32_t collecting,
                                  const uint8_t *ship_mask, int32_t *iaction, int32_t *ipointer) {
  if (!h) { ofx_set_error("ofx_policy_explore: null handle"); return OFX_ERR_INVALID; }
  if (!h->spawned) { ofx_set_error("ofx_policy_explore before ofx_spawn"); return OFX_ERR_STATE; }
  if (epsilon < 0.0 || epsilon > 1.0) { ofx_set_error("Value must me in range [0,1]"); return OFX_ERR_INVALID; }  // epsilon.py:56
  OFX_HIP(hipSetDevice(h->cfg.device));
  if (!iaction || !ipointer) {
    PolicyWs ws;
    int rc = policy_workspace(h, &ws, h->cfg.n_arenas, (size_t)h->cfg.n_arenas * h->cfg.n_ships);
    if (rc) return rc;
    if (!iaction) iaction = ws.iaction;
    if (!ipointer) ipointer = ws.ipointer;
  }
  const int S = h->cfg.n_arenas * h->cfg.n_ships;
  hipLaunchKernelGGL(k_policy_explore, dim3((S + 255) / 256), dim3(256), 0, h->stream, h->cfg.n_arenas, h->cfg.n_ships,
                     h->cfg.width, h->cfg.height, h->cfg.arena_base, epsilon, (uint32_t)seed, (uint32_t)(seed >> 32), tick,
                     collecting, ship_mask, iaction, ipointer);
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}

extern "C" int ofx_policy_actions(ofx_handle *h, const int32_t *iaction, const int32_t *ipointer,
                                  const uint8_t *ship_mask, ofx_action *actions) {
  if (!h || !actions) { ofx_set_error("ofx_policy_actions: null argument"); return OFX_ERR_INVALID; }
  if (!h->spawned) { ofx_set_error("ofx_policy_actions before ofx_spawn"); return OFX_ERR_STATE; }
  OFX_HIP(hipSetDevice(h->cfg.device));
  const int S = h->cfg.n_arenas * h->cfg.n_ships;
  if (!iaction || !ipointer) {  // use the results the last ofx_policy_forward kept in the workspace
    PolicyWs ws;
    int rc = policy_workspace(h, &ws, h->cfg.n_arenas, (size_t)h->cfg.n_arenas * h->cfg.n_ships);
    if (rc) return rc;
    if (!iaction) iaction = ws.iaction;
    if (!ipointer) ipointer = ws.ipointer;
  }
  hipLaunchKernelGGL(k_policy_actions, dim3((S + 255) / 256), dim3(256), 0, h->stream, S, h->st, iaction, ipointer,
                     ship_mask, actions);
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}
