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
//           per ship: upconv2-4 + arg-max in the row-streaming kernel of ofx_head.hip (4-phase
//           low-resolution form, the (400,400) heat-map is only materialised on request).
//
// The convolutions run on the matrix cores (k_convm: banded GEMM; ofx_head.hip: phase-form GEMMs); conv1 reads the
// 1-bit maps through a 512-entry table (k_conv1_lut).  k_conv is the plain VALU convolution: upconv1 (1 -> 2 @ 50x50,
// bilinear up-sampling fused into its LDS staging) and, under OFX_OPT_TRUNK_PLAIN, the reference trunk of the
// agreement test.  BatchNorm is folded into the conv weights by k_policy_prepare.
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <utility>

#include "ofx_internal.h"
#include "ofx_head.h"
#include "ofx_diag.h"
#include "ofx_lowp.h"

#define PS 400 /* the model's fixed input side: Input((DEFAULT_WIDTH, DEFAULT_HEIGHT, 2)) */

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
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
  int w3mf;           // [half 2][k = tap*4 + ci (36)][n = phase*4 + co_local (16)]  MFMA B operand
  int w2mf;           // upconv2 in phase form: [k = tap*2 + ci (18, padded to 20)][n = phase*4 + co (16)]
  int w2fr;           // [variant top|bottom|left|right][20][16]: w2mf with the taps that fall into the zero padding of the
                      // variant's frame line dropped (k_head_frames: exact frame bands of uprelu2)
  int w3fr;           // [variant][k = tap*4 + ci (36)][n = parity*8 + co (16)]: upconv3 phase weights of one frame line
                      // (top / bottom: row phase fixed, n's parity = column phase; left / right the other way round)
  int w4eff_c;        // [8 ci][4 phases][9 taps] (fused kernel: one contiguous slice per input channel)
  int w4raw;          // [9][8]
  int efr;            // [line h|v][side first|last][parity 2][low-res offset 3][ci 8]: phase weights of the taps of
                      // upconv4 that fall into the zero padding of a frame pixel (k_head_tail border pass)
  int b4;             // [1]
  int zero16;         // [4] zeros (never written: the block is cleared when it is allocated)
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
  L.w3mf = off; off += 36 * 32;
  L.w2mf = off; off += 20 * 16;
  L.w2fr = off; off += 4 * 20 * 16;
  L.w3fr = off; off += 4 * 36 * 16;
  L.w4eff_c = off; off += 8 * 4 * 9;
  L.w4raw = off; off += 72;
  L.efr = off; off += 2 * 2 * 2 * 3 * 8;
  L.b4 = off; off += 1;
  off = (off + 3) & ~3;
  L.zero16 = off; off += 4;      // 16 zero bytes, 16-byte aligned: the source of the padding cells of k_conv3_stream's LDS-direct loads
  L.lut1 = off; off += 2 * 512 * 8;
  for (int i = 0; i < 3; i++) { L.wbm[i] = off; off += 24 * 64; }
  L.total = (off + 63) & ~63;
  return L;
}

struct PrepParams {
  const float *w;
  float *prep;
  int src_k[7], src_b[7], src_g[7], cin[7], cout[7], dst_w[7], dst_b[7];
  int src_k4, src_b4, dst_w4raw, dst_b4, dst_efr;
  int dst_w4eff_c, dst_w3mf, dst_w2mf, dst_w2fr, dst_w3fr, dst_lut1, dst_wbm[3];
  int phase;
  int legacy;                      // OFX_OPT_BILINEAR_LEGACY
};

// interpolation coefficients of the x2 bilinear up-sampling: output row 2i+a, conv tap dy in {-1,0,1} touches low-res
// rows i-1, i, i+1 with these weights.  Half-pixel centres (default): up[2i] = .25 L[i-1] + .75 L[i],
// up[2i+1] = .75 L[i] + .25 L[i+1].  Legacy (TF1 resize_bilinear, OFX_OPT_BILINEAR_LEGACY): up[2i] = L[i],
// up[2i+1] = .5 L[i] + .5 L[i+1].  Both clamp at the edge, so the phase form and its frame handling are the same.
__device__ inline float up_coef(int a, int dy, int t, int legacy) {
  // a=0: dy-1 -> row 2i-1, dy0 -> row 2i, dy+1 -> row 2i+1 ; a=1: rows 2i, 2i+1, 2i+2
  const float c0[3][3] = {{0.75f, 0.25f, 0.f}, {0.25f, 0.75f, 0.f}, {0.f, 0.75f, 0.25f}};
  const float c1[3][3] = {{0.25f, 0.75f, 0.f}, {0.f, 0.75f, 0.25f}, {0.f, 0.25f, 0.75f}};
  const float l0[3][3] = {{0.5f, 0.5f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.5f, 0.5f}};
  const float l1[3][3] = {{0.f, 1.f, 0.f}, {0.f, 0.5f, 0.5f}, {0.f, 0.f, 1.f}};
  if (legacy) return a ? l1[dy][t] : l0[dy][t];
  return a ? c1[dy][t] : c0[dy][t];
}

// Two launches: phase 0 (many workgroups) folds and builds every table that depends on the raw weights only; phase 1
// builds what needs the folded kernels (k_convm's per-lane B operands).
__global__ void k_policy_prepare(PrepParams p) {
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
        acc += p.w[p.src_k4 + (dy * 3 + dx) * 8 + ci] * (up_coef(a, dy, ty, p.legacy) * up_coef(b, dx, tx, p.legacy));
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
      acc += p.w[p.src_k4 + tap * 8 + ci] * up_coef(b, d, o, p.legacy);
    }
    p.prep[p.dst_efr + e] = acc;
  }
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
          acc += (p.w[p.src_k[6] + ((dy * 3 + dx) * cin + ci) * cout + co] * inv) * (up_coef(a, dy, ty, p.legacy) * up_coef(b, dx, tx, p.legacy));
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
            acc += (p.w[p.src_k[5] + ((dy * 3 + dx) * cin + ci) * cout + co] * inv) * (up_coef(a, dy, ty, p.legacy) * up_coef(b, dx, tx, p.legacy));
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
            if (!drop) acc += (p.w[p.src_k[5] + ((dy * 3 + dx) * 2 + ci) * 4 + co] * inv) * (up_coef(a, dy, ty, p.legacy) * up_coef(b, dx, tx, p.legacy));
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
          if (!drop) acc += (p.w[p.src_k[6] + ((dy * 3 + dx) * 4 + ci) * 8 + co] * inv) * (up_coef(a, dy, ty, p.legacy) * up_coef(b, dx, tx, p.legacy));
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
  }
}

// ---- generic direct 3x3 convolution ------------------------------------------------
struct ConvParams {
  const float *in;                 // MODE 0: planar [img][CIN][H][W]; k_upconv1: [img][625]
  const unsigned *bits[2];         // MODE 1: word bits[ci][img * bits_stride + w], LSB-first (ch0 ship, ch1 laser)
  size_t bits_stride;              // words between consecutive images (PS*PS/32, or twice that for interleaved maps)
  const float *w, *b;              // folded [9][CIN][COUT], [COUT]
  float *out;                      // planar [img][COUT][Ho][Wo] or HWC [img][Ho][Wo][COUT]
  const uint8_t *mask;             // per image, may be null
  const float *wbm;                // k_convm, CIN = 8: per-lane B operand [24][64] (PrepLayout::wbm)
  int H, W;                        // conv domain (input after any upsampling) = conv output size
  int tiles_x, tiles;              // tiles per row / per image
  int legacy;                      // k_upconv1: TF1 legacy source mapping (src = dst / 2) instead of half-pixel centres
  int images;                      // k_convm: number of images (the grid is padded to a multiple of 8 of them)
  const int32_t *live;             // k_upconv1 with a mask: ordered list of the selected images (live[0] = count), else null
  unsigned long long *stat;        // k_trunk12<0, true>: [4] M-tiles executed / all, table waves executed / all (may be null)
};

// MODE: 0 planar f32 input, 1 two 1-bit maps
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
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
          if (MODE == 0) {
            v = p.in[(((size_t)img * CIN + ci) * H + gy) * W + gx];
          } else if (MODE == 1) {
            const int cell = gy * W + gx;
            v = (float)((p.bits[ci][(size_t)img * p.bits_stride + (cell >> 5)] >> (cell & 31)) & 1u);
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

  // ---- epilogue: folded bias, ReLU, optional 2x2 max-pool ----
  const int oy = ty0 + 2 * tr, ox = tx0 + 2 * tc;
#pragma unroll
  for (int co = 0; co < COUT; co++) {
    const float bias = p.b[co];
    float o00 = fmaxf(acc[0][0][co] + bias, 0.f), o01 = fmaxf(acc[0][1][co] + bias, 0.f);
    float o10 = fmaxf(acc[1][0][co] + bias, 0.f), o11 = fmaxf(acc[1][1][co] + bias, 0.f);
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
// bf16 operand helpers of the opt-in OFX_OPT_POLICY_BF16 forms (k_convm, ts_gemm_phase_bf16)
// the lane's six B operands from wbm [24][64] (value for MFMA j of the fp32 form, lane l: k = l >> 4 -> ci = (l >> 4) + 4 (j & 1), tap j >> 1)
template <int LP>
__device__ __forceinline__ void ts_bw_lp(const float *wbm, int n16, int kq, lp_x4 (&bwb)[6]) {
#pragma unroll
  for (int J = 0; J < 6; J++) {
    const int j = 2 * (2 * J + (kq >> 1)) + (kq & 1);
    const float *q = wbm + j * 64 + n16;
    bwb[J] = lp_pk4<LP>(q[0], q[16], q[32], q[48]);
  }
}
// MODE 0: planar f32 input [img][CIN][H][W] (the only mode left; conv1 reads the bit maps through k_conv1_lut).  Output: planar [img][8][H/2][W/2] or
// (OUT_HWC) [img][H/2][W/2][8].  TH rows x 16 NG columns per workgroup, TH even, H % TH == 0; W is masked.
// A workgroup walks TPW consecutive tiles of one image: the weights are fetched once, the global loads of tile i+1 are
// in flight (in registers) while tile i computes, and the grid stays small (the dispatcher needs ~5 ns per workgroup:
// one workgroup per tile cost 1.6 ms of launch floor for conv2 alone).
template <int CIN, int TH, int NG, int MODE, bool OUT_HWC, int TPW, int LP = 0>
__global__ __launch_bounds__(256) void k_convm(ConvParams p) {
  constexpr bool BF16 = LP != 0;
  static_assert(!BF16 || CIN == 8, "the bf16 form packs the four channels of a k-quarter");
  constexpr int TW = 16 * NG, LS = TW + 8;  // LDS row: image column tx0 + c sits at index c + 4 (16-byte aligned interior),
                                            // the left / right halo columns at 3 and TW + 4
  constexpr int PLS = ((TH + 2) * LS + 63) / 64 * 64 + 16;  // plane stride = 16 mod 64: the 4 k-quarters hit different banks
  constexpr int NK = 3 * CIN;                                // MFMAs per M-tile (K = 12 CIN)
  constexpr int JOBS = NG * (TH / 2);
  constexpr int ROWS = CIN * (TH + 2);
  constexpr bool VEC = MODE == 0 && !OUT_HWC;                // rows of W floats are 16-byte aligned (W % 4 == 0)
  __shared__ __align__(16) float tile[CIN * PLS];
  const unsigned bid = blockIdx.x;
  // XCD-aware order: workgroups go round-robin over the 8 XCDs (each with its own L2), so workgroup b works on image
  // 8 (b / 8 / wpi) + b % 8: the tiles of one image run back to back on ONE XCD and the halo rows a tile shares with
  // its vertical neighbour come out of that L2 (conv2: FETCH_SIZE 10.6 -> 5.1 GB for a 5.24 GB input; same time - the
  // kernel is bound by its compute phase, 2.45 ms with the staging ablated, 1.19 ms with only the staging)
  const int wpi = p.tiles / TPW, j = (int)(bid >> 3);        // p.tiles % TPW == 0: all tiles of a workgroup share the image
  const int img = (j / wpi) * 8 + (int)(bid & 7u), t_first = (j % wpi) * TPW;
  if (img >= p.images) return;
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
  // OFX_OPT_POLICY_BF16: see ts_gemm_phase_bf16 - MFMA J, element i: tap 2 J + (kq >> 1), channel 4 (kq & 1) + i
  lp_x4 bwb[6];
  int toff[6];
  if constexpr (BF16) {
    ts_bw_lp<LP ? LP : 1>(p.wbm, n16, kq, bwb);
#pragma unroll
    for (int J = 0; J < 6; J++) {
      const int tap = 2 * J + (kq >> 1);
      toff[J] = (4 * (kq & 1) - kq) * PLS + (tap / 3) * LS + tap % 3;
    }
  }
  const int H2 = H >> 1, W2 = W >> 1;

  // ---- staging, split into fetch (global -> registers) and commit (registers -> LDS) ----
  constexpr int V4 = TW / 4, RPW = VEC ? (ROWS + 3) / 4 : 1;       // VEC: float4 per row, rows per wave
  static_assert(!VEC || V4 + 2 <= 64, "tile row wider than one wave");
  f32x4 vpre[RPW];  // native vectors: a float4 select is lowered to a pointer select + flat loads
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

  auto fetch = [&](int t) {
    const int ty0 = (t / p.tiles_x) * TH, tx0 = (t % p.tiles_x) * TW;
    if constexpr (VEC) {
      // a wave moves one tile row per step: lane i < V4 the i-th float4 of the interior, lanes V4 / V4+1 the float4
      // that holds the left / right halo column
      const int gxl = lane < V4 ? tx0 + 4 * lane : (lane == V4 ? tx0 - 4 : tx0 + TW);
      const bool colok = lane < V4 + 2 && gxl >= 0 && gxl < W;
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
    if constexpr (VEC) {
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
    } else {  // rows that are not a multiple of 16 bytes (the 50x50 layer: W even, W <= TW): 8-byte pieces, every load of
              // a thread in flight before the first LDS store; the halo columns are zeroed once (see below)
      constexpr int TOTAL = CIN * (TH + 2) * (TW / 2), SU = (TOTAL + 255) / 256;
      (void)tx0;  // one tile per image row (launch_convm checks)
      f32x2 vals[SU];
      const int W2c = W >> 1;
#pragma unroll
      for (int u = 0; u < SU; u++) {
        const int e = u * 256 + tid;
        const int j = e % (TW / 2), rr = (e / (TW / 2)) % (TH + 2), ci = e / ((TW / 2) * (TH + 2));
        const int gy = ty0 - 1 + rr;
        vals[u] = (f32x2){0.f, 0.f};
        if (e < TOTAL && j < W2c && gy >= 0 && gy < H)
          vals[u] = *reinterpret_cast<const f32x2 *>(p.in + (((size_t)img * CIN + ci) * H + gy) * W + 2 * j);
      }
#pragma unroll
      for (int u = 0; u < SU; u++) {
        const int e = u * 256 + tid;
        const int j = e % (TW / 2), rr = (e / (TW / 2)) % (TH + 2), ci = e / ((TW / 2) * (TH + 2));
        if (e < TOTAL && j <= W2c) *reinterpret_cast<f32x2 *>(&tile[ci * PLS + rr * LS + 4 + 2 * j]) = vals[u];
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

  auto lda = [&](const float *a, int j) -> float { return a[aof(j)]; };
  if constexpr (!VEC) {  // left halo column (image column -1): never written by the staging above
    for (int e = tid; e < CIN * (TH + 2); e += 256) tile[(e / (TH + 2)) * PLS + (e % (TH + 2)) * LS + 3] = 0.f;
  }
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
      if (r == 0 && px < W2) {
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
        if constexpr (BF16) {
#pragma unroll
          for (int J = 0; J < 6; J++) {
            const float *q0 = a0 + toff[J], *q1 = a1 + toff[J];
            const lp_x4 A0 = lp_pk4<LP ? LP : 1>(q0[0], q0[PLS], q0[2 * PLS], q0[3 * PLS]);
            const lp_x4 A1 = lp_pk4<LP ? LP : 1>(q1[0], q1[PLS], q1[2 * PLS], q1[3 * PLS]);
            d0 = lp_mfma16<LP ? LP : 1>(A0, bwb[J], d0);
            d1 = lp_mfma16<LP ? LP : 1>(A1, bwb[J], d1);
          }
        } else {
#pragma unroll
          for (int j = 0; j < NK; j++) {
            d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(lda(a0, j), bw[j], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(lda(a1, j), bw[j], d1, 0, 0, 0);
          }
        }
        finish(d0, g0, t0);
        finish(d1, g1, t1);
      } else {
        f32x4 d0 = binit;
        if constexpr (BF16) {
#pragma unroll
          for (int J = 0; J < 6; J++) {
            const float *q0 = a0 + toff[J];
            d0 = lp_mfma16<LP ? LP : 1>(lp_pk4<LP ? LP : 1>(q0[0], q0[PLS], q0[2 * PLS], q0[3 * PLS]), bwb[J], d0);
          }
        } else {
#pragma unroll
          for (int j = 0; j < NK; j++) d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(lda(a0, j), bw[j], d0, 0, 0, 0);
        }
        finish(d0, g0, t0);
      }
    }
    if (i + 1 < TPW) __syncthreads();  // the next commit overwrites the tile
  }
}

template <int CIN, int TH, int NG, int MODE, bool OUT_HWC, int TPW, int LP = 0>
static int launch_convm(ofx_handle *h, ConvParams p, int images, int H) {
  p.H = H; p.W = H;
  p.tiles_x = (H + 16 * NG - 1) / (16 * NG);
  p.tiles = p.tiles_x * (H / TH);
  if (p.tiles % TPW) { ofx_set_error("launch_convm: %d tiles per image not divisible by %d", p.tiles, TPW); return OFX_ERR_INVALID; }
  if (OUT_HWC && (H % 2 || H > 16 * NG)) { ofx_set_error("launch_convm: the 8-byte staging takes even rows of one tile width"); return OFX_ERR_INVALID; }
  p.images = images;
  hipLaunchKernelGGL((k_convm<CIN, TH, NG, MODE, OUT_HWC, TPW, LP>), dim3((unsigned)((images + 7) / 8 * 8 * (p.tiles / TPW))), dim3(256), 0,
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
#pragma unroll
    for (int co = 0; co < 8; co++) *reinterpret_cast<f32x4 *>(obase + (size_t)co * (H / 2) * W2 + off) = m4[co];
  }
}

// ---- upconv1 (1 -> 2 channels @ 50x50 behind a x2 bilinear up-sampling of the 25x25 dense output) --------------------
// One workgroup per policy sample: the 625 inputs and the zero-padded up-sampled plane live in LDS (the generic k_conv
// gathered every staged cell from global memory with four loads: 0.26 ms for 32768 ships; this form 0.22, bound by its
// 0.66 GB of output).  Up-sample x first, then y (the restatement's order), taps in (dy, dx) order with fma, bias
// behind the sum, ReLU.  A thread owns 10 adjacent pixels of a row.
__global__ __launch_bounds__(256) void k_upconv1(ConvParams p) {
  __shared__ float u0s[625];
  __shared__ float up[52][53];   // up-sampled plane with its zero frame; pitch 53: the 5 segments of a row start on different banks
  const int tid = threadIdx.x;
  // work item it = the it-th ship (of the ordered list `live` with a mask): a bounded grid walks the items, so a
  // masked launch pays for its selected ships only (32768 workgroups that exit at once cost 0.13 ms)
  const int n_items = p.live ? p.live[0] : p.images;
#pragma unroll 1
  for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
  const int img = p.live ? p.live[1 + it] : it;
  __syncthreads();  // the previous item's readers of u0s / up are done
  for (int e = tid; e < 625; e += 256) u0s[e] = p.in[(size_t)img * 625 + e];
  __syncthreads();
  for (int e = tid; e < 52 * 52; e += 256) {
    const int c = e % 52, r = e / 52, gy = r - 1, gx = c - 1;
    float v = 0.f;
    if (gy >= 0 && gy < 50 && gx >= 0 && gx < 50) {
      const float sy = p.legacy ? (float)gy * 0.5f : ((float)gy + 0.5f) * 0.5f - 0.5f;
      const float sx = p.legacy ? (float)gx * 0.5f : ((float)gx + 0.5f) * 0.5f - 0.5f;
      const float fy = floorf(sy), fx = floorf(sx);
      const float ly = sy - fy, lx = sx - fx;
      int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
      y0 = max(y0, 0); x0 = max(x0, 0); y1 = min(y1, 24); x1 = min(x1, 24);
      const float a = u0s[y0 * 25 + x0], b = u0s[y0 * 25 + x1], d = u0s[y1 * 25 + x0], g = u0s[y1 * 25 + x1];
      const float top = a + (b - a) * lx, bot = d + (g - d) * lx;
      v = top + (bot - top) * ly;
    }
    up[r][c] = v;
  }
  __syncthreads();
  if (tid < 250) {
  const int row = tid / 5, x0 = 10 * (tid - 5 * row);
  float acc[2][10];
#pragma unroll
  for (int co = 0; co < 2; co++)
#pragma unroll
    for (int i = 0; i < 10; i++) acc[co][i] = 0.f;
#pragma unroll
  for (int dy = 0; dy < 3; dy++) {
    float v[12];
#pragma unroll
    for (int i = 0; i < 12; i++) v[i] = up[row + dy][x0 + i];
#pragma unroll
    for (int dx = 0; dx < 3; dx++)
#pragma unroll
      for (int co = 0; co < 2; co++) {
        const float wv = p.w[(dy * 3 + dx) * 2 + co];  // uniform -> scalar load
#pragma unroll
        for (int i = 0; i < 10; i++) acc[co][i] = __builtin_fmaf(v[i + dx], wv, acc[co][i]);
      }
  }
#pragma unroll
  for (int co = 0; co < 2; co++) {
    const float bias = p.b[co];
    float *o = p.out + (((size_t)img * 2 + co) * 50 + row) * 50 + x0;
#pragma unroll
    for (int i = 0; i < 10; i += 2)
      *reinterpret_cast<float2 *>(o + i) = make_float2(fmaxf(acc[co][i] + bias, 0.f), fmaxf(acc[co][i + 1] + bias, 0.f));
  }
  }  // tid < 250
  }  // work items
}

// ---- streaming trunk kernel (k_trunk12): the GEMM phase ---------------------------------------------
// k_convm's banded GEMM over an LDS tile of RP row pairs x WD columns (planes PLS apart, rows LS apart; abase = the
// lane's k-quarter plane at column -1 of tile row 0).  M-tiles of 16 pixels are enumerated FLAT over (row pair, column):
// RP * WD / 16 of them, no masked columns; wave w takes tiles w, w + 16, w + 32, w + 48 as four interleaved accumulator
// chains (a wave with three: a pair and a single).  Epilogue as k_convm: 2x2 pool + ReLU, one 8-byte store per lane to
// orow = the lane's channel plane at the step's first pooled row (planar [.][WD/2]).
template <int WD, int RP, int LS, int PLS>
__device__ __forceinline__ void ts_gemm_phase(const float *abase, const float (&bw)[24], const f32x4 binit, int wv, int n16,
                                              int kq, int r, float *orow) {
  constexpr int NK = 24, NPX = RP * WD, NT = (NPX + 15) / 16;
  static_assert(WD % 4 == 0 && NT <= 64, "four M-tiles per wave at most");
  auto aof = [&](int j) -> int { return (4 * (j & 1)) * PLS + ((j >> 1) / 3) * LS + ((j >> 1) % 3); };
  auto finish = [&](const f32x4 d, int T) {
    float q0, q1;
    q0 = max_raw(max_raw(d[0], 0.f), d[1]);  // compiler-visible reads of the MFMA result (see max_raw)
    q1 = max_raw(max_raw(d[2], 0.f), d[3]);
    q0 = max_raw(q0, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, q0), 0xB1, 0xF, 0xF, true)));
    q1 = max_raw(q1, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, q1), 0xB1, 0xF, 0xF, true)));
    const int P = 16 * T + 4 * kq;                                 // the lane's four pixels P .. P + 3 of one row pair
    const int rp = P / WD, x = P - rp * WD;
    if (r == 0 && P < NPX) *reinterpret_cast<float2 *>(orow + rp * (WD / 2) + (x >> 1)) = make_float2(q0, q1);
  };
  auto a_of_tile = [&](int T) -> const float * {                   // the lane's A row: pixel 16 T + n16 (clamped past the end)
    const int P = min(16 * T + n16, NPX - 1);
    const int rp = P / WD, x = P - rp * WD;
    return abase + 2 * rp * LS + x;
  };
  if (wv + 48 < NT) {  // all four M-tiles of the wave at once (k_trunk12: 3.0 -> 2.84 ms against two pairs)
    const float *a0 = a_of_tile(wv), *a1 = a_of_tile(wv + 16), *a2 = a_of_tile(wv + 32), *a3 = a_of_tile(wv + 48);
    f32x4 d0 = binit, d1 = binit, d2 = binit, d3 = binit;
#pragma unroll
    for (int j = 0; j < NK; j++) {
      d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[aof(j)], bw[j], d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[aof(j)], bw[j], d1, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[aof(j)], bw[j], d2, 0, 0, 0);
      d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3[aof(j)], bw[j], d3, 0, 0, 0);
    }
    finish(d0, wv); finish(d1, wv + 16); finish(d2, wv + 32); finish(d3, wv + 48);
  } else {
#pragma unroll 1
    for (int T = wv; T < NT; T += 32) {
      const int T1 = T + 16;
      const float *a0 = a_of_tile(T);
      if (T1 < NT) {  // wave-uniform
        const float *a1 = a_of_tile(T1);
        f32x4 d0 = binit, d1 = binit;
#pragma unroll
        for (int j = 0; j < NK; j++) {
          d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[aof(j)], bw[j], d0, 0, 0, 0);
          d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[aof(j)], bw[j], d1, 0, 0, 0);
        }
        finish(d0, T);
        finish(d1, T1);
      } else {
        f32x4 d0 = binit;
#pragma unroll
        for (int j = 0; j < NK; j++) d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[aof(j)], bw[j], d0, 0, 0, 0);
        finish(d0, T);
      }
    }
  }
}

// OFX_OPT_TRUNK_SPARSE (opt-in, exact): the same GEMM phase, but an M-tile whose whole input window - 4 tile rows x 18
// columns - holds the layer's CONSTANT input (conv1 of an empty neighbourhood: every channel plane at K1[ci]) and touches
// no zero padding stores the constant K2[co] the dense MFMA sequence gives for such a window (computed by that very
// sequence once per workgroup: the same bits) instead of running its 24 MFMAs.  nz: per tile row 8 words, bit x set
// <-> column x of that row is NOT known to be constant (phase A).  An M-tile that crosses from one row pair into the
// next contains columns 199 and 0, i.e. touches the padding: only tiles inside one row pair with 1 <= x, x + 15 <= 198
// can be constant.  top / bottom: tile row 0 / the last tile row is a padding row of the image.
// LP != 0: the same with the 16-bit operand sequence of ts_gemm_phase_bf16 (K2 then comes from THAT sequence).
template <int WD, int RP, int LS, int PLS, int LP>
__device__ __forceinline__ void ts_gemm_phase_sparse(const float *abase, const float (&bw)[LP ? 1 : 24], const lp_x4 (&bwb)[6],
                                                     const f32x4 binit, int wv, int lane,
                                                     int n16, int kq, int r, float *orow, const unsigned *nz, float k2,
                                                     bool top, bool bottom, unsigned &n_exec, unsigned &n_all,
                                                     unsigned long long *dbg = nullptr) {
#if OFX_TRUNK_STAMPS
  unsigned long long g_last = __builtin_amdgcn_s_memtime();
#define TSG_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); dbg[i] += t_ - g_last; g_last = t_; } while (0)
#else
#define TSG_STAMP(i) do { } while (0)
#endif
  constexpr int NK = 24, NPX = RP * WD, NT = (NPX + 15) / 16;
  auto aof = [&](int j) -> int { return (4 * (j & 1)) * PLS + ((j >> 1) / 3) * LS + ((j >> 1) % 3); };
  int toff[6];  // LP: abase carries the fp32 form's + kq * PLS: taken out again (ts_gemm_phase_bf16)
#pragma unroll
  for (int J = 0; J < 6; J++) {
    const int tap = 2 * J + (kq >> 1);
    toff[J] = (4 * (kq & 1) - kq) * PLS + (tap / 3) * LS + tap % 3;
  }
  // the M-tile's matrix sequence: one or two accumulator chains
  auto mm1 = [&](const float *a0, f32x4 &d0) {
    if constexpr (LP != 0) {
#pragma unroll
      for (int J = 0; J < 6; J++) {
        const float *q0 = a0 + toff[J];
        d0 = lp_mfma16<LP ? LP : 1>(lp_pk4<LP ? LP : 1>(q0[0], q0[PLS], q0[2 * PLS], q0[3 * PLS]), bwb[J], d0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NK; j++) d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[aof(j)], bw[LP ? 0 : j], d0, 0, 0, 0);
    }
  };
  auto mm2 = [&](const float *a0, const float *a1, f32x4 &d0, f32x4 &d1) {
    if constexpr (LP != 0) {
#pragma unroll
      for (int J = 0; J < 6; J++) {
        const float *q0 = a0 + toff[J], *q1 = a1 + toff[J];
        const lp_x4 A0 = lp_pk4<LP ? LP : 1>(q0[0], q0[PLS], q0[2 * PLS], q0[3 * PLS]);
        const lp_x4 A1 = lp_pk4<LP ? LP : 1>(q1[0], q1[PLS], q1[2 * PLS], q1[3 * PLS]);
        d0 = lp_mfma16<LP ? LP : 1>(A0, bwb[J], d0);
        d1 = lp_mfma16<LP ? LP : 1>(A1, bwb[J], d1);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NK; j++) {
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[aof(j)], bw[LP ? 0 : j], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[aof(j)], bw[LP ? 0 : j], d1, 0, 0, 0);
      }
    }
  };
  auto finish = [&](const f32x4 d, int T) {
    float q0, q1;
    q0 = max_raw(max_raw(d[0], 0.f), d[1]);
    q1 = max_raw(max_raw(d[2], 0.f), d[3]);
    q0 = max_raw(q0, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, q0), 0xB1, 0xF, 0xF, true)));
    q1 = max_raw(q1, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, q1), 0xB1, 0xF, 0xF, true)));
    const int P = 16 * T + 4 * kq;
    const int rp = P / WD, x = P - rp * WD;
    if (r == 0 && P < NPX) *reinterpret_cast<float2 *>(orow + rp * (WD / 2) + (x >> 1)) = make_float2(q0, q1);
  };
  auto a_of_tile = [&](int T) -> const float * {
    const int P = min(16 * T + n16, NPX - 1);
    const int rp = P / WD, x = P - rp * WD;
    return abase + 2 * rp * LS + x;
  };
  // Every wave classifies ALL the step's M-tiles by itself, lane T tile T: no list to build, no barrier.  Bit T of the
  // mask: tile T has to run.  (A first version let a wave test its own four tiles one after the other with eight lanes each
  // and appended to a shared list behind a barrier: 4 500 cycles per step for the classification alone, stamps.)
  unsigned long long run_mask;
  {
    const int T = lane, P = 16 * T, rp = P / WD, x = P - rp * WD;
    bool run = T < NT;
    if (run && !(x < 1 || x + 15 > WD - 2 || P + 15 >= NPX || (top && rp == 0) || (bottom && rp == RP - 1))) {
      // window = bits x - 1 .. x + 16 of tile rows 2 rp .. 2 rp + 3: two words per row
      const int w0 = (x - 1) >> 5, lo = (x - 1) & 31;
      const unsigned m0 = 0x3FFFFu << lo, m1 = lo + 18 > 32 ? (1u << (lo + 18 - 32)) - 1u : 0u;
      unsigned any = 0u;
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        const unsigned *rw = nz + (2 * rp + rr) * 8 + w0;
        any |= (rw[0] & m0) | (rw[1] & m1);
      }
      run = any != 0u;
    }
    run_mask = __builtin_amdgcn_ballot_w64(run);
  }
  auto store_const = [&](int T) {                                          // the constant tile: the dense result, stored
    const int P = 16 * T + 4 * kq, rp = P / WD, x = P - rp * WD;
    if (r == 0) *reinterpret_cast<float2 *>(orow + rp * (WD / 2) + (x >> 1)) = make_float2(k2, k2);
  };
  auto run1 = [&](int T0) {
    f32x4 d0 = binit;
    mm1(a_of_tile(T0), d0);
    finish(d0, T0);
  };
  // the wave's own four tiles: the constant ones are stored; then the RUN tiles are dealt out over all 16 waves - wave w
  // takes the w-th, (w + 16)-th, ... set bit of the mask (a tile's result does not depend on who computes it)
#pragma unroll 1
  for (int T = wv; T < NT; T += 16)
    if (!((run_mask >> T) & 1ull)) store_const(T);
  if (wv == 0) { n_all += (unsigned)NT; n_exec += (unsigned)__builtin_popcountll(run_mask); }
  TSG_STAMP(0);
  unsigned long long m = run_mask;
  for (int i = 0; i < wv && m; i++) m &= m - 1;                             // skip the bits of the waves in front
#pragma unroll 1
  while (m) {
    const int T0 = __builtin_ctzll(m);
    unsigned long long m2 = m;
    for (int i = 0; i < 16 && m2; i++) m2 &= m2 - 1;                        // 16 set bits further on: this wave's next tile
    if (m2) {                                                               // two accumulator chains
      const int T1 = __builtin_ctzll(m2);
      f32x4 d0 = binit, d1 = binit;
      mm2(a_of_tile(T0), a_of_tile(T1), d0, d1);
      finish(d0, T0);
      finish(d1, T1);
      for (int i = 0; i < 16 && m2; i++) m2 &= m2 - 1;
    } else run1(T0);
    m = m2;
  }
  TSG_STAMP(2);
}

// OFX_OPT_POLICY_BF16 (opt-in): the same banded GEMM on v_mfma_f32_16x16x16_bf16 - K = 96 as 6 MFMAs of K = 16 instead
// of 24 of K = 4.  MFMA J, lane (pixel n16, k-quarter kq), element i: k <-> (tap = 2 J + (kq >> 1), ci = 4 (kq & 1) + i):
// the A operand is the four channel planes 4 (kq & 1) .. + 3 at the tap's (row, dx) of the fp32 LDS tile, rounded to
// bf16 on the way in (one address per J: lane-constant tap offset, planes by immediate); the B operand is packed once
// from the same PrepLayout::wbm the fp32 kernel uses.  fp32 accumulation, same epilogue.
template <int WD, int RP, int LS, int PLS, int LP>
__device__ __forceinline__ void ts_gemm_phase_bf16(const float *abase, const lp_x4 (&bwb)[6], const f32x4 binit, int wv, int n16,
                                                   int kq, int r, float *orow) {
  constexpr int NPX = RP * WD, NT = (NPX + 15) / 16;
  static_assert(WD % 4 == 0 && NT <= 64, "four M-tiles per wave at most");
  int toff[6];  // abase carries the fp32 form's + kq * PLS: taken out again
#pragma unroll
  for (int J = 0; J < 6; J++) {
    const int tap = 2 * J + (kq >> 1);
    toff[J] = (4 * (kq & 1) - kq) * PLS + (tap / 3) * LS + tap % 3;
  }
  auto finish = [&](const f32x4 d, int T) {
    float q0, q1;
    q0 = max_raw(max_raw(d[0], 0.f), d[1]);
    q1 = max_raw(max_raw(d[2], 0.f), d[3]);
    q0 = max_raw(q0, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, q0), 0xB1, 0xF, 0xF, true)));
    q1 = max_raw(q1, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, q1), 0xB1, 0xF, 0xF, true)));
    const int P = 16 * T + 4 * kq;
    const int rp = P / WD, x = P - rp * WD;
    if (r == 0 && P < NPX) *reinterpret_cast<float2 *>(orow + rp * (WD / 2) + (x >> 1)) = make_float2(q0, q1);
  };
  auto a_of_tile = [&](int T) -> const float * {
    const int P = min(16 * T + n16, NPX - 1);
    const int rp = P / WD, x = P - rp * WD;
    return abase + 2 * rp * LS + x;
  };
#pragma unroll 1
  for (int T = wv; T < NT; T += 32) {  // two M-tiles at a time: two accumulator chains
    const int T1 = T + 16;
    const bool two = T1 < NT;          // wave-uniform
    const float *a0 = a_of_tile(T), *a1 = a_of_tile(two ? T1 : T);
    f32x4 d0 = binit, d1 = binit;
#pragma unroll
    for (int J = 0; J < 6; J++) {
      const float *q0 = a0 + toff[J], *q1 = a1 + toff[J];
      const lp_x4 A0 = lp_pk4<LP>(q0[0], q0[PLS], q0[2 * PLS], q0[3 * PLS]);
      const lp_x4 A1 = lp_pk4<LP>(q1[0], q1[PLS], q1[2 * PLS], q1[3 * PLS]);
      d0 = lp_mfma16<LP>(A0, bwb[J], d0);
      d1 = lp_mfma16<LP>(A1, bwb[J], d1);
    }
    finish(d0, T);
    if (two) finish(d1, T1);
  }
}

// ---- conv1 -> conv2 fused: the 5.2 GB pooled conv1 activation never exists -----------------------------------------
// One 1024-thread workgroup walks ONE image top to bottom in 20 steps of F12_TH = 10 conv2 rows.  Per step:
//   phase A (all threads, VALU + LDS): the table form of conv1 (see k_conv1_lut) for the 10 NEW rows of p1 = pool(relu(
//            bn(conv1))) - a thread owns two adjacent p1 pixels, 1000 of the 1024 threads busy - written straight into
//            the planar LDS tile the GEMM reads (rows R0 - 1 .. R0 + 10 of p1, zero outside the image);
//   phase B (MFMA): conv2 as k_convm's banded GEMM on that tile, M-tiles of 16 pixels enumerated FLAT over the 5 row
//            pairs x 200 columns (62.5 M-tiles: no masked columns), 4 per wave as four accumulator chains; the global loads of
//            the next step's bit rows are in flight meanwhile;
//            (the step's last two p1 rows are also kept aside: they are the first two tile rows of the next step).
// Separating the phases in time costs little as long as each phase has its four waves per SIMD busy; one workgroup per CU.
// (r03: a form with two half-workgroups in anti-phase, 8 table waves beside 8 GEMM waves on two images, took 3.19 ms
// against 2.83 ms - measured on the chip and removed again, DESIGN.md section 3.  MFMA and VALU instructions share one
// issue port per SIMD and a matrix wave with MFMAs queued starves the vector instructions of the waves beside it
// (tools/ubench_coexec.hip, fixed-window form); the table phase is also one chain of LDS latencies of ~8 400 cycles
// per WAVE whatever the number of waves, so halving the table waves halves the table throughput.)
// Output = k_convm's: planar [img][8][100][100].
constexpr int F12_TH = 10, F12_LS = 216, F12_ROWS = F12_TH + 2;
constexpr int F12_PLS = (F12_ROWS * F12_LS + 63) / 64 * 64 + 16;  // plane stride = 16 mod 64, as k_convm
constexpr int F12_WR = 14, F12_BR = 2 * (F12_TH + 1) + 2;         // words per staged bit row; bit rows of the first step
constexpr int F12_THREADS = 1024;
static_assert(F12_PLS % 64 == 16 && F12_PLS % 4 == 0 && F12_LS % 4 == 0, "tile layout");
static_assert(2 * F12_BR * F12_WR <= F12_THREADS, "one staged word per thread");
static_assert((F12_TH + 1) * 100 <= 2 * F12_THREADS && F12_TH * 100 <= F12_THREADS, "pixel pairs per step");

// SPARSE (OFX_OPT_TRUNK_SPARSE, opt-in, fp32 only): the two input planes are ~1 % set bits (lib/observation.py:79-95), so
// most of conv1's output is ONE value per channel - K1[c] = relu(bn(conv1(empty window))), table pattern 0 - and most
// of conv2's M-tiles multiply that constant.  Exact: a wave whose 64 pixel pairs all see empty 4 x 6 bit windows writes
// K1 instead of reading its 16 table rows per pixel (the same sum of the same two table entries), every other pixel
// pair marks its two columns in a per-row bit map, and the GEMM phase skips M-tiles whose whole window is unmarked and
// away from the padding (ts_gemm_phase_sparse).  Bit-identical to the dense kernel (tests/test_gpu_policy.py).
template <int LP, bool SPARSE = false>
__global__ __launch_bounds__(F12_THREADS) void k_trunk12(ConvParams p, const float *lut) {
  constexpr bool BF16 = LP != 0;
  constexpr int W = PS, H = PS, H1 = PS / 2, H2 = PS / 4, LS = F12_LS, PLS = F12_PLS, NK = 24;
  __shared__ __align__(16) float slut[2 * 512 * 8];
  __shared__ __align__(16) float tile[8 * F12_PLS];
  __shared__ unsigned rows[2][2][F12_BR][F12_WR];  // [buffer][channel][bit row][word]
  __shared__ __align__(16) float halo[2][8][2][F12_LS];  // the last two p1 rows of a step = the first two of the next
  __shared__ unsigned nz[SPARSE ? 2 : 1][F12_ROWS][8];   // SPARSE: [buffer][tile row]: bit x <-> column x may differ from K1
  __shared__ __align__(16) float k1s[8], k2s[8];         // SPARSE: the constants of an empty neighbourhood
  // SPARSE (fp32): the 24 B operands of a lane are re-read from here in every phase B instead of living in registers
  // through phase A - with them the loop spilled, and every spill reload is an s_waitcnt vmcnt(0), i.e. a wait for the bit
  // rows in flight and for the stores' acknowledgements (stamps: 3 400 of a step's 13 700 cycles in front of phase A's
  // barrier, 5 000 behind the last M-tile)
  __shared__ float wbs[(SPARSE && !BF16) ? 24 * 64 : 1];
  unsigned n_exec = 0, n_all = 0, t_exec = 0, t_all = 0; // SPARSE: the wave's counts (M-tiles run / all, table passes run / all)
#if OFX_TRUNK_STAMPS
  unsigned long long st_a = 0, st_b = 0, st_last = __builtin_amdgcn_s_memtime(), st_begin = st_last, st_g[3] = {0, 0, 0};
#define T12_STAMP(acc) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - st_last; st_last = t_; } while (0)
#else
#define T12_STAMP(acc) do { } while (0)
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kq = lane >> 4, co = n16 >> 1, r = n16 & 1;

  // bit rows of the p1 rows [pa, pb): image rows 2 pa - 1 .. 2 pb, re-aligned like k_conv1_lut (bit x + 1 of a staged
  // row <-> image column x; bit 0 and the bits past column W - 1 are the zero padding).  One word per thread.
  auto bits_fetch = [&](int img, int pa, int pb) -> unsigned {
    const int nrows = 2 * (pb - pa) + 2;
    unsigned out = 0u;
    if (tid < 2 * nrows * F12_WR) {
      const int w = tid % F12_WR, rr = (tid / F12_WR) % nrows, ci = tid / (F12_WR * nrows);
      const int gy = 2 * pa - 1 + rr;
      if (gy >= 0 && gy < H) {
        const long long s0 = (long long)gy * W + 32 * w - 1;      // cell of output bit 0 (-1 only for gy = 0, w = 0)
        const unsigned *bits = p.bits[ci] + (size_t)img * p.bits_stride;
        const long long sw = s0 >> 5;                             // arithmetic shift: -1 -> word -1
        const unsigned lo = (sw >= 0 && sw < (PS * PS) >> 5) ? bits[sw] : 0u;
        const unsigned hi = (sw + 1 < (PS * PS) >> 5) ? bits[sw + 1] : 0u;
        out = __funnelshift_r(lo, hi, (unsigned)(s0 & 31));
        const int xlo = 32 * w - 1;
        if (xlo < 0) out &= ~1u;
        const int over = xlo + 32 - W;                            // bits past the last column
        if (over > 0) out = over >= 32 ? 0u : (out & (0xFFFFFFFFu >> over));
      }
    }
    return out;
  };
  auto bits_commit = [&](int buf, int pa, int pb, unsigned word) {
    const int nrows = 2 * (pb - pa) + 2;
    if (tid < 2 * nrows * F12_WR) {
      const int w = tid % F12_WR, rr = (tid / F12_WR) % nrows, ci = tid / (F12_WR * nrows);
      rows[buf][ci][rr][w] = word;
    }
  };

  // ---- prologue: table, weights, zeroed tile (halo columns and the row above the image stay zero), first bit rows ----
  for (int e = tid; e < 2 * 512 * 8 / 4; e += F12_THREADS)
    reinterpret_cast<f32x4 *>(slut)[e] = reinterpret_cast<const f32x4 *>(lut)[e];
  for (int e = tid; e < 8 * PLS / 4; e += F12_THREADS) reinterpret_cast<f32x4 *>(tile)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int e = tid; e < 2 * 8 * 2 * LS / 4; e += F12_THREADS) reinterpret_cast<f32x4 *>(&halo[0][0][0][0])[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bw[(BF16 || SPARSE) ? 1 : NK];
  lp_x4 bwb[6];
  if constexpr (BF16) ts_bw_lp<LP ? LP : 1>(p.wbm, n16, kq, bwb);
  else if constexpr (SPARSE) {
    for (int e = tid; e < NK * 64; e += F12_THREADS) wbs[e] = p.wbm[e];
  } else {
#pragma unroll
    for (int j = 0; j < NK; j++) bw[j] = p.wbm[j * 64 + lane];   // per-lane B operand of k_convm (PrepLayout::wbm)
  }
  const float bias = p.b[co];
  const f32x4 binit = {bias, bias, bias, bias};
  const float *abase = &tile[kq * PLS + 3];
  if constexpr (SPARSE) {
    // K1: the table form's value for pattern 0 in both channels (the max of four equal sums and 0); K2: conv2 + pool + ReLU
    // of a window of K1 planes, through the kernel's own MFMA sequence on a tile filled with them
    __syncthreads();                                               // slut stands
    if (tid < 8) k1s[tid] = fmaxf(slut[tid] + slut[512 * 8 + tid], 0.f);
    for (int e = tid; e < 2 * F12_ROWS * 8; e += F12_THREADS) (&nz[0][0][0])[e] = 0u;
    __syncthreads();
    for (int e = tid; e < 8 * PLS; e += F12_THREADS) tile[e] = k1s[min(e / PLS, 7)];
    __syncthreads();
    if (wv == 0) {
      const float *a0 = abase + 2 * LS + 16 + n16;                 // any interior M-tile: row pair 1, columns 16 .. 31
      f32x4 d0 = binit;
      if constexpr (BF16) {
#pragma unroll
        for (int J = 0; J < 6; J++) {
          const int tap = 2 * J + (kq >> 1);
          const float *q0 = a0 + (4 * (kq & 1) - kq) * PLS + (tap / 3) * LS + tap % 3;
          d0 = lp_mfma16<LP ? LP : 1>(lp_pk4<LP ? LP : 1>(q0[0], q0[PLS], q0[2 * PLS], q0[3 * PLS]), bwb[J], d0);
        }
      } else {
        auto aof = [&](int j) -> int { return (4 * (j & 1)) * PLS + ((j >> 1) / 3) * LS + ((j >> 1) % 3); };
#pragma unroll
        for (int j = 0; j < NK; j++) d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[aof(j)], wbs[(BF16 ? 0 : j) * 64 + lane], d0, 0, 0, 0);
      }
      float q0 = max_raw(max_raw(d0[0], 0.f), d0[1]);
      q0 = max_raw(q0, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, q0), 0xB1, 0xF, 0xF, true)));
      if (r == 0 && kq == 0) k2s[co] = q0;
    }
    __syncthreads();
    for (int e = tid; e < 8 * PLS / 4; e += F12_THREADS) reinterpret_cast<f32x4 *>(tile)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  bits_commit(0, 0, F12_TH + 1, bits_fetch((int)blockIdx.x, 0, F12_TH + 1));
  unsigned w_ahead = 0u;                                // SPARSE: the bit rows of the step after next, in flight
  if constexpr (SPARSE) { if ((int)blockIdx.x < p.images) w_ahead = bits_fetch((int)blockIdx.x, F12_TH + 1, 2 * F12_TH + 1); }
  __syncthreads();
  const float k2 = SPARSE ? k2s[co] : 0.f;

  // persistent: a workgroup takes images blockIdx.x, blockIdx.x + gridDim.x, ... (table, weights and the zero frame of
  // the tile are set up once; the first bit rows of the next image are fetched under the last step of this one)
#pragma unroll 1
  for (int img = (int)blockIdx.x; img < p.images; img += (int)gridDim.x)
#pragma unroll 1
  for (int step = 0; step < H1 / F12_TH; step++) {
    const int R0 = step * F12_TH;                                  // first conv2 row of the step
    // new p1 rows [pa, pb) -> tile rows pa - (R0 - 1) ..; the first step also makes row 0 (its row -1 is the zero row),
    // the last one leaves row 200 zero
    const int pa = step ? R0 + 1 : 0, pb = min(R0 + F12_TH + 1, H1);
    const int buf = step & 1;
    // the next step's bit rows (of the workgroup's next image behind the last step)
    const bool last = step + 1 == H1 / F12_TH;
    const int nimg = last ? img + (int)gridDim.x : img;
    const bool more = nimg < p.images;
    const int na = last ? 0 : R0 + F12_TH + 1, nb = last ? F12_TH + 1 : min(R0 + 2 * F12_TH + 1, H1);
    unsigned nextw = 0u;
    T12_STAMP(st_b);
    // ---- phase A: conv1 table look-up + pool + ReLU, two adjacent p1 pixels per thread ----
    if (step && pb - pa < F12_TH)  // last step: p1 row 200 does not exist - the tile row behind the image is zero
      for (int e = tid; e < 8 * LS; e += F12_THREADS) tile[(e / LS) * PLS + (F12_TH + 1) * LS + e % LS] = 0.f;
    if (!step)  // the row above the image
      for (int e = tid; e < 8 * LS; e += F12_THREADS) tile[(e / LS) * PLS + e % LS] = 0.f;
    if (step && tid < 8 * 2 * (LS / 4)) {  // rows R0 - 1, R0: kept by the previous step (nobody reads the tile in phase A)
      const int c4 = tid % (LS / 4), rr = (tid / (LS / 4)) & 1, ci = tid / (2 * (LS / 4));
      reinterpret_cast<f32x4 *>(&tile[ci * PLS + rr * LS])[c4] = reinterpret_cast<const f32x4 *>(&halo[buf][ci][rr][0])[c4];
    }
    if constexpr (SPARSE) {  // ... and their column marks (the rows this step writes start at tile row 2: no overlap)
      if (step && tid < 16) nz[buf][tid >> 3][tid & 7] = nz[buf ^ 1][F12_TH + (tid >> 3)][tid & 7];
    }
    for (int q = tid; q < (pb - pa) * 100; q += F12_THREADS) {
      const int py = q / 100, pp = q - py * 100;
      const int x0 = 4 * pp;                                       // window = staged bits x0 .. x0 + 5 of rows 2 py .. 2 py + 3
      unsigned f[2][4];
#pragma unroll
      for (int ci = 0; ci < 2; ci++)
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          const unsigned *rw = &rows[buf][ci][2 * py + rr][x0 >> 5];
          f[ci][rr] = __funnelshift_r(rw[0], rw[1], (unsigned)(x0 & 31)) & 63u;
        }
      float m2[8][2];
      bool dense_pass = true;
      if constexpr (SPARSE) {
        const unsigned any = (f[0][0] | f[0][1] | f[0][2] | f[0][3]) | (f[1][0] | f[1][1] | f[1][2] | f[1][3]);
        dense_pass = __builtin_amdgcn_ballot_w64(any != 0u) != 0;  // wave-uniform
        t_all++;
        if (dense_pass) {
          t_exec++;
          if (any != 0u) atomicOr(&nz[buf][pa + py - (R0 - 1)][(2 * pp) >> 5], 3u << ((2 * pp) & 31));  // columns 2 pp, 2 pp + 1: an even bit and its neighbour
        } else {
          const f32x4 ka = *reinterpret_cast<const f32x4 *>(&k1s[0]), kb = *reinterpret_cast<const f32x4 *>(&k1s[4]);
#pragma unroll
          for (int c = 0; c < 8; c++) m2[c][0] = m2[c][1] = c < 4 ? ka[c & 3] : kb[c & 3];
        }
      }
      if (dense_pass) {
#pragma unroll
      for (int j = 0; j < 2; j++) {
        f32x4 acc[4][2];                                           // [2x2 pixel][channels 0-3 | 4-7]
#pragma unroll
        for (int ci = 0; ci < 2; ci++)
#pragma unroll
          for (int qq = 0; qq < 4; qq++) {
            const int dy = qq >> 1, sh = 2 * j + (qq & 1);
            const unsigned pat = ((f[ci][dy] >> sh) & 7u) | (((f[ci][dy + 1] >> sh) & 7u) << 3) | (((f[ci][dy + 2] >> sh) & 7u) << 6);
            const f32x4 *e = reinterpret_cast<const f32x4 *>(&slut[(ci * 512 + pat) * 8]);
            if (ci == 0) { acc[qq][0] = e[0]; acc[qq][1] = e[1]; }  // the table of channel 0 carries the bias
            else { acc[qq][0] += e[0]; acc[qq][1] += e[1]; }
          }
#pragma unroll
        for (int c = 0; c < 8; c++) {
          float m;  // the operands are ordinary VALU results (interlocked), not MFMA results: asm is safe here
          asm("v_max3_f32 %0, %1, %2, 0" : "=v"(m) : "v"(acc[0][c >> 2][c & 3]), "v"(acc[1][c >> 2][c & 3]));
          asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(acc[2][c >> 2][c & 3]), "v"(acc[3][c >> 2][c & 3]), "v"(m));
          m2[c][j] = m;
        }
      }
      }  // dense_pass
      const int trow = pa + py - (R0 - 1);
      float *dst = &tile[trow * LS + 4 + 2 * pp];
#pragma unroll
      for (int c = 0; c < 8; c++) *reinterpret_cast<float2 *>(dst + c * PLS) = make_float2(m2[c][0], m2[c][1]);
      if (trow >= F12_TH) {  // also the next step's first two rows
#pragma unroll
        for (int c = 0; c < 8; c++)
          *reinterpret_cast<float2 *>(&halo[buf ^ 1][c][trow - F12_TH][4 + 2 * pp]) = make_float2(m2[c][0], m2[c][1]);
      }
    }
    __syncthreads();

    T12_STAMP(st_a);
    // ---- phase B: conv2 on the tile; the next step's bit rows are fetched meanwhile ----
    if constexpr (!SPARSE) { if (more) nextw = bits_fetch(nimg, na, nb); }
    // SPARSE: the bit rows travel TWO steps ahead.  The dense GEMM phase hides the ~2 us of a fetch issued at its start;
    // the sparse step is over before the word lands.  Here, BETWEEN the two phases, the word fetched a whole step ago is
    // committed (the next step's rows) and the step after that is requested: the wait for the word is a vmcnt(0), which
    // also waits for every p2 store in flight - at the top of the step those are the stores the previous GEMM phase has
    // just issued (stamps: 5 000 of a step's 12 700 cycles went there), behind phase A they are a table phase old.
    if constexpr (SPARSE) {
      if (more) bits_commit(buf ^ 1, na, nb, w_ahead);
      int i2 = nimg, s2 = last ? 0 : step + 1;
      if (++s2 == H1 / F12_TH) { s2 = 0; i2 += (int)gridDim.x; }
      const int r2 = s2 * F12_TH;
      w_ahead = i2 < p.images ? bits_fetch(i2, s2 ? r2 + 1 : 0, min(r2 + F12_TH + 1, H1)) : 0u;
    }

    if constexpr (SPARSE) {
      float bwl[BF16 ? 1 : NK];
      if constexpr (!BF16) {
        int zoff;                                   // an offset the compiler cannot see through: the reads stay in the loop
        asm volatile("s_mov_b32 %0, 0" : "=s"(zoff));
#pragma unroll
        for (int j = 0; j < NK; j++) bwl[j] = wbs[j * 64 + lane + zoff];
      }
      ts_gemm_phase_sparse<200, F12_TH / 2, F12_LS, F12_PLS, LP>(abase, bwl, bwb, binit, wv, lane, n16, kq, r,
                                                                  p.out + (((size_t)img * 8 + co) * H2 + (R0 >> 1)) * H2,
                                                                  &nz[buf][0][0], k2, step == 0, last, n_exec, n_all
#if OFX_TRUNK_STAMPS
                                                                  , st_g
#endif
                                                                  );
      // the other buffer's marks are last step's: cleared for the next step (its first two rows are copied in there)
      for (int e = tid; e < F12_ROWS * 8; e += F12_THREADS) (&nz[buf ^ 1][0][0])[e] = 0u;
    } else if constexpr (BF16)
      ts_gemm_phase_bf16<200, F12_TH / 2, F12_LS, F12_PLS, LP ? LP : 1>(abase, bwb, binit, wv, n16, kq, r,
                                                            p.out + (((size_t)img * 8 + co) * H2 + (R0 >> 1)) * H2);
    else
      ts_gemm_phase<200, F12_TH / 2, F12_LS, F12_PLS>(abase, bw, binit, wv, n16, kq, r,
                                                       p.out + (((size_t)img * 8 + co) * H2 + (R0 >> 1)) * H2);
    if constexpr (!SPARSE) { if (more) bits_commit(buf ^ 1, na, nb, nextw); }  // the other buffer: phase A of this step is behind every wave
    __syncthreads();
  }
  if constexpr (SPARSE) {
#if OFX_TRUNK_STAMPS
    T12_STAMP(st_b);
    if (p.stat && tid == 0) {   // diagnostic: cycles of wave 0 in phase A (up to its barrier) / phase B / whole kernel / blocks
      atomicAdd(&p.stat[0], st_a); atomicAdd(&p.stat[1], st_b);          // phase A (with its barrier) / the rest of the step
      atomicAdd(&p.stat[2], st_g[0]); atomicAdd(&p.stat[3], st_g[2]);    // of that: classification + constant stores / the run tiles
    }
#else
    if (p.stat && lane == 0) {
      atomicAdd(&p.stat[0], (unsigned long long)n_exec); atomicAdd(&p.stat[1], (unsigned long long)n_all);
      atomicAdd(&p.stat[2], (unsigned long long)t_exec); atomicAdd(&p.stat[3], (unsigned long long)t_all);
    }
#endif
  }
}

// ---- conv3 as a streaming kernel: k_trunk12's GEMM phase on tiles that arrive by LDS-direct loads --------------------
// Planar f32 input [img][8][100][100] -> planar [img][8][50][50].  One persistent 1024-thread workgroup per CU walks an
// image in 5 steps of 20 rows over TWO tiles (8 planes x 22 rows x 108 floats each, 157 KB): while the GEMM phase
// (ts_gemm_phase: 10 row pairs x 100 columns = 62.5 M-tiles, four chains per wave) runs on one, the 22 rows of the next
// step land in the other by global_load_lds_dwordx4 - no staging registers (a register prefetch of 20 rows spills next
// to four accumulator chains), no LDS store instructions.  An LDS-direct load writes lane l's 16 bytes at base + 16 l, so a
// plane is filled front to back in pieces of 64 float4: lane q of a plane -> (row q / 27, float4 q % 27); float4 0 and
// 26 of a row (the halo columns) and the rows outside the image read 16 zero bytes instead (PrepLayout::zero16).
constexpr int C3_W = 100, C3_TH = 20, C3_LS = 108, C3_ROWS = C3_TH + 2, C3_F4 = C3_LS / 4;
constexpr int C3_PLS = (C3_ROWS * C3_LS + 63) / 64 * 64 + 16;
constexpr int C3_PF4 = C3_ROWS * C3_F4, C3_NI = (C3_PF4 + 63) / 64;  // float4 of a plane, wave-instructions per plane
static_assert(C3_PLS % 64 == 16 && C3_W % C3_TH == 0 && C3_W / 4 + 2 == C3_F4 && 64 * C3_NI * 4 <= C3_PLS + 64 * 4, "tile layout");
static_assert(2 * 8 * C3_PLS * 4 <= 160 * 1024, "two tiles in LDS");

template <int LP>
__global__ __launch_bounds__(F12_THREADS) void k_conv3_stream(ConvParams p, const float *zero16) {
  constexpr bool BF16 = LP != 0;
  constexpr int W = C3_W, H = C3_W, H2 = C3_W / 2, PLS = C3_PLS, NK = 24, STEPS = H / C3_TH;
  __shared__ __align__(16) float tiles[2][8 * C3_PLS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kq = lane >> 4, co = n16 >> 1, r = n16 & 1;

  // rows R0 - 1 .. R0 + C3_TH of image img -> tiles[buf]; 80 wave-instructions spread over the 16 waves
  auto stage = [&](int buf, int img, int R0) {
#pragma unroll 1
    for (int i = wv; i < 8 * C3_NI; i += F12_THREADS / 64) {
      const int ci = i / C3_NI, k = i - ci * C3_NI;
      const int q = 64 * k + lane;
      if (q < C3_PF4) {
        const int row = q / C3_F4, c4 = q - row * C3_F4, gy = R0 - 1 + row;
        const float *src = zero16;
        if (c4 >= 1 && c4 <= W / 4 && gy >= 0 && gy < H) src = p.in + (((size_t)img * 8 + ci) * H + gy) * W + 4 * (c4 - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)&tiles[buf][ci * PLS + 256 * k], 16, 0, 0);
      }
    }
  };

  float bw[BF16 ? 1 : NK];
  lp_x4 bwb[6];
  if constexpr (BF16) ts_bw_lp<LP ? LP : 1>(p.wbm, n16, kq, bwb);
  else {
#pragma unroll
    for (int j = 0; j < NK; j++) bw[j] = p.wbm[j * 64 + lane];
  }
  const float bias = p.b[co];
  const f32x4 binit = {bias, bias, bias, bias};
  if ((int)blockIdx.x < p.images) stage(0, (int)blockIdx.x, 0);
  // the LDS-direct loads are published to the other waves by vmcnt(0) BEFORE the barrier: the memory model only
  // promises lgkmcnt(0) at a workgroup fence, so the wait is written out rather than left to the compiler
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  __syncthreads();
  int buf = 0;
#pragma unroll 1
  for (int img = (int)blockIdx.x; img < p.images; img += (int)gridDim.x)
#pragma unroll 1
  for (int step = 0; step < STEPS; step++) {
    const int R0 = step * C3_TH;
    const bool last = step + 1 == STEPS;
    const int nimg = last ? img + (int)gridDim.x : img;
    if (nimg < p.images) stage(buf ^ 1, nimg, last ? 0 : R0 + C3_TH);
    if constexpr (BF16)
      ts_gemm_phase_bf16<W, C3_TH / 2, C3_LS, C3_PLS, LP ? LP : 1>(&tiles[buf][kq * PLS + 3], bwb, binit, wv, n16, kq, r,
                                                      p.out + (((size_t)img * 8 + co) * H2 + (R0 >> 1)) * H2);
    else
      ts_gemm_phase<W, C3_TH / 2, C3_LS, C3_PLS>(&tiles[buf][kq * PLS + 3], bw, binit, wv, n16, kq, r,
                                                 p.out + (((size_t)img * 8 + co) * H2 + (R0 >> 1)) * H2);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the next tile's LDS-direct loads have landed (see above)
    __syncthreads();
    buf ^= 1;
  }
}

// ---- fp32 MFMA GEMM for the dense layers -------------------------------------------
// C[M][N] = act(A[M][K] (lda) x B[K][N] (ldb) + bias[N])   one wave per 32x32 tile,
// v_mfma_f32_32x32x2_f32: lane l holds A[row l&31][k l>>5], B[k l>>5][col l&31];
// C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).

// `live` (may be null): ordered list of the rows to compute, live[0] = count, live[1 + i] = row (a masked forward:
// the i-th selected ship) - the M-tiles run over the list, A is read and C written at the listed rows
__global__ __launch_bounds__(256) void k_gemm_f32(const float *A, int lda, const float *B, int ldb, const float *bias,
                                                  float *C, int ldc, int M, int N, int K, int relu, const int32_t *live) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int tiles_n = (N + 31) / 32;
  const int tile = blockIdx.x * 4 + wv;
  if (live) M = live[0];
  if (tile >= ((M + 31) / 32) * tiles_n) return;
  const int m0 = (tile / tiles_n) * 32, n0 = (tile % tiles_n) * 32;
  const int r0 = m0 + (lane & 31), c = n0 + (lane & 31), kh = lane >> 5;
  const bool rv = r0 < M, cv = c < N;
  const int r = live ? live[1 + (rv ? r0 : 0)] : r0;
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
        C[(size_t)(live ? live[1 + row] : row) * ldc + c] = v;
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
  const int32_t *live;   // ordered list of the selected ships (with a mask) or null
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
  // with a mask: work item i = the i-th selected ship (p.live: count, then the ordered list); the blocks behind the
  // count leave at once
  const int item = blockIdx.x * 4 + wv, n_items = p.live ? p.live[0] : p.N * p.M;
  if ((int)blockIdx.x * 4 >= n_items) return;  // block-uniform
  const bool on = item < n_items;
  const int s = on ? (p.live ? p.live[1 + item] : item) : 0;
  const int a = s / p.M;
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

__device__ inline float unordered_f32(unsigned o) {  // inverse of ordered_f32
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}

__global__ void k_policy_finish(int S, const uint8_t *mask, const unsigned long long *best, int32_t *ipointer,
                                float *ptr_max) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S || (mask && !mask[s])) return;
  // best == 0: no value ever beat the initial -inf, i.e. the whole map is NaN (a diverged fit) or -inf: np.argmax
  // answers 0 there (the first NaN / the first element) and np.max NaN / -inf; never an out-of-range pointer
  const bool none = best[s] == 0ull;
  const unsigned k = none ? 0u : ~(unsigned)(best[s] & 0xFFFFFFFFull);
  ipointer[2 * s] = (int)(k % PS);      // unravel_index(order='F') of a C-order flat index = (x, y)
  ipointer[2 * s + 1] = (int)(k / PS);  // qlearnIA_V2.py:218-220
  if (ptr_max) ptr_max[s] = none ? __builtin_nanf("") : unordered_f32((unsigned)(best[s] >> 32));  // np.max(ptr_prediction)
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
  float *p1, *p2, *p3, *p4, *g1, *d1, *u0, *up1, *u2fr, *vfr, *c4;
  unsigned long long *best;
  int32_t *iaction, *ipointer, *live;
};

static size_t al(size_t b) { return (b + 255) & ~(size_t)255; }
constexpr int kDense1Chunks = 25;  // split-K of dense1: 5000 = 25 x 200

// N images (trunk runs), S policy samples (heads); the layout depends on both, so a forward with other sizes
// (ofx_policy_forward_obs) invalidates the results a previous forward left in the workspace
// conv1 -> conv2 fused (k_trunk12, one persistent workgroup per CU) once the images fill the CUs a few times over;
// below that the two-kernel form has the shorter critical path (OFX_OPT_TRUNK_FUSE: 1 always, 2 never)
static bool trunk_fused(const ofx_handle *h, size_t N) {
  if (h->opt_trunk_plain) return false;
  return h->opt_trunk_fuse == 1 || (h->opt_trunk_fuse == 0 && N >= 4 * (size_t)h->n_cus);
}

static int policy_workspace(ofx_handle *h, PolicyWs *ws, size_t N, size_t S) {
  size_t f2, f3, f4;
  ofx_head_frame_bytes(S, &f2, &f3, &f4);
  // p1 (5.2 GB at 4096 arenas) exists only in the two-kernel form of the trunk
  const size_t sz[] = {trunk_fused(h, N) ? 0 : al(4ull * N * 8 * 200 * 200), al(4ull * N * 8 * 100 * 100), al(4ull * N * 8 * 50 * 50),
                       al(4ull * N * 5000),          al(4ull * N * 100 * kDense1Chunks), al(4ull * S * 100),
                       al(4ull * S * 625),           al(4ull * S * 2 * 50 * 50),   al(f2), al(f3), al(f4),
                       al(8ull * S),                 al(4ull * S),                 al(8ull * S),
                       al(4ull * (S + 1))};
  size_t total = 0;
  for (size_t b : sz) total += b;
  int rc = ofx_ensure_scratch(h, total);
  if (rc) return rc;
  char *base = (char *)h->scratch;
  void **dst[] = {(void **)&ws->p1,   (void **)&ws->p2,   (void **)&ws->p3,      (void **)&ws->p4,      (void **)&ws->g1,
                  (void **)&ws->d1,   (void **)&ws->u0,   (void **)&ws->up1,     (void **)&ws->u2fr,    (void **)&ws->vfr,
                  (void **)&ws->c4,   (void **)&ws->best, (void **)&ws->iaction, (void **)&ws->ipointer,
                  (void **)&ws->live};
  for (size_t i = 0; i < sizeof(sz) / sizeof(sz[0]); i++) { *dst[i] = base; base += sz[i]; }
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

// conv1 + BatchNorm + ReLU + pool of n stored observations (bits [n][2][5000]) through a caller-built table
// [2][512][8] (channel 0 carries the bias): out [n][8][200][200].  The fit's first layer (ofx_fit.hip): its table folds
// the BATCH statistics.
int ofx_launch_conv1_lut(ofx_handle *h, const void *bits, int n, const float *lut, float *out) {
  ConvParams cp{};
  cp.bits[0] = reinterpret_cast<const unsigned *>(bits);
  cp.bits[1] = cp.bits[0] + (PS * PS) / 32;
  cp.bits_stride = 2 * (size_t)((PS * PS) / 32);
  cp.out = out; cp.H = PS; cp.W = PS; cp.images = n;
  hipLaunchKernelGGL(k_conv1_lut<40>, dim3((unsigned)(n * (PS / 40))), dim3(256), 0, h->stream, cp, lut);
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}

// (also the dense layers of the fit's forward, ofx_train.hip)
int ofx_launch_gemm(ofx_handle *h, const float *A, int lda, const float *B, int ldb, const float *bias, float *C, int ldc,
                    int M, int N, int K, int relu, const int32_t *live) {
  const int tiles = ((M + 31) / 32) * ((N + 31) / 32);
  hipLaunchKernelGGL(k_gemm_f32, dim3((tiles + 3) / 4), dim3(256), 0, h->stream, A, lda, B, ldb, bias, C, ldc, M, N, K,
                     relu, live);
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}

static const int t_d1 = 24, t_d2 = 26, t_o1 = 28, t_ud = 30, t_up = 32, t_u4 = 50;  // tensor indices of the blob

// BN folding, phase weights, tables: `weights` -> the handle's prepared-weights buffer (two small launches)
// `slot`: &h->prep (the pinned blob's buffer) or &h->prep_tmp (any other blob: a target network or a one-off forward
// next to a pinned blob must not overwrite the pinned blob's prepared weights)
static int policy_prepare(ofx_handle *h, const float *weights, float **slot) {
  const PrepLayout L = prep_layout();
  if (!*slot) {
    OFX_HIP(hipMalloc((void **)slot, sizeof(float) * L.total));
    OFX_HIP(hipMemsetAsync(*slot, 0, sizeof(float) * L.total, h->stream));
  }
  int32_t off[64], cnt[64];
  policy_layout(off, cnt);
  PrepParams pp;
  pp.w = weights; pp.prep = *slot;
  for (int i = 0; i < 4; i++) {
    pp.src_k[i] = off[6 * i]; pp.src_b[i] = off[6 * i + 1]; pp.src_g[i] = off[6 * i + 2];
    pp.cin[i] = kTrunkCin[i]; pp.cout[i] = 8; pp.dst_w[i] = L.tw[i]; pp.dst_b[i] = L.tb[i];
  }
  for (int i = 0; i < 3; i++) {
    pp.src_k[4 + i] = off[t_up + 6 * i]; pp.src_b[4 + i] = off[t_up + 6 * i + 1]; pp.src_g[4 + i] = off[t_up + 6 * i + 2];
    pp.cin[4 + i] = kUpCin[i]; pp.cout[4 + i] = kUpCout[i]; pp.dst_w[4 + i] = L.uw[i]; pp.dst_b[4 + i] = L.ub[i];
  }
  pp.src_k4 = off[t_u4]; pp.src_b4 = off[t_u4 + 1];
  pp.dst_w4raw = L.w4raw; pp.dst_b4 = L.b4; pp.dst_efr = L.efr;
  pp.dst_w4eff_c = L.w4eff_c; pp.dst_w3mf = L.w3mf; pp.dst_w2mf = L.w2mf; pp.dst_w2fr = L.w2fr; pp.dst_w3fr = L.w3fr; pp.dst_lut1 = L.lut1;
  for (int i = 0; i < 3; i++) pp.dst_wbm[i] = L.wbm[i];
  pp.legacy = h->opt_bilinear_legacy;
  pp.phase = 0;
  hipLaunchKernelGGL(k_policy_prepare, dim3(32), dim3(256), 0, h->stream, pp);
  pp.phase = 1;
  hipLaunchKernelGGL(k_policy_prepare, dim3(4), dim3(256), 0, h->stream, pp);
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}

// Pinned weights: prepared once, reused by every forward that names the same blob until it is unpinned, re-pinned or
// trained on (ofx_dqn_fit re-prepares a pinned blob behind its update).
extern "C" int ofx_policy_pin_weights(ofx_handle *h, const float *weights) {
  if (!h) { ofx_set_error("ofx_policy_pin_weights: null handle"); return OFX_ERR_INVALID; }
  OFX_HIP(hipSetDevice(h->cfg.device));
  h->prep_pinned = nullptr;
  if (!weights) return OFX_OK;
  int rc = policy_prepare(h, weights, &h->prep);
  if (rc) return rc;
  h->prep_pinned = weights;
  return OFX_OK;
}

int ofx_policy_weights_updated(ofx_handle *h, const float *weights) {  // ofx_train.hip: the blob changed in place
  if (h->prep_pinned && h->prep_pinned == weights) return policy_prepare(h, weights, &h->prep);
  return OFX_OK;
}

extern "C" int ofx_policy_trunk_stats(ofx_handle *h, int64_t *counts_host) {
  if (!h || !counts_host) { ofx_set_error("ofx_policy_trunk_stats: null argument"); return OFX_ERR_INVALID; }
  for (int i = 0; i < 4; i++) counts_host[i] = 0;
  if (!h->trunk_stat) return OFX_OK;
  OFX_HIP(hipSetDevice(h->cfg.device));
  OFX_HIP(hipMemcpyAsync(counts_host, h->trunk_stat, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
  OFX_HIP(hipMemsetAsync(h->trunk_stat, 0, 4 * sizeof(unsigned long long), h->stream));
  OFX_HIP(hipStreamSynchronize(h->stream));
  return OFX_OK;
}

extern "C" int ofx_set_option(ofx_handle *h, int32_t option, int32_t value) {
  if (!h) { ofx_set_error("ofx_set_option: null handle"); return OFX_ERR_INVALID; }
  switch (option) {
    case OFX_OPT_TRUNK_PLAIN: h->opt_trunk_plain = value != 0; return OFX_OK;
    case OFX_OPT_TRUNK_FUSE:
      if (value < 0 || value > 2) { ofx_set_error("ofx_set_option: OFX_OPT_TRUNK_FUSE takes 0 (auto), 1 (always), 2 (never)"); return OFX_ERR_INVALID; }
      h->opt_trunk_fuse = value; return OFX_OK;
    case OFX_OPT_FRAMES_REF: h->opt_frames_ref = value != 0; return OFX_OK;
    case OFX_OPT_TRUNK_SPARSE:
      if (value && !h->trunk_stat) {
        OFX_HIP(hipSetDevice(h->cfg.device));
        OFX_HIP(hipMalloc((void **)&h->trunk_stat, 4 * sizeof(unsigned long long)));
        OFX_HIP(hipMemsetAsync(h->trunk_stat, 0, 4 * sizeof(unsigned long long), h->stream));
      }
      h->opt_trunk_sparse = value != 0; return OFX_OK;
    case OFX_OPT_FIT_PLAIN: h->opt_fit_plain = value != 0; return OFX_OK;
    case OFX_OPT_POLICY_BF16:
      if (value < 0 || value > 2) { ofx_set_error("ofx_set_option: OFX_OPT_POLICY_BF16 takes 0 (fp32), 1 (bf16 operands), 2 (fp16 operands)"); return OFX_ERR_INVALID; }
      h->opt_policy_lowp = value; return OFX_OK;
    case OFX_OPT_BILINEAR_LEGACY:  // a different function, not a variant: the prepared phase weights depend on it
      if (h->opt_bilinear_legacy == (value != 0)) return OFX_OK;
      h->opt_bilinear_legacy = value != 0;
      OFX_HIP(hipSetDevice(h->cfg.device));
      return h->prep_pinned ? policy_prepare(h, h->prep_pinned, &h->prep) : OFX_OK;
    default: ofx_set_error("ofx_set_option: unknown option %d", option); return OFX_ERR_INVALID;
  }
}

// The forward proper: N images (two 1-bit maps each: word bits?[img * bits_stride + w]) with M policy samples per
// image; vec8 = explicit observation heads [N*M][8] or null (the live state of the handle's arenas).
static int policy_forward_impl(ofx_handle *h, const float *weights, int N, int M, const unsigned *bits0,
                               const unsigned *bits1, size_t bits_stride, const float *vec8, const uint8_t *ship_mask,
                               float *act_values, int32_t *iaction, int32_t *ipointer, float *heatmap, float *ptr_max,
                               const int32_t *probe, float *ptr_probe) {
  const int S = N * M;
  PolicyWs ws;
  int rc = policy_workspace(h, &ws, N, S);
  if (rc) return rc;
  int32_t off[64], cnt[64];
  policy_layout(off, cnt);
  const PrepLayout L = prep_layout();

  // 0. prepared weights: reused when the blob is pinned
  //    (h->prep holds exactly the pinned blob's preparation; every other blob goes through h->prep_tmp)
  const bool pinned = h->prep && h->prep_pinned == weights;
  if (!pinned && (rc = policy_prepare(h, weights, &h->prep_tmp))) return rc;
  const float *prep = pinned ? h->prep : h->prep_tmp;

  // 1. trunk, once per arena
  ConvParams cp;
  memset(&cp, 0, sizeof(cp));
  cp.bits[0] = bits0;
  cp.bits[1] = bits1;
  cp.bits_stride = bits_stride;
  cp.w = prep + L.tw[0]; cp.b = prep + L.tb[0]; cp.out = ws.p1;
  const int lowp = vec8 == nullptr ? h->opt_policy_lowp : 0;  // opt-in (1 bf16, 2 fp16 operands), the rollout's forward only
  const bool plain = h->opt_trunk_plain;  // OFX_OPT_TRUNK_PLAIN: every trunk layer through the plain VALU kernel
  const bool fused12 = trunk_fused(h, (size_t)N);
  if (plain) rc = launch_conv<2, 8, 10, 100, 1, true, false>(h, cp, N, 400);
  else if (fused12) {
    cp.out = ws.p2; cp.b = prep + L.tb[1]; cp.wbm = prep + L.wbm[0]; cp.images = N;
    const dim3 g12((unsigned)(N < h->n_cus ? N : h->n_cus));
    const bool sparse = h->opt_trunk_sparse || vec8 != nullptr;   // exact either way: the forwards on stored observations (DQN targets) always take it
    cp.stat = h->opt_trunk_sparse ? h->trunk_stat : nullptr;
    const float *lut1 = prep + L.lut1;
#define T12(LP_) do { if (sparse) hipLaunchKernelGGL((k_trunk12<LP_, true>), g12, dim3(F12_THREADS), 0, h->stream, cp, lut1); \
                      else hipLaunchKernelGGL((k_trunk12<LP_, false>), g12, dim3(F12_THREADS), 0, h->stream, cp, lut1); } while (0)
    if (lowp == 1) T12(1); else if (lowp == 2) T12(2); else T12(0);
#undef T12
    OFX_HIP(hipGetLastError());
  } else {
    cp.H = 400; cp.W = 400;
    hipLaunchKernelGGL(k_conv1_lut<40>, dim3((unsigned)(N * (400 / 40))), dim3(256), 0, h->stream, cp,
                       (const float *)(prep + L.lut1));
    OFX_HIP(hipGetLastError());
  }
  if (rc) return rc;
  // k_convm tile shapes from an A/B on the chip (conv2: 2 row pairs x 208 columns, 29 KB of LDS, five workgroups per CU)
  cp.in = ws.p1; cp.w = prep + L.tw[1]; cp.b = prep + L.tb[1]; cp.out = ws.p2; cp.wbm = prep + L.wbm[0];
  if (plain) rc = launch_conv<8, 8, 10, 100, 0, true, false>(h, cp, N, 200);
  else if (!fused12) rc = launch_convm<8, 2, 13, 0, false, 1>(h, cp, N, 200);
  if (rc) return rc;
  cp.in = ws.p2; cp.w = prep + L.tw[2]; cp.b = prep + L.tb[2]; cp.out = ws.p3; cp.wbm = prep + L.wbm[1];
  if (plain) rc = launch_conv<8, 8, 10, 100, 0, true, false>(h, cp, N, 100);
  else if (fused12) {  // large batches: the streaming form, like conv1 -> conv2
    cp.images = N;
    const dim3 g3((unsigned)(N < h->n_cus ? N : h->n_cus));
    if (lowp == 1) hipLaunchKernelGGL(k_conv3_stream<1>, g3, dim3(F12_THREADS), 0, h->stream, cp, (const float *)(prep + L.zero16));
    else if (lowp == 2) hipLaunchKernelGGL(k_conv3_stream<2>, g3, dim3(F12_THREADS), 0, h->stream, cp, (const float *)(prep + L.zero16));
    else hipLaunchKernelGGL(k_conv3_stream<0>, g3, dim3(F12_THREADS), 0, h->stream, cp, (const float *)(prep + L.zero16));
    OFX_HIP(hipGetLastError());
  } else rc = launch_convm<8, 4, 7, 0, false, 1>(h, cp, N, 100);
  if (rc) return rc;
  cp.in = ws.p3; cp.w = prep + L.tw[3]; cp.b = prep + L.tb[3]; cp.out = ws.p4; cp.wbm = prep + L.wbm[2];
  if (plain) rc = launch_conv<8, 8, 10, 50, 0, true, true>(h, cp, N, 50);  // (h,w,c) = Flatten order
  else if (lowp == 1 && fused12) rc = launch_convm<8, 10, 4, 0, true, 1, 1>(h, cp, N, 50);
  else if (lowp == 2 && fused12) rc = launch_convm<8, 10, 4, 0, true, 1, 2>(h, cp, N, 50);
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
  // a masked forward works on the ordered list of the selected ships from here on (one scan of the mask)
  const int32_t *live = nullptr;
  if (ship_mask) {
    if ((rc = ofx_head_compact(h, S, ship_mask, ws.live))) return rc;
    live = ws.live;
  }
  hp.mask = ship_mask; hp.live = live; hp.d1 = ws.d1; hp.act = act_values; hp.iaction = iaction ? iaction : ws.iaction;
  hipLaunchKernelGGL(k_head_dense, dim3((S + 3) / 4), dim3(256), 0, h->stream, hp);
  OFX_HIP(hipGetLastError());

  // 3. head-2: updense1 on MFMA, upconv1 (1 -> 2 @ 50x50), then upconv2-4 + arg-max in the streaming kernel (ofx_head.hip)
  if ((rc = ofx_launch_gemm(h, ws.d1, 100, weights + off[t_ud], 625, weights + off[t_ud + 1], ws.u0, 625, S, 625, 100, 1, live)))
    return rc;
  ConvParams up;
  memset(&up, 0, sizeof(up));
  up.live = live; up.images = S; up.legacy = h->opt_bilinear_legacy;
  up.in = ws.u0; up.w = prep + L.uw[0]; up.b = prep + L.ub[0]; up.out = ws.up1;
  // every ship: one workgroup each (0.22 ms; a bounded grid that loops is 0.02 ms slower); with a mask: a bounded grid over the list
  hipLaunchKernelGGL(k_upconv1, dim3((unsigned)(live && S > 4096 ? 4096 : S)), dim3(256), 0, h->stream, up);
  OFX_HIP(hipGetLastError());
  OFX_HIP(hipMemsetAsync(ws.best, 0, sizeof(unsigned long long) * S, h->stream));
  HeadParams2 hp2;
  memset(&hp2, 0, sizeof(hp2));
  hp2.S = S; hp2.up1 = ws.up1;
  hp2.w2mf = prep + L.w2mf; hp2.b2 = prep + L.ub[1]; hp2.w2raw = prep + L.uw[1];
  hp2.w3mf = prep + L.w3mf; hp2.b3 = prep + L.ub[2]; hp2.w3raw = prep + L.uw[2];
  hp2.w4eff_c = prep + L.w4eff_c; hp2.b4 = prep + L.b4; hp2.w4raw = prep + L.w4raw;
  hp2.w2fr = prep + L.w2fr; hp2.w3fr = prep + L.w3fr; hp2.efr = prep + L.efr;
  hp2.u2fr = ws.u2fr; hp2.vfr = ws.vfr; hp2.c4 = ws.c4;
  hp2.frames_ref = h->opt_frames_ref; hp2.legacy = h->opt_bilinear_legacy;
  hp2.bf16 = lowp;  // the rollout's forward only: targets and fit stay fp32
  hp2.mask = ship_mask; hp2.live = ws.live; hp2.live_ready = ship_mask != nullptr; hp2.best = ws.best; hp2.heat = heatmap; hp2.probe = probe; hp2.ptr_probe = probe ? ptr_probe : nullptr;
  const int pb = h->prof_base;  // ofx_policy_profile: events around the dominant kernel, until the ring is full
  hp2.event_base = pb;
  if ((rc = ofx_launch_head(h, hp2))) return rc;
  if (pb >= 0) h->prof_base = pb + 3 < OFX_RING_MAX ? pb + 2 : -1;
  hipLaunchKernelGGL(k_policy_finish, dim3((S + 255) / 256), dim3(256), 0, h->stream, S, ship_mask, ws.best,
                     ipointer ? ipointer : ws.ipointer, ptr_max);
  OFX_HIP(hipGetLastError());
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
  OFX_HIP(hipSetDevice(h->cfg.device));
  const size_t words = (size_t)(PS * PS) >> 5;
  return policy_forward_impl(h, weights, n_obs, 1, (const unsigned *)bits, (const unsigned *)bits + words, 2 * words, vec8,
                             nullptr, act_values, iaction, ipointer, nullptr, ptr_max, probe, ptr_probe);
}

// model.predict on stored observations with the whole heat map written out (ofx_dqn_fit_reference, ofx_train.hip)
int ofx_policy_predict_obs(ofx_handle *h, const float *weights, int32_t n_obs, const void *bits, const float *vec8,
                           float *act_values, float *heatmap, float *ptr_max) {
  const size_t words = (size_t)(PS * PS) >> 5;
  return policy_forward_impl(h, weights, n_obs, 1, (const unsigned *)bits, (const unsigned *)bits + words, 2 * words, vec8,
                             nullptr, act_values, nullptr, nullptr, heatmap, ptr_max, nullptr, nullptr);
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
  if (r.ship < 0) {
    if (q_sa) q_sa[i] = p_sp[i] = 0.f;
    y_act[i] = y_ptr[i] = 0.f;
    return;
  }
  const float live = r.done ? 0.f : 1.f;  // int(not done)
  if (q_sa) {   // the forward on `state` was run
    q_sa[i] = act_prev[2 * i + (r.iaction ? 1 : 0)];
    p_sp[i] = probe_prev[i];
  }
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
  if (!h || !weights || !rows || !bits_prev || !bits_next || (!q_sa) != (!p_sp) || !y_act || !y_ptr || n < 1) {
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
  // q_sa = p_sp = NULL: the caller only wants the targets (ofx_dqn_fit's own forward gives the current values) - the forward
  // on `state` is skipped
  if (q_sa && (rc = ofx_policy_forward_obs(h, weights, n, bits_prev, vec_prev, act_prev, nullptr, nullptr, nullptr, probe, probe_prev)))
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
