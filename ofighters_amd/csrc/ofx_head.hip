// ofx_head.hip - head-2 tail of the bi-head "pointer_model" (agents/qlearnIA_V2.py:170-188 + the arg-max glue of
// :218-220): [bilinear x2 + conv 2->4 + BN + ReLU] -> [bilinear x2 + conv 4->8 + BN + ReLU] -> [bilinear x2 +
// conv 8->1] -> arg-max, per policy ship, as a ROW-STREAMING kernel for gfx950.  fp32 (exact f32 MFMAs).
//
// A x2 bilinear upsample followed by a 3x3 convolution is, per output parity (phase), a 3x3 convolution on the
// low-resolution grid with pre-combined weights (k_policy_prepare), so no upsampled tensor is ever formed:
//   stage A  uprelu2 (100x100x4) from uprelu1: GEMM  [pixel][phase*4+co] , K = 9 taps x 2 ci      v_mfma_f32_16x16x4
//   stage B  uprelu3 (200x200x8) from uprelu2: GEMM  [quad][phase*4+co] x 2 channel halves, K = 36  v_mfma_f32_16x16x4
//   stage C  heat map (400x400)  from uprelu3, in TAP-PLANE form (r04).  The up-sampling and the 3x3 taps are linear and
//            the layer has ONE output channel, so the channel sum goes first, on the low-resolution grid:
//              V_t[y][x] = sum_c w4[t][c] uprelu3_c[y][x]            t = (ky, kx), 9 tap planes    (1x1 convolution, 72 MACs)
//              heat[2y+a][2x+b] = b4 + sum_t up2(V_t)[2y+a+ky-1][2x+b+kx-1]                         (separable stencil, 60 MACs)
//            against 288 MACs per low-resolution pixel for the pre-combined phase weights (r01 - r03: 144 v_mfma_f32_4x4x1
//            per pixel pair).  The 1x1 runs on the matrix cores in the PRODUCER, straight from stage B's accumulators:
//            stage B is issued with the weights as the A operand, so a lane ends up with all 8 channels of ONE uprelu3
//            pixel, which is the B operand of 24 v_mfma_f32_4x4x1 (16 blocks, the weight block broadcast with CBSZ /
//            ABID: M = 4 taps, N = 64 pixels).  uprelu3 itself is never stored; the 9 tap planes are.  The stencil runs
//            on the vector ALU in the consumers (v_pk_fma_f32 over a pixel pair).
//
// k_head_stream: one 512-thread workgroup streams a 100-column half of one ship's plane top to bottom.  The tap planes
// and uprelu2 live only as ROLLING WINDOWS of rows in LDS (16-row rings); waves 0-3 produce (stage B + the 1x1), waves
// 4-7 consume (stage A for the producers' next rows, the stencil + arg-max) two sub-steps behind, ONE barrier per
// sub-step of 5 rows.  Nothing of the 42 GB uprelu3 tensor or the 5 GB uprelu2 tensor (32768 ships) touches HBM.  The
// schedule (lags, ring sizes) is checked by tools/head_schedule.py.
//
// Zero padding: the phase form sees the clamp-extended low-resolution plane, which differs from Keras' zero padding
// only on the 1-pixel frame of each layer's output.  k_head_frames computes those thin lines exactly (frame lines of
// uprelu2 / uprelu3, correction lines of the heat map) in a small pre-pass; k_head_stream overwrites the frame cells
// of its rings with the exact values and starts the heat-map accumulators of frame pixels at bias - correction.
#include "ofx_head.h"
#include "ofx_lowp.h"
#include "ofx_diag.h"
#include <stdlib.h>


typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define HD_PS 400

// max without the canonicalising v_max(x, x) of fmaxf(); compiler-visible (never feed an MFMA result to inline asm:
// the hazard recogniser does not look inside an asm statement)
__device__ __forceinline__ float hd_max_raw(float x, float floor) {
  float pinf;
  asm("s_mov_b32 %0, 0x7f800000" : "=s"(pinf));
  return __builtin_amdgcn_fmed3f(x, floor, pinf);
}

__device__ __forceinline__ unsigned hd_ordered_f32(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ---------------------------------------------------------------------------------------------------------------
// Ship mask -> ordered list.  A masked launch used to give every ship a workgroup and let the unselected ones exit:
// with a regular mask (the reference's own line-up has ONE policy ship per arena, lib/ofighters.py:53) the live
// workgroups then fall on a fraction of the CUs - the dispatcher hands workgroups out round-robin - and the kernel
// ran at half speed (3.8 ms for 4096 ships against 15.5 ms for 32768).  Work item i of a masked launch is the i-th
// selected ship instead.  One workgroup; live[0] = count, live[1 + i] = ship.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_head_compact(int S, const uint8_t *mask, int32_t *live) {
  __shared__ int cnt[1024];
  const int t = threadIdx.x, per = (S + 1023) / 1024, lo = min(t * per, S), hi = min(lo + per, S);
  int c = 0;
  for (int s = lo; s < hi; s++) c += mask[s] != 0;
  cnt[t] = c;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {  // inclusive scan
    const int v = t >= d ? cnt[t - d] : 0;
    __syncthreads();
    cnt[t] += v;
    __syncthreads();
  }
  int o = cnt[t] - c;
  for (int s = lo; s < hi; s++)
    if (mask[s]) live[1 + o++] = s;
  if (t == 1023) live[0] = cnt[1023];
}

// ship of work item i (block-uniform), -1 when there is none
__device__ __forceinline__ int hd_ship(const HeadParams2 &p, int i) {
  if (p.live) return i < p.live[0] ? p.live[1 + i] : -1;
  return i < p.S ? i : -1;
}

// ---------------------------------------------------------------------------------------------------------------
// k_head_frames: exact frame lines.  One workgroup per ship.
// ---------------------------------------------------------------------------------------------------------------

// x2 bilinear with edge clamp along a line of n samples, at up-res index u in [-1, 2n] (oracle: upsample2).
// Half-pixel centres: even u = 2k -> L[k-1] + (L[k] - L[k-1]) * .75 ; odd u = 2k+1 -> L[k] + (L[k+1] - L[k]) * .25.
// Legacy (src = u / 2): even -> L[k] ; odd -> L[k] + (L[k+1] - L[k]) * .5.
__device__ __forceinline__ void hf_taps(int u, int n, int &ka, int &kb, float &w, int legacy) {
  const int k = u >> 1;  // arithmetic: u = -1 -> k = -1
  if (legacy) { ka = k; kb = k + 1; w = (u & 1) ? 0.5f : 0.f; }
  else if (u & 1) { ka = k; kb = k + 1; w = 0.25f; } else { ka = k - 1; kb = k; w = 0.75f; }
  ka = min(max(ka, 0), n - 1);
  kb = min(max(kb, 0), n - 1);
}
__device__ __forceinline__ float hf_up(const float *L, int stride, int n, int u, int legacy) {
  int ka, kb; float w;
  hf_taps(u, n, ka, kb, w, legacy);
  const float a = L[ka * stride], b = L[kb * stride];
  return a + (b - a) * w;
}

constexpr int HF_THREADS = 512;
// LDS map of k_head_frames (floats).  Region A is used twice: uprelu1 + its up-sampled bands, later the up-sampled
// bands of uprelu2.
constexpr int HF_L1 = 0;                       // l1[2][50][50]
constexpr int HF_U1R = HF_L1 + 5000;           // U1r[6][2][102]: up-res rows {0,1,2,97,98,99}, zero-extended columns (index x + 1)
constexpr int HF_U1C = HF_U1R + 6 * 2 * 102;   // U1c[6][2][102]: up-res columns, zero-extended rows (index y + 1)
constexpr int HF_A_END = HF_U1C + 6 * 2 * 102; // 7448
constexpr int HF_U2R = 0;                      // U2r[4][4][202]: up-res rows {0,1,198,199} of uprelu2 (index x + 1)
constexpr int HF_U2C = HF_U2R + 4 * 4 * 202;   // U2c[4][4][202]
static_assert(HF_U2C + 4 * 4 * 202 <= HF_A_END, "region A");
constexpr int HF_B2R = HF_A_END;               // u2rb[4][100][4]: uprelu2 rows {0,1,98,99}
constexpr int HF_B2C = HF_B2R + 1600;          // u2cb[4][100][4]: uprelu2 columns {0,1,98,99}, [band][y][ch]
constexpr int HF_U3L = HF_B2C + 1600;          // u3l[4][200][8]: exact frame lines of uprelu3 (top, bottom, left, right)
constexpr int HF_TOTAL = HF_U3L + 6400;        // 17048 floats = 68 KB

// Reference version: everything from the definition (up-sample, then convolve with zero padding) on the VALU.  Kept
// for the agreement test of the phase-form version below (OFX_OPT_FRAMES_REF).
__global__ __launch_bounds__(HF_THREADS) void k_head_frames_ref(HeadParams2 p) {
  __shared__ __align__(16) float sm[HF_TOTAL];
  const int tid = threadIdx.x;
  const int s = hd_ship(p, (int)blockIdx.x);
  if (s < 0) return;  // block-uniform
  float *l1 = sm + HF_L1, *U1r = sm + HF_U1R, *U1c = sm + HF_U1C, *U2r = sm + HF_U2R, *U2c = sm + HF_U2C;
  float *u2rb = sm + HF_B2R, *u2cb = sm + HF_B2C, *u3l = sm + HF_U3L;

  for (int e = tid; e < 5000; e += HF_THREADS) l1[e] = p.up1[(size_t)s * 5000 + e];
  __syncthreads();

  // ---- up-sampled bands of uprelu1 (100x100 domain), zero outside the image ----
  for (int e = tid; e < 2 * 6 * 2 * 102; e += HF_THREADS) {
    const int t = e % 102 - 1, ci = (e / 102) % 2, b = (e / 204) % 6, isc = e / 1224;
    const int f = b < 3 ? b : b + 94;  // the band's fixed up-res coordinate
    float v = 0.f;
    if (t >= 0 && t < 100) {
      const int Y = isc ? t : f, X = isc ? f : t;
      int ya, yb, xa, xb; float wy, wx;
      hf_taps(Y, 50, ya, yb, wy, p.legacy);
      hf_taps(X, 50, xa, xb, wx, p.legacy);
      const float *pl = l1 + ci * 2500;
      const float a = pl[ya * 50 + xa], bq = pl[ya * 50 + xb], d = pl[yb * 50 + xa], g = pl[yb * 50 + xb];
      const float top = a + (bq - a) * wx, bot = d + (g - d) * wx;
      v = top + (bot - top) * wy;
    }
    (isc ? U1c : U1r)[(b * 2 + ci) * 102 + t + 1] = v;
  }
  __syncthreads();

  // ---- uprelu2 on its 2-wide frame bands, from the definition (zero padding) ----
  for (int e = tid; e < 2 * 4 * 100 * 4; e += HF_THREADS) {
    const int co = e % 4, t = (e / 4) % 100, b = (e / 400) % 4, isc = e / 1600;
    const int f = b < 2 ? b : b + 96;  // row (or column) of the band
    float acc = p.b2[co];
#pragma unroll
    for (int d0 = 0; d0 < 3; d0++) {   // offset along the band's fixed axis
      const int u = f + d0 - 1;         // up-res coordinate on that axis
      if (u < 0 || u > 99) continue;    // zero padding
      const int ub = u < 3 ? u : u - 94;
#pragma unroll
      for (int d1 = 0; d1 < 3; d1++)
#pragma unroll
        for (int ci = 0; ci < 2; ci++) {
          const int dy = isc ? d1 : d0, dx = isc ? d0 : d1;
          const float uv = (isc ? U1c : U1r)[(ub * 2 + ci) * 102 + t + d1];  // index (t + d1 - 1) + 1
          acc += p.w2raw[((dy * 3 + dx) * 2 + ci) * 4 + co] * uv;
        }
    }
    (isc ? u2cb : u2rb)[(b * 100 + t) * 4 + co] = fmaxf(acc, 0.f);
  }
  __syncthreads();
  if (tid < 16) {  // the corner cells exist in both bands: one value (row band wins)
    const int co = tid & 3, c = tid >> 2;  // corners: (row 0 | 99) x (col 0 | 99)
    const int rb = (c & 1) ? 3 : 0, x = (c & 2) ? 99 : 0;
    const int cb = (c & 2) ? 3 : 0, y = (c & 1) ? 99 : 0;
    u2cb[(cb * 100 + y) * 4 + co] = u2rb[(rb * 100 + x) * 4 + co];
  }
  __syncthreads();
  for (int e = tid; e < 4 * 100 * 4; e += HF_THREADS) {  // lines: row 0, row 99, col 0, col 99
    const int ln = e / 400, r = e - ln * 400;
    p.u2fr[(size_t)s * 1600 + e] = ln < 2 ? u2rb[(ln ? 3 : 0) * 400 + r] : u2cb[(ln == 3 ? 3 : 0) * 400 + r];
  }

  // ---- up-sampled bands of uprelu2 (200x200 domain), zero outside: rows / columns {0,1,198,199} ----
  // (region A is dead: every read of l1 / U1 was in front of the barriers above)
  for (int e = tid; e < 2 * 4 * 4 * 202; e += HF_THREADS) {
    const int t = e % 202 - 1, ci = (e / 202) % 4, b = (e / 808) % 4, isc = e / 3232;
    const int f = b < 2 ? b : b + 196;
    float v = 0.f;
    if (t >= 0 && t < 200) {
      int fa, fb, ta, tb; float wf, wt;
      hf_taps(f, 100, fa, fb, wf, p.legacy);  // band axis: source rows (columns) fa, fb are inside the 2-wide band
      hf_taps(t, 100, ta, tb, wt, p.legacy);
      const int ba = fa < 2 ? fa : fa - 96, bb = fb < 2 ? fb : fb - 96;
      const float *src = isc ? u2cb : u2rb;
      const float a = src[(ba * 100 + ta) * 4 + ci], bq = src[(ba * 100 + tb) * 4 + ci];
      const float d = src[(bb * 100 + ta) * 4 + ci], g = src[(bb * 100 + tb) * 4 + ci];
      // oracle order: x first, then y
      if (!isc) {  // rows: band axis = y
        const float top = a + (bq - a) * wt, bot = d + (g - d) * wt;
        v = top + (bot - top) * wf;
      } else {     // columns: band axis = x ; a = (x fa, y ta), bq = (fa, tb), d = (fb, ta), g = (fb, tb)
        const float top = a + (d - a) * wf, bot = bq + (g - bq) * wf;
        v = top + (bot - top) * wt;
      }
    }
    (isc ? U2c : U2r)[(b * 4 + ci) * 202 + t + 1] = v;
  }
  __syncthreads();

  // ---- exact frame lines of uprelu3: top (y = 0), bottom (y = 199), left (x = 0), right (x = 199) ----
  for (int e = tid; e < 4 * 200 * 8; e += HF_THREADS) {
    const int co = e % 8, t = (e / 8) % 200, ln = e / 1600, isc = ln >> 1;
    const int f = (ln & 1) ? 199 : 0;
    float acc = p.b3[co];
#pragma unroll
    for (int d0 = 0; d0 < 3; d0++) {
      const int u = f + d0 - 1;
      if (u < 0 || u > 199) continue;
      const int ub = u < 2 ? u : u - 196;
#pragma unroll
      for (int d1 = 0; d1 < 3; d1++)
#pragma unroll
        for (int ci = 0; ci < 4; ci++) {
          const int dy = isc ? d1 : d0, dx = isc ? d0 : d1;
          const float uv = (isc ? U2c : U2r)[(ub * 4 + ci) * 202 + t + d1];
          acc += p.w3raw[((dy * 3 + dx) * 4 + ci) * 8 + co] * uv;
        }
    }
    u3l[(ln * 200 + t) * 8 + co] = fmaxf(acc, 0.f);
  }
  __syncthreads();
  if (tid < 32) {  // corners: the row lines win
    const int co = tid & 7, c = tid >> 3;
    const int rl = (c & 1) ? 1 : 0, x = (c & 2) ? 199 : 0;
    const int cl = (c & 2) ? 3 : 2, y = (c & 1) ? 199 : 0;
    u3l[(cl * 200 + y) * 8 + co] = u3l[(rl * 200 + x) * 8 + co];
  }
  __syncthreads();
  // the tap planes along the four lines: V_t = sum_c w4[t][c] uprelu3_c (what k_head_stream's ring holds)
  for (int e = tid; e < 4 * 200 * 9; e += HF_THREADS) {
    const int tp = e % 9, cell = e / 9;
    float acc = 0.f;
    for (int ci = 0; ci < 8; ci++) acc = fmaf(p.w4raw[tp * 8 + ci], u3l[cell * 8 + ci], acc);
    p.vfr[(size_t)s * 7200 + e] = acc;
  }

  // ---- corrections of the heat-map frame pixels: the taps of upconv4 that fall into the zero padding, evaluated
  // on the clamp-extended up-sampled plane the phase form sees there ----
  for (int e = tid; e < 4 * 400; e += HF_THREADS) {
    const int ln = e / 400, t = e - ln * 400, isc = ln >> 1, side = ln & 1;
    float acc = 0.f;
    if (!isc || (t > 0 && t < 399)) {  // the corner pixels are counted with the row lines
      const int d0 = side ? 2 : 0;      // the tap row (column) outside the image
#pragma unroll
      for (int d1 = 0; d1 < 3; d1++)
#pragma unroll
        for (int ci = 0; ci < 8; ci++) {
          const int tap = isc ? d1 * 3 + d0 : d0 * 3 + d1;
          acc += p.w4raw[tap * 8 + ci] * hf_up(u3l + ln * 1600 + ci, 8, 200, t + d1 - 1, p.legacy);
        }
      if (!isc && (t == 0 || t == 399)) {  // corner: the two taps of the column outside the image, rows inside
        const int cl = t ? 3 : 2, dx = t ? 2 : 0;
        for (int dy = 0; dy < 3; dy++) {
          if (dy == d0) continue;            // already counted with the row
          const int yy = (side ? 399 : 0) + dy - 1;
          if (yy < 0 || yy > 399) continue;  // cannot happen (dy == d0 covers it)
          for (int ci = 0; ci < 8; ci++) acc += p.w4raw[(dy * 3 + dx) * 8 + ci] * hf_up(u3l + cl * 1600 + ci, 8, 200, yy, p.legacy);
        }
      }
    }
    p.c4[(size_t)s * 1600 + e] = acc;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// k_head_frames: the same lines in PHASE FORM on the matrix cores.  A frame line of a layer's output is one phase
// row (column) of the low-resolution grid's first / last row (column), with the taps that fall into the zero padding
// left out of the pre-combined phase weights (PrepLayout::w2fr / w3fr): no up-sampled value is ever formed.
//   uprelu2: four 2-wide bands (rows {0,1}, {98,99}, columns {0,1}, {98,99}) = all four phases of the first / last
//            low-res row / column: 4 bands x 4 M-tiles of 16 pixels, K = 20, N = 16
//   uprelu3: the four frame lines = one phase row / column of the first / last quad row / column, from the uprelu2
//            bands: 4 lines x 7 M-tiles of 16 quads, K = 36, N = 2 parities x 8 channels
//   heat-map corrections: PrepLayout::efr along the uprelu3 lines (VALU, 24 MACs per pixel)
// The four corner cells of each plane lose taps in both directions: they are evaluated from the definition.
// ---------------------------------------------------------------------------------------------------------------
constexpr int HG_THREADS = 256;
constexpr int HG_L1 = 0;                     // l1p[2][52][52]: uprelu1, clamp-extended by one cell; dead once the bands stand
constexpr int HG_U3 = 0;                     // u3l[4][200][8]: over l1p
constexpr int HG_RB = 6400;                  // u2rb[4][102][4]: uprelu2 rows {0,1,98,99}, columns -1 .. 100 (clamp copies)
constexpr int HG_CB = HG_RB + 4 * 102 * 4;   // u2cb[4][102][4]: uprelu2 columns {0,1,98,99}, rows -1 .. 100
constexpr int HG_TOTAL = HG_CB + 4 * 102 * 4;  // 9664 floats = 38 KB: four workgroups per CU
static_assert(2 * 52 * 52 <= 6400, "l1p fits under u3l");

// zero-padded conv output at a plane corner from the definition: cell (yc, xc) in {0, n2 - 1}^2 of the up-sampled
// (n2 = 2 n) plane of `get(y, x, ci)` (low-res, n x n)
template <int CIN, int COUT, class G>
__device__ __forceinline__ float hg_corner(G get, int n, int yc, int xc, const float *w, float bias, int co, int legacy) {
  float acc = bias;
  for (int dy = 0; dy < 3; dy++) {
    const int uy = yc + dy - 1;
    if (uy < 0 || uy >= 2 * n) continue;
    int ya, yb; float wy;
    hf_taps(uy, n, ya, yb, wy, legacy);
    for (int dx = 0; dx < 3; dx++) {
      const int ux = xc + dx - 1;
      if (ux < 0 || ux >= 2 * n) continue;
      int xa, xb; float wx;
      hf_taps(ux, n, xa, xb, wx, legacy);
      for (int ci = 0; ci < CIN; ci++) {
        const float a = get(ya, xa, ci), b = get(ya, xb, ci), d = get(yb, xa, ci), g = get(yb, xb, ci);
        const float top = a + (b - a) * wx, bot = d + (g - d) * wx;
        acc += w[((dy * 3 + dx) * CIN + ci) * COUT + co] * (top + (bot - top) * wy);
      }
    }
  }
  return fmaxf(acc, 0.f);
}

__global__ __launch_bounds__(HG_THREADS) void k_head_frames(HeadParams2 p) {
  __shared__ __align__(16) float sm[HG_TOTAL];
  __shared__ float efr_s[192], w4s[72];
  // the ship of a block rotates inside its group of 8 (blocks go round-robin over the 8 XCDs: a regular ship mask
  // must not leave all the live workgroups on one of them)
  const int s = hd_ship(p, (int)((blockIdx.x & ~7u) | ((blockIdx.x + (blockIdx.x >> 3)) & 7u)));
  const int tid = threadIdx.x, lane = tid & 63;
  if (s < 0) return;  // block-uniform
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kq = lane >> 4;
  float *l1p = sm + HG_L1, *u2rb = sm + HG_RB, *u2cb = sm + HG_CB, *u3l = sm + HG_U3;

  {  // uprelu1 -> LDS, clamp-extended by one cell.  Only the three outer rows / columns on each side are ever read (the
     // frame bands and the corner cells): 588 of the 2704 cells of a channel; all loads of a thread in flight before
     // the first store
    constexpr int NB = 6 * 52 + 46 * 6, NE = 2 * NB, PER = (NE + HG_THREADS - 1) / HG_THREADS;  // 588 per channel, 5 per thread
    float v[PER];
    int at[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) {
      const int e = min(u * HG_THREADS + tid, NE - 1);
      const int ci = e / NB, k = e - ci * NB;
      int r, c;
      if (k < 6 * 52) { const int rr = k / 52; r = rr < 3 ? rr : rr + 46; c = k - rr * 52; }
      else { const int k2 = k - 6 * 52, cc = k2 % 6; r = 3 + k2 / 6; c = cc < 3 ? cc : cc + 46; }
      at[u] = (ci * 52 + r) * 52 + c;
      v[u] = p.up1[(size_t)s * 5000 + (ci * 50 + min(max(r - 1, 0), 49)) * 50 + min(max(c - 1, 0), 49)];
    }
#pragma unroll
    for (int u = 0; u < PER; u++)
      if (u * HG_THREADS + tid < NE) l1p[at[u]] = v[u];
  }
  if (tid < 192) efr_s[tid] = p.efr[tid];
  if (tid < 72) w4s[tid] = p.w4raw[tid];
  __syncthreads();

  // ---- uprelu2 bands: band v, M-tile m: pixels 16 m + n16 along the band (50), all four phases ----
  // wave = band: its B operands are loaded once
  float bw2[5];
#pragma unroll
  for (int jj = 0; jj < 5; jj++) bw2[jj] = p.w2fr[(wv * 20 + 4 * jj + kq) * 16 + n16];
  float bw3[9];
#pragma unroll
  for (int jj = 0; jj < 9; jj++) bw3[jj] = p.w3fr[(wv * 36 + 4 * jj + kq) * 16 + n16];
#pragma unroll 1
  for (int m = 0; m < 4; m++) {
    const int v = wv;
    const int pp = min(16 * m + n16, 49);
    const int i = v == 0 ? 0 : v == 1 ? 49 : pp, j = v == 2 ? 0 : v == 3 ? 49 : pp;  // low-res pixel of the A row
    const float bias = p.b2[n16 & 3];
    f32x4 d = {bias, bias, bias, bias};
#pragma unroll
    for (int jj = 0; jj < 5; jj++) {  // k = 4 jj + kq: tap = 2 jj + (kq >> 1), ci = kq & 1 (k >= 18: zero weights)
      const int tap = min(2 * jj + (kq >> 1), 8);
      const float a = l1p[((kq & 1) * 52 + i + tap / 3) * 52 + j + tap % 3];
      d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw2[jj], d, 0, 0, 0);
    }
    // D: rows = pixels 16 m + 4 kq + r, column n16 = (phase a, b; channel)
    const int co = n16 & 3, pa = n16 >> 3, pb = (n16 >> 2) & 1;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int q = 16 * m + 4 * kq + r;
      if (q >= 50) continue;
      const float val = fmaxf(d[r], 0.f);
      if (v < 2) {  // rows 2 i + pa, column 2 q + pb
        u2rb[((2 * v + pa) * 102 + 2 * q + pb + 1) * 4 + co] = val;
      } else {      // column 2 j + pb, row 2 q + pa
        u2cb[((2 * (v - 2) + pb) * 102 + 2 * q + pa + 1) * 4 + co] = val;
      }
    }
  }
  __syncthreads();
  // the 2 x 2 corner blocks exist in both band sets and a band is wrong where the OTHER direction loses taps too:
  // (frame row, inner column) comes from the row band, (inner row, frame column) from the column band, the plane's
  // corner cell from the definition
  if (tid < 64) {
    const int co = tid & 3, cx = (tid >> 2) & 1, cy = (tid >> 3) & 1, bx = (tid >> 4) & 1, by = tid >> 5;
    const int y = by ? 98 + cy : cy, x = bx ? 98 + cx : cx;              // the cell
    const int rb = by ? 2 + cy : cy, cb = bx ? 2 + cx : cx;              // its band index in u2rb / u2cb
    const bool fy = y == 0 || y == 99, fx = x == 0 || x == 99;
    float val;
    if (fy && fx) {
      auto get = [&](int yy, int xx, int ci) { return l1p[(ci * 52 + yy + 1) * 52 + xx + 1]; };
      val = hg_corner<2, 4>(get, 50, y, x, p.w2raw, p.b2[co], co, p.legacy);
    } else if (fx) {
      val = u2cb[(cb * 102 + y + 1) * 4 + co];
    } else {
      val = u2rb[(rb * 102 + x + 1) * 4 + co];
    }
    // (the two reads above and the writes below touch different cells of the arrays for every thread pair: a cell is
    // read from the array it is right in and written to the other one; the corner is written to both)
    if (fy && fx) { u2rb[(rb * 102 + x + 1) * 4 + co] = val; u2cb[(cb * 102 + y + 1) * 4 + co] = val; }
    else if (fx) u2rb[(rb * 102 + x + 1) * 4 + co] = val;
    else u2cb[(cb * 102 + y + 1) * 4 + co] = val;
  }
  __syncthreads();
  if (tid < 128) {  // clamp copies at -1 / 100 of every band line
    const int co = tid & 3, b = (tid >> 2) & 3, hi = (tid >> 4) & 1, isc = tid >> 5;
    if (isc < 2) {
      float *ln = (isc ? u2cb : u2rb) + b * 102 * 4;
      ln[(hi ? 101 : 0) * 4 + co] = ln[(hi ? 100 : 1) * 4 + co];
    }
  }
  for (int e = tid; e < 4 * 100 * 4; e += HG_THREADS) {  // lines: row 0, row 99, col 0, col 99
    const int ln = e / 400, r = e - ln * 400;
    const float *src = ln < 2 ? u2rb + (ln ? 3 : 0) * 408 : u2cb + (ln == 3 ? 3 : 0) * 408;
    p.u2fr[(size_t)s * 1600 + e] = src[4 + r];
  }
  __syncthreads();

  // ---- uprelu3 frame lines: line v, M-tile m: quads 16 m + n16 along the line (100) ----
#pragma unroll 1
  for (int m = 0; m < 7; m++) {  // wave = line
    const int v = wv;
    const int qq = min(16 * m + n16, 99);
    const float *band = (v < 2 ? u2rb : u2cb);
    const int b0 = (v & 1) ? 2 : 0;              // the band pair {0,1} or {98,99}
    const float bias = p.b3[n16 & 7];
    f32x4 d = {bias, bias, bias, bias};
#pragma unroll
    for (int jj = 0; jj < 9; jj++) {            // k = 4 jj + kq: tap jj, ci = kq
      const int ty = jj / 3, tx = jj % 3;
      // across the line: the low-res rows (columns) -1 | 0 | 1 of the first, 98 | 99 | 100 of the last, clamped
      const int tf = v < 2 ? ty : tx, ta = v < 2 ? tx : ty;
      const int bsel = (v & 1) ? b0 + min(tf, 1) : b0 + max(tf - 1, 0);
      const float a = band[(bsel * 102 + qq + ta) * 4 + kq];
      d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw3[jj], d, 0, 0, 0);
    }
    const int co = n16 & 7, par = n16 >> 3;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int q = 16 * m + 4 * kq + r;
      if (q < 100) u3l[(v * 200 + 2 * q + par) * 8 + co] = fmaxf(d[r], 0.f);
    }
  }
  __syncthreads();
  if (tid < 32) {  // the four corner cells, from the definition, into both lines they belong to
    const int co = tid & 7, c = tid >> 3;
    const int y = (c & 1) ? 199 : 0, x = (c & 2) ? 199 : 0;
    auto get = [&](int yy, int xx, int ci) {  // uprelu2 near the corner: rows {0,1} / {98,99} of the row bands
      return u2rb[((yy < 50 ? yy : yy - 96) * 102 + xx + 1) * 4 + ci];
    };
    const float val = hg_corner<4, 8>(get, 100, y, x, p.w3raw, p.b3[co], co, p.legacy);
    u3l[(((c & 1) ? 1 : 0) * 200 + x) * 8 + co] = val;
    u3l[(((c & 2) ? 3 : 2) * 200 + y) * 8 + co] = val;
  }
  __syncthreads();
  // the tap planes along the four lines: V_t = sum_c w4[t][c] uprelu3_c (what k_head_stream's ring holds; the same
  // sequential sum over c as the 1x1's MFMA chain)
  // 252 threads = 28 cells x 9 taps per pass: a thread keeps ITS tap (its 8 weights in registers), consecutive lanes store
  // consecutive floats, a cell's channels arrive as two 16-byte LDS reads.  (r04 also tried a thread per cell with its nine taps
  // and the weights as scalar loads - every store instruction of a wave then strides over 36 bytes: 0.91 against 0.79 ms for the
  // kernel - and (cell, tap) = (e / 9, e % 9) over all 256 threads with the weights read from LDS per FMA: 0.73 ms;
  // tools/ab_frames.sh)
  if (tid < 252) {
    const int tp = tid % 9, c0 = tid / 9;
    float wt[8];
#pragma unroll
    for (int ci = 0; ci < 8; ci++) wt[ci] = w4s[tp * 8 + ci];
    float *dst = p.vfr + (size_t)s * 7200 + tid;
    for (int cell = c0; cell < 4 * 200; cell += 28) {
      const f32x4 ua = *reinterpret_cast<const f32x4 *>(u3l + cell * 8), ub = *reinterpret_cast<const f32x4 *>(u3l + cell * 8 + 4);
      float acc = 0.f;
#pragma unroll
      for (int ci = 0; ci < 4; ci++) acc = fmaf(wt[ci], ua[ci], acc);
#pragma unroll
      for (int ci = 0; ci < 4; ci++) acc = fmaf(wt[4 + ci], ub[ci], acc);
      dst[(cell - c0) * 9] = acc;   // element (cell, tp) = 9 cell + tp = tid + 9 (cell - c0)
    }
  }

  // ---- corrections of the heat-map frame pixels (PrepLayout::efr: the conv taps of the row / column outside the
  // image in phase form along the line): pixel 2 j + b of a line gets sum_o E[b][o] L[j + o - 1] ----
  for (int e = tid; e < 4 * 400; e += HG_THREADS) {
    const int ln = e / 400, t = e - ln * 400, isc = ln >> 1, side = ln & 1, jq = t >> 1, b = t & 1;
    float acc = 0.f;
    if (!isc || (t > 0 && t < 399)) {  // the corner pixels are counted with the row lines
      const float *E = efr_s + ((isc * 2 + side) * 2 + b) * 24;
      const float *L = u3l + ln * 1600;
#pragma unroll
      for (int o = 0; o < 3; o++) {
        const float *Lp = L + min(max(jq + o - 1, 0), 199) * 8;
#pragma unroll
        for (int ci = 0; ci < 8; ci++) acc += E[o * 8 + ci] * Lp[ci];
      }
      if (!isc && (t == 0 || t == 399)) {
        // corner pixel: + the column line's sum at this row, minus the tap (outside row, outside column) it shares
        // with the row line
        const int cs = t ? 1 : 0;                       // left | right column line
        const int yq = side ? 199 : 0, a = side ? 1 : 0;  // low-res row and row phase of heat-map row 0 | 399
        const float *E2 = efr_s + ((2 + cs) * 2 + a) * 24;
        const float *L2 = u3l + (2 + cs) * 1600;
        for (int o = 0; o < 3; o++) {
          const float *Lp = L2 + min(max(yq + o - 1, 0), 199) * 8;
          for (int ci = 0; ci < 8; ci++) acc += E2[o * 8 + ci] * Lp[ci];
        }
        const int tap = (side ? 2 : 0) * 3 + (cs ? 2 : 0);
        const float *Lc = L + (t ? 199 : 0) * 8;        // the corner cell of uprelu3
        for (int ci = 0; ci < 8; ci++) acc -= p.w4raw[tap * 8 + ci] * Lc[ci];
      }
    }
    p.c4[(size_t)s * 1600 + e] = acc;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// k_head_stream
// ---------------------------------------------------------------------------------------------------------------
constexpr int HS_GPR = 13;               // groups of 4 quads per quad row: 50 quads + 1 halo + 1 pad
constexpr int HS_NTILES = 325;           // 16-quad M-tiles of a strip (100 quad rows x 13 groups / 4)
constexpr int HS_NS = 42;                // sub-steps
constexpr int HS_NR3 = 16, HS_P3 = 106;  // tap-plane ring (one row per uprelu3 row): rows, row pitch (column x at index x - c0 + 1 + 4 side)
constexpr int HS_PL3 = HS_NR3 * HS_P3 + 4;   // plane stride (floats)
constexpr int HS_NV = 9;                 // tap planes V_t, t = 3 ky + kx
// uprelu2 ring (column c at index c - (side ? 46 : -1)): 16 row slots + 2 mirror slots (16, 17 repeat 0, 1), so the
// three rows of a window are always contiguous and the gather of stage B is one base address + immediates
constexpr int HS_NR2 = 16, HS_P2 = 56;
constexpr int HS_PL2 = (HS_NR2 + 2) * HS_P2;  // 1008 = 16 mod 32: the channel planes of the MFMA gather land on different banks
static_assert(HS_PL2 % 32 == 16, "uprelu2 plane stride");
// group (flat over quad rows) -> its offsets in the two rings: both repeat every 16 quad rows = 208 groups, and the four
// groups of M-tile T are entries 4 (T mod 52) .. + 3
constexpr int HS_TABP = 16 * HS_GPR;
static_assert(HS_TABP == 4 * 52, "table period in tiles");
// LDS of a workgroup: two of them per CU (160 KB)
static_assert(4 * (HS_NV * HS_PL3 + 4 * HS_PL2 + HS_TABP) <= 80 * 1024, "two workgroups per CU");
// uprelu2 row pairs finished by the end of sub-step s - 1 (0 = by the prologue): the table of tools/head_schedule.py,
// 2, 3, 5, 6, 7, 8, 10, ...: min(50, s + 2 + ((s + 2) >> 2)) (hs_pairs_done_c below)
// Stage-A tile of consumer wave cw in sub-step st (uprelu2 rows for the M-tiles of st + 1): tile id 2 pair + half belongs
// to wave id & 3, so the half is cw & 1 and the table holds the row pair (-1 = none).  A compile-time table in constant
// memory: one scalar load + one bit-field extract instead of ~35 scalar instructions per sub-step.
struct alignas(16) HsPick { signed char v[HS_NS][4]; };
constexpr int hs_pairs_done_c(int s) { return s + 2 + ((s + 2) >> 2) < 50 ? s + 2 + ((s + 2) >> 2) : 50; }
constexpr HsPick hs_make_pick() {
  HsPick t{};
  for (int st = 0; st < HS_NS; st++)
    for (int cw = 0; cw < 4; cw++) {
      const int lo = 2 * hs_pairs_done_c(st), hi = 2 * hs_pairs_done_c(st + 1);
      const int at = lo + ((cw - lo) & 3);
      t.v[st][cw] = (signed char)(at < hi ? at >> 1 : -1);
    }
  return t;
}
__constant__ HsPick kHsPick = hs_make_pick();

__device__ __forceinline__ int hs_tiles_done(int s) {  // tiles finished by the end of sub-step s
  return s < 0 ? 0 : min(HS_NTILES, 8 * (s + 1) + ((s + 1) >> 3));
}

// LDS reads of the stencil: two 8-byte reads of one row.  volatile keeps them apart (merged into one ds_read2_b64 they
// take 8 LDS cycles instead of 2 + 2, MI355X_MICROARCH.md) and in program order (the pipeline below is explicit)
typedef const volatile __attribute__((address_space(3))) f32x2 hs_lds_v2;

// uprelu2 store: row slot + its mirror
__device__ __forceinline__ void hs_u2_store(float *u2r, int ch, int row, int col, float v) {
  const int slot = (row + 1) & (HS_NR2 - 1);
  float *q = &u2r[ch * HS_PL2 + slot * HS_P2 + col];
  q[0] = v;
  if (slot < 2) q[HS_NR2 * HS_P2] = v;
}

// x2 bilinear: output row 2 i + a, tap k in {0, 1, 2} = up-res row 2 i + a + k - 1, weight of low-res row i + t - 1
// (k_policy_prepare's up_coef).  The set of non-zero (k, t) is the half-pixel one under both conventions.
constexpr bool HS_NZ[2][3][3] = {{{1, 1, 0}, {1, 1, 0}, {0, 1, 1}}, {{1, 1, 0}, {0, 1, 1}, {0, 1, 1}}};
constexpr float HS_CH[2][3][3] = {{{0.75f, 0.25f, 0.f}, {0.25f, 0.75f, 0.f}, {0.f, 0.75f, 0.25f}},
                                  {{0.25f, 0.75f, 0.f}, {0.f, 0.75f, 0.25f}, {0.f, 0.25f, 0.75f}}};
constexpr float HS_CL[2][3][3] = {{{0.5f, 0.5f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.5f, 0.5f}},
                                  {{0.f, 1.f, 0.f}, {0.f, 0.5f, 0.5f}, {0.f, 0.f, 1.f}}};
static_assert(HS_CH[0][0][0] == HS_CH[0][2][1] && HS_CH[0][0][1] == HS_CH[0][2][2] && HS_CH[1][0][0] == HS_CH[1][2][1] &&
              HS_CH[1][0][1] == HS_CH[1][2][2] && HS_CL[0][0][0] == HS_CL[0][2][1] && HS_CL[0][0][1] == HS_CL[0][2][2] &&
              HS_CL[1][0][0] == HS_CL[1][2][1] && HS_CL[1][0][1] == HS_CL[1][2][2], "the stencil shares the sums of the outer taps");
// the (k, t) pairs either output parity needs, in read order: 7 of the 9
constexpr int HS_NKT = 7;
constexpr int HS_KT[HS_NKT][2] = {{0, 0}, {0, 1}, {1, 0}, {1, 1}, {1, 2}, {2, 1}, {2, 2}};
__device__ __forceinline__ f32x2 hs_fma2(float c, f32x2 v, f32x2 z) { return __builtin_elementwise_fma((f32x2){c, c}, v, z); }

constexpr int HS_NB = 4;                 // producer waves (two M-tiles per sub-step each); 4 consumer waves behind them
constexpr int HS_THREADS = 64 * (HS_NB + 4);

// EXTRA: the launch wants the heat map and / or a probed value written out (ofx_policy_forward with a heatmap pointer,
// ofx_policy_forward_obs with a probe: the DQN targets); the rollout's forward does not, and its instantiation carries
// neither the tests nor the branches of those paths in the consumers' loop
// LP != 0 (OFX_OPT_POLICY_BF16, opt-in; 1 = bf16, 2 = fp16 operands): stage B on v_mfma_f32_16x16x16_bf16 / _f16 (K = 36 ->
// 3 MFMAs of K = 16 per channel half instead of 9 of K = 4): the operands - uprelu2 values read from the fp32 ring and
// the phase weights - are rounded on the way into the matrix instruction, the sums stay fp32.  The 1x1, the stencil,
// rings, schedule, frame lines (exact fp32) and stage A are the fp32 kernel's.
template <bool EXTRA, int LP>
__global__ __launch_bounds__(HS_THREADS, 4) void k_head_stream(HeadParams2 p) {
  constexpr bool BF16 = LP != 0;  // reduced-precision operands (1 = bf16, 2 = fp16), fp32 sums
  __shared__ __align__(16) float vr[HS_NV * HS_PL3];
  __shared__ __align__(16) float u2r[4 * HS_PL2];
  // lo 16 bits: float offset of the group's first quad in the uprelu2 ring (row slot of quad row - 1); hi: in a tap
  // plane (row slot of uprelu3 row 2 * quad row), bit 31: the group holds the strip's frame column
  __shared__ unsigned tabG[HS_TABP];
  // blocks b and b + 8 (same XCD under round-robin placement) are the two halves of one ship; the ship of a block
  // rotates with the group of 16 blocks, so that a regular ship mask (say the first ship of every arena: the
  // reference's own line-up has one policy ship) does not put all the live workgroups on one XCD
  const int blk = blockIdx.x;
  const int s = hd_ship(p, (blk >> 4) * 8 + ((blk + (blk >> 4)) & 7)), side = (blk >> 3) & 1;
  if (s < 0) return;  // block-uniform
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int g = tid; g < HS_TABP; g += HS_THREADS) {
    const int q = g / HS_GPR, gg = g - HS_GPR * q;
    const unsigned lo = (unsigned)((q & (HS_NR2 - 1)) * HS_P2 + 4 * gg);
    const unsigned hi = (unsigned)(((2 * q) & (HS_NR3 - 1)) * HS_P3 + 8 * gg) | ((gg == (side ? 12 : 0)) ? 0x8000u : 0u);
    tabG[g] = lo | (hi << 16);
  }
  __syncthreads();  // the producers read the table in front of their first barrier

#if OFX_HEAD_HOOKS
  unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0}, stamp_last = __builtin_amdgcn_s_memtime();
#define HS_STAMP(i) do { if (p.dbg) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp_acc[i] += t_ - stamp_last; stamp_last = t_; } } while (0)
#define HS_STAMP_OUT() do { if (p.dbg && lane == 0) for (int i_ = 0; i_ < 6; i_++) p.dbg[((size_t)blk * (HS_NB + 4) + wv) * 6 + i_] = stamp_acc[i_]; } while (0)
#else
#define HS_STAMP(i) do { } while (0)
#define HS_STAMP_OUT() do { } while (0)
#endif
  // Both roles run the same number of barriers (1 + HS_NS); each has its own loop so that the register allocator
  // sees only one role's long-lived state at a time.
  if (wv < HS_NB) {
    // ================================================================ producer waves (stage B + the 1x1)
    // lane = (quad n16 of the M-tile, output parity kq = (pa, pb)): ONE uprelu3 pixel, all 8 channels
    const int n16 = lane & 15, kq = lane >> 4;
    const int qd = n16 & 3, lg = n16 >> 2, pa = kq >> 1, pb = kq & 1;
    float bw[2][9];          // A operands W[half][k = 4 j + kq][m = n16 = phase * 4 + channel], constant over the strip
    f32x4 binit3[2];         // the folded bias as the C operand of a tile's first MFMA (row m = 4 kq + i: channel 4 half + i)
#pragma unroll
    for (int hf = 0; hf < 2; hf++) {
#pragma unroll
      for (int j = 0; j < 9; j++) bw[hf][j] = p.w3mf[(hf * 36 + 4 * j + kq) * 16 + n16];
      binit3[hf] = (f32x4){p.b3[4 * hf], p.b3[4 * hf + 1], p.b3[4 * hf + 2], p.b3[4 * hf + 3]};
    }
    // the 1x1: lane L holds w4[tap 4 g + (L & 3)][channel c] for (g, c) = L >> 2 = 8 g + c (taps 0 - 7); the MFMA for (g, c)
    // picks its 4-lane block with ABID and broadcasts it (CBSZ = 4)
    float wq[1];
    {
      const int idx = lane >> 2, tap = 4 * (idx >> 3) + (lane & 3);
      wq[0] = p.w4raw[tap * 8 + (idx & 7)];
    }
    // the ninth tap would fill one row of a third group of four: 8 scalar-weight FMAs on the vector ALU instead of 8 MFMAs
    float w8[8];
#pragma unroll
    for (int c = 0; c < 8; c++) w8[c] = p.w4raw[64 + c];
    const float *vf = p.vfr + (size_t)s * 7200;
    // BF16: A operand of MFMA J of half hf = W[k = 16 J + 4 kq + i][n16], i = 0..3 (k >= 36: zero); the B operand is the
    // four channels of tap 4 J + kq at the lane's quad: one address per J (lane-constant tap offset), channel planes
    // by immediate
    lp_x4 bwb[2][3];
    int toff[3];
    if constexpr (BF16) {
#pragma unroll
      for (int J = 0; J < 3; J++) {
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
          float w[4];
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const int k = 16 * J + 4 * kq + i;
            w[i] = k < 36 ? p.w3mf[(hf * 36 + k) * 16 + n16] : 0.f;
          }
          bwb[hf][J] = lp_pk4<LP ? LP : 1>(w[0], w[1], w[2], w[3]);
        }
        const int tap = min(4 * J + kq, 8);
        toff[J] = (tap / 3) * HS_P2 + tap % 3 - kq * HS_PL2;   // t.a carries the f32 kernel's + kq * HS_PL2: taken out
      }
    }

    // ---- M-tile T = groups 4 T .. 4 T + 3 (flat over quad rows), 16 quads x 4 parities = 64 uprelu3 pixels ----
    struct TileB {
      const float *a;   // gather: cell (row qi - 1, column of the lane's quad - 1) of channel kq; rows + P2, columns + 1
      float *w;         // the lane's pixel in tap plane 0
      bool fcol;        // wave-uniform: one of the tile's groups holds the strip's frame column
      bool frow;        // wave-uniform: the tile has groups in the first / last quad row
      int fq;           // wave-uniform: the quad row of the frame-column group
      float fv;         // lanes 0-17: exact frame-column value (row parity lane / 9, tap plane lane % 9), requested a sub-step ahead
      unsigned e;       // the lane group's own table entry
      int T;
    };
    const float *const a_lane = &u2r[kq * HS_PL2 + qd + side];
    float *const w_lane = &vr[pa * HS_P3 + 2 * qd + pb + 1];
    const unsigned *const tab_lane = &tabG[lg];
    auto tile_setup = [&](int T, TileB &t) {  // T < HS_NTILES
      const int Tm = T - 52 * ((T * 1261) >> 16);   // T mod 52 (wave-uniform)
      const unsigned e = tab_lane[4 * Tm];
      t.e = e;
      t.T = T;
      t.a = a_lane + (e & 0xFFFFu);
      t.w = w_lane + ((e >> 16) & 0x3FFFu);
      t.frow = 4 * T < HS_GPR || 4 * T + 3 >= 99 * HS_GPR;
      const unsigned long long fm = __builtin_amdgcn_ballot_w64((e >> 31) != 0);
      t.fcol = fm != 0;
      t.fq = 0;
      t.fv = 0.f;
      if (fm != 0) {
        // the zero padding differs from the clamp-extended phase form on the plane's frame: those cells take the exact
        // values of k_head_frames' lines.  The strip's frame column crosses the tile in ONE group: 2 rows x 9 planes
        const int fl = (int)__builtin_ctzll(fm);                      // a lane of that group (n16 = 4 lg + ..: lanes 4 lg .. 4 lg + 3)
        const int g = 4 * T + ((fl & 15) >> 2);
        t.fq = (g * 5042) >> 16;
        const int ln = min(lane, 17), y = 2 * t.fq + ln / 9;
        t.fv = vf[((side ? 3 : 2) * 200 + y) * 9 + ln % 9];
      }
    };
    // ReLU -> the 9 tap values of the lane's pixel -> the ring
    auto tile_tail = [&](const TileB &t, const f32x4 d0, const f32x4 d1) {
      float u[8];
#pragma unroll
      for (int i = 0; i < 4; i++) { u[i] = hd_max_raw(d0[i], 0.f); u[4 + i] = hd_max_raw(d1[i], 0.f); }
      f32x4 v[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      float v8 = 0.f;
#pragma unroll
      for (int c = 0; c < 8; c++) {
#pragma unroll
        for (int g = 0; g < 2; g++) {
          const int idx = 8 * g + c;
          switch (idx) {
#define HS_CASE(B) case B: v[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(wq[0], u[c], v[g], 4, B, 0); break;
            HS_CASE(0) HS_CASE(1) HS_CASE(2) HS_CASE(3) HS_CASE(4) HS_CASE(5) HS_CASE(6) HS_CASE(7)
            HS_CASE(8) HS_CASE(9) HS_CASE(10) HS_CASE(11) HS_CASE(12) HS_CASE(13) HS_CASE(14) HS_CASE(15)
#undef HS_CASE
          }
        }
        v8 = fmaf(w8[c], u[c], v8);
      }
#pragma unroll
      for (int tp = 0; tp < 8; tp++) t.w[tp * HS_PL3] = v[tp >> 2][tp & 3];
      t.w[8 * HS_PL3] = v8;
    };
    // exact frame cells of the tap planes (and the clamp copies around the plane) behind a tile that touches the frame:
    // same wave, so the LDS writes land behind the tile's own
    auto tile_frames = [&](const TileB &t) {
      if (t.fcol) {  // the strip's frame column, 2 rows x 9 planes on lanes 0-17, + the clamp column
        if (lane < 18) {
          const int yp = lane / 9, tp = lane - 9 * yp, y = 2 * t.fq + yp;
          const float v = t.fv;
          const int cf = side ? 104 : 1, cc = side ? 105 : 0;
          float *q = &vr[tp * HS_PL3 + (y & (HS_NR3 - 1)) * HS_P3];
          q[cf] = v; q[cc] = v;
          if (y == 0) { float *q2 = &vr[tp * HS_PL3 + (HS_NR3 - 1) * HS_P3]; q2[cf] = v; q2[cc] = v; }             // row -1
          if (y == 199) { float *q2 = &vr[tp * HS_PL3 + (200 & (HS_NR3 - 1)) * HS_P3]; q2[cf] = v; q2[cc] = v; }  // row 200
        }
      }
      // frame row of the plane over the lane group's 8 columns x 9 planes (and its clamp copy): the 16 lanes of the group
      if (!t.frow) return;
      const int g = 4 * t.T + lg, qi = (g * 5042) >> 16, gg = g - HS_GPR * qi;
      if (qi == 0 || qi == 99) {
        const int y = qi ? 199 : 0, yc = qi ? 200 : -1;
        for (int i = 4 * qd + kq; i < 72; i += 16) {
          const int c = i / 9, tp = i - 9 * c;
          const int ci3 = 8 * gg + 1 + c, x = ci3 - 1 - 4 * side + 100 * side;
          const float v = vf[((qi ? 1 : 0) * 200 + min(x, 199)) * 9 + tp];
          vr[tp * HS_PL3 + (y & (HS_NR3 - 1)) * HS_P3 + ci3] = v;
          vr[tp * HS_PL3 + (yc & (HS_NR3 - 1)) * HS_P3 + ci3] = v;
        }
      }
    };
    auto tile_bf16 = [&](const TileB &t) {
      lp_x4 A[3];
#pragma unroll
      for (int J = 0; J < 3; J++) {
        const float *q = t.a + toff[J];
        A[J] = lp_pk4<LP ? LP : 1>(q[0], q[HS_PL2], q[2 * HS_PL2], q[3 * HS_PL2]);
      }
      f32x4 d0 = binit3[0], d1 = binit3[1];
#pragma unroll
      for (int J = 0; J < 3; J++) {
        d0 = lp_mfma16<LP ? LP : 1>(bwb[0][J], A[J], d0);
        d1 = lp_mfma16<LP ? LP : 1>(bwb[1][J], A[J], d1);
      }
      tile_tail(t, d0, d1);
      if (t.fcol || t.frow) tile_frames(t);
    };
    auto run_pair = [&](const TileB &t0, const TileB &t1) {
      if constexpr (BF16) { tile_bf16(t0); tile_bf16(t1); return; }
      // all 18 gathers requested up front; two accumulator chains per tile (64 cycles between dependent MFMAs, the
      // 16x16x4 latency is 40); the 1x1 of the first tile follows the second tile's MFMAs
      float a0[9], a1[9];
#pragma unroll
      for (int j = 0; j < 9; j++) { a0[j] = t0.a[(j / 3) * HS_P2 + j % 3]; a1[j] = t1.a[(j / 3) * HS_P2 + j % 3]; }
      f32x4 d00, d01, d10, d11;
#pragma unroll
      for (int j = 0; j < 9; j++) {
        d00 = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[0][j], a0[j], j ? d00 : binit3[0], 0, 0, 0);
        d01 = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[1][j], a0[j], j ? d01 : binit3[1], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 9; j++) {
        d10 = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[0][j], a1[j], j ? d10 : binit3[0], 0, 0, 0);
        d11 = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[1][j], a1[j], j ? d11 : binit3[1], 0, 0, 0);
      }
      tile_tail(t0, d00, d01);
      tile_tail(t1, d10, d11);
      if (t0.fcol || t0.frow) tile_frames(t0);
      if (t1.fcol || t1.frow) tile_frames(t1);
    };
    auto run_single = [&](const TileB &t0) {
      if constexpr (BF16) { tile_bf16(t0); return; }
      f32x4 d00, d01;
#pragma unroll
      for (int j = 0; j < 9; j++) {
        const float a0 = t0.a[(j / 3) * HS_P2 + j % 3];
        d00 = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[0][j], a0, j ? d00 : binit3[0], 0, 0, 0);
        d01 = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[1][j], a0, j ? d01 : binit3[1], 0, 0, 0);
      }
      tile_tail(t0, d00, d01);
      if (t0.fcol || t0.frow) tile_frames(t0);
    };

    // every global load so far (the weights) has landed: without this the compiler's wait-count bookkeeping carries
    // the weight loads into the loop as "maybe pending" and its in-order vmcnt waits then block on the loads of the
    // loop itself (vmcnt(0); expcnt / lgkmcnt fields left at their maxima)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    // Software pipeline over the sub-steps: the tile set-ups (table reads) and the frame-cell loads of the NEXT
    // sub-step's M-tiles are requested at the end of a sub-step, in front of the barrier; nothing is loaded in front of
    // its use, so every vmcnt wait names loads a whole sub-step old.  (vmcnt counts in order: one late load in front of
    // a wait exposes the full HBM latency, ~1 us, per sub-step.)
    TileB ta, tb;
    auto prepare = [&](int st) {  // for sub-step st
      const int t0 = hs_tiles_done(st - 1), n = hs_tiles_done(st) - t0;
      tile_setup(n > 0 ? t0 + wv : HS_NTILES - 1, ta);
      tile_setup(n > 0 ? t0 + wv + 4 : HS_NTILES - 1, tb);
    };
    prepare(0);
    __syncthreads();  // the consumers' prologue (uprelu2 row pairs 0, 1)
#pragma unroll 1
    for (int st = 0; st < HS_NS; st++) {
#if OFX_HEAD_HOOKS
      if (p.ablate & 2) { __syncthreads(); continue; }
#endif
      const int t0 = hs_tiles_done(st - 1), n = hs_tiles_done(st) - t0;
      HS_STAMP(0);
      if (n > 0) {
        run_pair(ta, tb);
        HS_STAMP(1);
        if (n == 9 && wv == 0) {  // every eighth sub-step: a ninth tile, set up on the spot
          tile_setup(t0 + 8, ta);
          run_single(ta);
        }
      }
      HS_STAMP(2);
      prepare(min(st + 1, HS_NS - 1));
      HS_STAMP(4);
      __syncthreads();
      HS_STAMP(5);
    }
    HS_STAMP_OUT();
  } else {
    // ================================================================ consumer waves (stage A + stencil + arg-max)
    // Every VALU instruction here is paid in matrix throughput (the f32 MFMAs and the VALU share the SIMD's issue
    // slots), so the per-pass bookkeeping is kept to running counters and the arg-max to a snapshot of the best pass.
    const int n16 = lane & 15, kq = lane >> 4;
    const int ph = n16 >> 2, cl = n16 & 3, pa = ph >> 1, pb = ph & 1;
    const int cw = wv - HS_NB;
    float bw2[5];
#pragma unroll
    for (int j = 0; j < 5; j++) bw2[j] = p.w2mf[(4 * j + kq) * 16 + n16];
    const float bias2 = p.b2[cl];
    const float *up1s = p.up1 + (size_t)s * 5000;
    const float *fr2 = p.u2fr + (size_t)s * 1600;
    // ---- stage A (the consumers open a sub-step with it): one M-tile = 16 uprelu1 pixels of row pair `pr` -> 2 x 32 cells x 4 channels of uprelu2 ----
    // A[pixel][k = 4 j + kq]: tap = 2 j + (kq >> 1), ci = kq & 1 (K = 18, padded to 20)
    // (kept as lambdas over the consumer's locals: the same code as a struct with init/load/compute members compiles to
    // 114 instead of 122 VGPRs and a different schedule, and measures 15.05 against 13.63 ms in one call - r04,
    // tools/ab_so.sh; stage A on the producer waves: 15.85 ms)
    auto stageA_load = [&](int pr, int hh, float *av) {
      const int px0 = side ? (hh ? 34 : 23) : (hh ? 11 : 0);
      const int j1 = px0 + n16;
#pragma unroll
      for (int j = 0; j < 5; j++) {
        const int tap = min(2 * j + (kq >> 1), 8);
        const int yy = min(max(pr + tap / 3 - 1, 0), 49), xx = min(max(j1 + tap % 3 - 1, 0), 49);
        av[j] = up1s[((kq & 1) * 50 + yy) * 50 + xx];
      }
      // the exact frame-column cell lanes 0-7 write behind the tile (2 rows x 4 channels), requested with the rest: a
      // load in front of its use would expose the HBM latency in every second stage-A tile.
      // (r03: lane-constant column offsets + a wave-uniform row base take these six loads' ~18 address instructions
      // away and the kernel got SLOWER, 15.5 against 15.3 ms in one call - tools/ab_so.sh; left as the compiler has it)
      av[5] = fr2[((side ? 3 : 2) * 100 + 2 * pr + ((lane >> 2) & 1)) * 4 + (lane & 3)];
    };
    auto stageA_compute = [&](int pr, int hh, const float *av) {
      const int px0 = side ? (hh ? 34 : 23) : (hh ? 11 : 0);
      f32x4 d = {bias2, bias2, bias2, bias2};
#pragma unroll
      for (int j = 0; j < 5; j++) d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bw2[j], d, 0, 0, 0);
      // D: rows = pixels px0 + 4 kq + i, column n16 = (phase, channel)
      const int col = 2 * (px0 + 4 * kq) + pb - (side ? 46 : -1);
      float *w = &u2r[cl * HS_PL2 + ((2 * pr + pa + 1) & (HS_NR2 - 1)) * HS_P2 + col];
      const bool mir = ((2 * pr + pa + 1) & (HS_NR2 - 1)) < 2;  // rows 2 pr, 2 pr + 1 sit in slots 0 / 1 when pr % 8 is 7 / 0
#pragma unroll
      for (int i = 0; i < 4; i++) w[2 * i] = hd_max_raw(d[i], 0.f);
      if ((pr & 7) == 0 || (pr & 7) == 7) {  // wave-uniform
        if (mir) {
#pragma unroll
          for (int i = 0; i < 4; i++) w[HS_NR2 * HS_P2 + 2 * i] = hd_max_raw(d[i], 0.f);
        }
      }
      // exact frame cells + the clamp copies outside the plane (same wave: LDS operations of a wave stay in order)
      if ((side == 0 && hh == 0) || (side == 1 && hh == 1)) {  // the strip's frame column: 2 rows x 4 channels
        if (lane < 8) {
          const int yp = lane >> 2, ch = lane & 3, y = 2 * pr + yp;
          const float v = av[5];
          const int cf = side ? 53 : 1, cc = side ? 54 : 0;
          hs_u2_store(u2r, ch, y, cf, v);
          hs_u2_store(u2r, ch, y, cc, v);
          if (y == 0) { hs_u2_store(u2r, ch, -1, cf, v); hs_u2_store(u2r, ch, -1, cc, v); }
          if (y == 99) { hs_u2_store(u2r, ch, 100, cf, v); hs_u2_store(u2r, ch, 100, cc, v); }
        }
      }
      if (pr == 0 || pr == 49) {  // frame row of the plane (and its clamp copy) over the tile's 32 columns
        const int c = lane >> 1, chh = lane & 1, x = 2 * px0 + c;
        const int y = pr ? 99 : 0, yc = pr ? 100 : -1;
        const f32x2 v = *reinterpret_cast<const f32x2 *>(fr2 + ((pr ? 1 : 0) * 100 + x) * 4 + 2 * chh);
        const int ci2 = x - (side ? 46 : -1);
#pragma unroll
        for (int k = 0; k < 2; k++) {
          hs_u2_store(u2r, 2 * chh + k, y, ci2, v[k]);
          hs_u2_store(u2r, 2 * chh + k, yc, ci2, v[k]);
        }
      }
    };

    const int task = 64 * (wv - HS_NB) + lane;       // 250 two-pixel tasks per sub-step: 5 rows x 50
    const bool task_ok = task < 250;
    const int tk = task_ok ? task : 249;
    const int r_in = tk / 50, jx = tk - 50 * r_in;   // row of the block, pixel pair of the row
    const float bias4 = p.b4[0];
    // coefficients of the x2 bilinear (wave-uniform: scalar registers)
    float cy[2][3][3];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int k = 0; k < 3; k++)
#pragma unroll
        for (int t = 0; t < 3; t++) cy[a][k][t] = p.legacy ? HS_CL[a][k][t] : HS_CH[a][k][t];
    const bool fcol = task_ok && (side ? jx == 49 : jx == 0);  // the lane owns pixels of the strip's frame column
    const float *c4s = p.c4 + (size_t)s * 1600;
    const int x0 = 100 * side + 2 * jx;              // uprelu3 column of the lane's first pixel
    // running per-lane state: R = uprelu3 row of the lane's pixels in this sub-step (+5 per sub-step); q[dy] = byte
    // offset of ring row R + dy - 1 at the lane's column (+5 rows per sub-step, wrapped at 16 rows)
    constexpr unsigned PB = HS_P3 * 4, RING = HS_NR3 * PB;
    const unsigned lcol = (unsigned)(2 * jx + 4 * side) * 4;
    int R = r_in - 8;
    unsigned q[3];
#pragma unroll
    for (int dy = 0; dy < 3; dy++) q[dy] = (unsigned)((R + dy - 1) & (HS_NR3 - 1)) * PB + lcol;
    const unsigned qlim = RING + lcol;
    int coff = (side ? 3 : 2) * 400 + 2 * R;         // frame-column correction of the lane's two heat-map rows
    float tv = -INFINITY;
    int tst = 0;
    f32x4 snap[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};  // the 8 values of the pass that holds the lane's maximum
    const char *const vb = reinterpret_cast<const char *>(vr);

    {  // prologue: uprelu2 row pairs 0 and 1, one tile per consumer wave
      float av0[6];
      stageA_load(cw >> 1, cw & 1, av0);
      stageA_compute(cw >> 1, cw & 1, av0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // see the producers
    // stage-A tile of this wave in sub-step st (uprelu2 rows for the M-tiles of st + 1): tile id 2 pair + half belongs to
    // wave id & 3; its uprelu1 loads are requested one sub-step ahead
    float av[6];
    int apr, ahh;
    auto stageA_pick = [&](int st) {
      const int row = reinterpret_cast<const int *>(&kHsPick.v[0][0])[st];   // wave-uniform: a scalar load
      apr = (int)(signed char)(row >> (8 * cw));
      ahh = cw & 1;
      stageA_load(max(apr, 0), ahh, av);
    };
    stageA_pick(0);
    __syncthreads();
#pragma unroll 1
    for (int st = 0; st < HS_NS; st++) {
      const bool ok = task_ok && (unsigned)R < 200u;
#if OFX_HEAD_HOOKS
      if (p.ablate & 1) { __syncthreads(); continue; }
#endif
      HS_STAMP(0);
      // The consumers open a sub-step with stage A for the NEXT sub-step's M-tiles and the loads of the one after.
      // Nothing inside a sub-step orders stage A against the pass: uprelu2 rows written in sub-step st are read by the
      // producers in st + 1 (tools/head_schedule.py).
      if (apr >= 0) stageA_compute(apr, ahh, av);
      stageA_pick(min(st + 1, HS_NS - 1));
      HS_STAMP(3);
      if (__builtin_amdgcn_ballot_w64(ok) != 0) {    // wave-uniform
        // zero-padding corrections of frame pixels (global memory: requested up front, zero for the other lanes)
        f32x2 ccol = {0.f, 0.f};
        f32x4 crow = {0.f, 0.f, 0.f, 0.f};
        if (fcol && ok) ccol = *reinterpret_cast<const f32x2 *>(c4s + coff);
        const bool edge = st < 3 || st > 38;         // wave-uniform: the block may hold row 0 or row 199
        if (edge && ok && (R == 0 || R == 199)) crow = *reinterpret_cast<const f32x4 *>(c4s + (R ? 1 : 0) * 400 + 2 * x0);
        // ---- the stencil.  Vertical first: Z[a][kx] = sum_(ky, ty) cy[a][ky][ty] V_(ky, kx)[R + ty - 1] on the four
        // columns x0 - 1 .. x0 + 2 the lane reads (two pairs); one (tap column, column pair) at a time, the 7 reads of the
        // next one requested in front of the 12 packed FMAs of this one
        f32x2 Z[2][3][2];
        f32x2 V[2][HS_NKT];
        auto ldv = [&](int u) {   // unit u = (tap column kx = u >> 1, column pair u & 1): 7 reads
#pragma unroll
          for (int n = 0; n < HS_NKT; n++)
            V[u & 1][n] = *(hs_lds_v2 *)(vb + q[HS_KT[n][1]] + (3 * HS_KT[n][0] + (u >> 1)) * (HS_PL3 * 4) + 8 * (u & 1));
        };
        ldv(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 6; u++) {
          if (u + 1 < 6) ldv(u + 1);
          // reads n = 0 .. 6: V0[R-1], V0[R], V1[R-1], V1[R], V1[R+1], V2[R], V2[R+1] (V_ky of this tap column).  The
          // first and the last tap row carry the same pair of coefficients one row apart (cy[a][0][0] = cy[a][2][1],
          // cy[a][0][1] = cy[a][2][2], both conventions), so their sums are shared by the two parities: 10 packed
          // operations per unit instead of 12
          const f32x2 *v = V[u & 1];
          const f32x2 s1 = v[0] + v[5], s2 = v[1] + v[6];
#pragma unroll
          for (int a = 0; a < 2; a++) {
            f32x2 z = (f32x2){cy[a][0][0], cy[a][0][0]} * s1;
            z = hs_fma2(cy[a][0][1], s2, z);
#pragma unroll
            for (int ty = 0; ty < 3; ty++)
              if (HS_NZ[a][1][ty]) z = hs_fma2(cy[a][1][ty], v[2 + ty], z);
            Z[a][u >> 1][u & 1] = z;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        HS_STAMP(1);
        // horizontal: out[a][b] of the pair (x0, x0 + 1) = bias + sum_(kx, tx) cy[b][kx][tx] Z[a][kx](x + tx - 1); the pair
        // of columns (x0 - 1, x0) / (x0, x0 + 1) / (x0 + 1, x0 + 2) for tx = 0 / 1 / 2, the same sharing between the
        // first and the last tap column
        f32x4 o[2];
#pragma unroll
        for (int a = 0; a < 2; a++) {
          f32x2 M[3];
#pragma unroll
          for (int kx = 0; kx < 3; kx++) M[kx] = (f32x2){Z[a][kx][0][1], Z[a][kx][1][0]};
          const f32x2 t1 = Z[a][0][0] + M[2], t2 = M[0] + Z[a][2][1];
#pragma unroll
          for (int b = 0; b < 2; b++) {
            f32x2 acc = hs_fma2(cy[b][0][0], t1, (f32x2){bias4, bias4});
            acc = hs_fma2(cy[b][0][1], t2, acc);
#pragma unroll
            for (int tx = 0; tx < 3; tx++)
              if (HS_NZ[b][1][tx]) acc = hs_fma2(cy[b][1][tx], tx == 0 ? Z[a][1][0] : tx == 1 ? M[1] : Z[a][1][1], acc);
            o[0][2 * a + b] = acc[0];
            o[1][2 * a + b] = acc[1];
          }
        }
        // frame pixels: phase (a, b) of pixel px is heat-map pixel (2 R + a, 2 x + b); the corrections are zero elsewhere
        if (side) { o[1][1] -= ccol[0]; o[1][3] -= ccol[1]; }  // block-uniform
        else { o[0][0] -= ccol[0]; o[0][2] -= ccol[1]; }
        if (edge) {
          if (st < 3) { o[0][0] -= crow[0]; o[0][1] -= crow[1]; o[1][0] -= crow[2]; o[1][1] -= crow[3]; }
          else { o[0][2] -= crow[0]; o[0][3] -= crow[1]; o[1][2] -= crow[2]; o[1][3] -= crow[3]; }
        }
        if (EXTRA && p.heat && ok) {
          float *hp = p.heat + (size_t)s * HD_PS * HD_PS + (size_t)(2 * R) * HD_PS + 2 * x0;
          *reinterpret_cast<f32x4 *>(hp) = (f32x4){o[0][0], o[0][1], o[1][0], o[1][1]};
          *reinterpret_cast<f32x4 *>(hp + HD_PS) = (f32x4){o[0][2], o[0][3], o[1][2], o[1][3]};
        }
        if (EXTRA && p.ptr_probe && ok) {
          const int qx = p.probe[2 * s] - 2 * x0, qy = p.probe[2 * s + 1] - 2 * R;
          if (qx >= 0 && qx < 4 && qy >= 0 && qy < 2) p.ptr_probe[s] = o[qx >> 1][2 * qy + (qx & 1)];
        }
        // arg-max: the maximum of the pass (med3 with +inf = max without the canonicalising pre-op), and where it beats
        // the lane's running maximum (strictly: the rows grow with the sub-step, so the first maximum in C order
        // stays) the 8 values are kept; the position inside the pass is looked up once, behind the loop
        const float m01 = hd_max_raw(hd_max_raw(o[0][0], o[0][1]), hd_max_raw(o[0][2], o[0][3]));
        const float m23 = hd_max_raw(hd_max_raw(o[1][0], o[1][1]), hd_max_raw(o[1][2], o[1][3]));
        const float m8 = hd_max_raw(m01, m23);
        const bool better = ok && m8 > tv;
        if (__builtin_amdgcn_ballot_w64(better) != 0) {   // wave-uniform: most passes improve no lane's maximum
          if (better) {
            tv = m8;
            tst = st;
            snap[0] = o[0];
            snap[1] = o[1];
          }
        }
      }
      HS_STAMP(2);
      R += 5;
      coff += 10;
#pragma unroll
      for (int dy = 0; dy < 3; dy++) {  // + 5 rows, wrapped at the ring size
        const unsigned n = q[dy] + 5 * PB;
        q[dy] = n >= qlim ? n - RING : n;
      }
      HS_STAMP(4);
      __syncthreads();
      HS_STAMP(5);
    }
    HS_STAMP_OUT();

    // ---- arg-max of the strip (first maximum in C order) ----
    int slot = 0;  // first of the 8 values, in flat order (row a, pixel, column b), that equals the maximum
#pragma unroll
    for (int i = 7; i >= 0; i--) {
      const int a = i >> 2, px = (i >> 1) & 1, b = i & 1;
      if (snap[px][2 * a + b] == tv) slot = i;
    }
    const int Rb = 5 * tst - 8 + r_in;
    const int y = 2 * Rb + (slot >> 2), x = 2 * x0 + (slot & 3);
    unsigned long long key = 0ull;
    if (tv > -INFINITY) key = ((unsigned long long)hd_ordered_f32(tv) << 32) | (unsigned long long)(~(unsigned)(y * HD_PS + x));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long other = __shfl_xor(key, o);
      key = other > key ? other : key;
    }
    if (lane == 0 && key) atomicMax(&p.best[s], key);
  }
}

int ofx_head_compact(ofx_handle *h, int S, const uint8_t *mask, int32_t *live) {
  hipLaunchKernelGGL(k_head_compact, dim3(1), dim3(1024), 0, h->stream, S, mask, live);
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}

size_t ofx_head_frame_bytes(size_t S, size_t *u2fr, size_t *vfr, size_t *c4) {
  *u2fr = 4 * S * 1600;
  *vfr = 4 * S * 7200;
  *c4 = 4 * S * 1600;
  return *u2fr + *vfr + *c4;
}

int ofx_launch_head(ofx_handle *h, const HeadParams2 &p0) {
  HeadParams2 p = p0;
#if OFX_HEAD_HOOKS
  { const char *e = getenv("OFX_HEAD_ABLATE"); p.ablate = e ? atoi(e) : 0; }  // diagnostics: 1 no stage C, 2 no stage A / B
  static unsigned long long *dbg = nullptr;
  static int dbg_calls = 0;
  const unsigned dbg_blocks = (unsigned)(((p.S + 7) / 8) * 16);
  p.dbg = nullptr;
  if (getenv("OFX_HEAD_STAMPS")) {  // s_memtime stamps per wave, printed for the 20th forward
    if (!dbg) (void)hipMalloc(&dbg, (size_t)dbg_blocks * 6 * (HS_NB + 4) * 8);
    if (++dbg_calls == 20) p.dbg = dbg;
  }
#endif
  if (p.mask) {
    if (!p.live) { ofx_set_error("ofx_launch_head: a mask needs the live-list scratch"); return OFX_ERR_STATE; }
    if (!p.live_ready) hipLaunchKernelGGL(k_head_compact, dim3(1), dim3(1024), 0, h->stream, p.S, p.mask, p.live);
  } else {
    p.live = nullptr;
  }
  if (p.frames_ref) hipLaunchKernelGGL(k_head_frames_ref, dim3((unsigned)p.S), dim3(HF_THREADS), 0, h->stream, p);
  else hipLaunchKernelGGL(k_head_frames, dim3((unsigned)((p.S + 7) & ~7)), dim3(HG_THREADS), 0, h->stream, p);
  OFX_HIP(hipGetLastError());
  const unsigned blocks = (unsigned)(((p.S + 7) / 8) * 16);
  int rc;
  if (p.event_base >= 0 && (rc = ofx_event_record(h, p.event_base))) return rc;
  const bool extra = p.heat || p.ptr_probe;
#define HS_LAUNCH(E, L) hipLaunchKernelGGL((k_head_stream<E, L>), dim3(blocks), dim3(HS_THREADS), 0, h->stream, p)
  if (p.bf16 == 1) { if (extra) HS_LAUNCH(true, 1); else HS_LAUNCH(false, 1); }
  else if (p.bf16 == 2) { if (extra) HS_LAUNCH(true, 2); else HS_LAUNCH(false, 2); }
  else { if (extra) HS_LAUNCH(true, 0); else HS_LAUNCH(false, 0); }
#undef HS_LAUNCH
  OFX_HIP(hipGetLastError());
  if (p.event_base >= 0 && (rc = ofx_event_record(h, p.event_base + 1))) return rc;
#if OFX_HEAD_HOOKS
  if (p.dbg) {
    (void)hipStreamSynchronize(h->stream);
    unsigned long long *hst = (unsigned long long *)malloc((size_t)blocks * 6 * (HS_NB + 4) * 8);
    (void)hipMemcpy(hst, p.dbg, (size_t)blocks * 6 * (HS_NB + 4) * 8, hipMemcpyDeviceToHost);
    double sum[2][6] = {{0}};
    for (unsigned b = 0; b < blocks; b++)
      for (int w = 0; w < HS_NB + 4; w++)
        for (int i = 0; i < 6; i++) sum[w >= HS_NB][i] += (double)hst[((size_t)b * (HS_NB + 4) + w) * 6 + i] / (w >= HS_NB ? 4.0 : (double)HS_NB);
    for (int r = 0; r < 2; r++) {
      fprintf(stderr, "head stamps %s (cycles per wave and sub-step):", r ? "C" : "B");
      for (int i = 0; i < 6; i++) fprintf(stderr, " %.0f", sum[r][i] / ((double)blocks * HS_NS));
      fprintf(stderr, "\n");
    }
    free(hst);
  }
#endif
  return OFX_OK;
}
