// ofx_fit.hip - the convolution side of model.fit (agents/qlearnIA_V2.py:284, graph :123-190) in its LEAN form:
// only the pre-activation tensor z of every convolution is kept in HBM.  What the plain form (ofx_train.hip) also
// stored - relu(bn(z)), the pooled / up-sampled inputs of the next convolution, x-hat, the gradient of every one of them -
// is recomputed where it is used, from z and the layer's batch statistics, inside the kernel that needs it:
//
//   f_conv_fwd   z = conv3x3(src) + b for a tile of 10 x 100 pixels, the input tile built in LDS by a SOURCE functor:
//                bits -> float | pool2(relu(bn(z_prev))) | up2(relu(bn(z_prev))) | up2(u0); a thread owns 4 pixels x
//                all output channels; the epilogue keeps the per-channel sums of z and z*z of the block (doubles, one
//                ordered combine per layer): BatchNorm's batch statistics without a pass of their own.
//   f_b1_pool /  g = d loss / d z's activation, masked by the ReLU: the transposed convolution of the NEXT layer's dz
//   f_b1_up      (tile in LDS) pushed back through the pooling (first maximum of the window, recomputed) or through the
//                x2 bilinear up-sampling (gather form), plus the block's sums of g and g * xhat for BatchNorm's backward.
//   f_bw         dz = gamma rs (g - mean(g) - xhat mean(g xhat)) written in place over g, and the weight gradient
//                dW[tap][ci][co] = sum in[ci][p + tap] dz[co][p] with the input tile rebuilt by the same source functor:
//                a wave owns an input channel, a lane a pixel, 9 x CO accumulators per lane, flushed into doubles.
//                (f_bw_small: the layers with CI x CO <= 8, a thread owns 4 pixels and all sums.)
//   f_first_* /  the first layer (1-bit inputs) is never materialised: batch statistics and the weight gradient (BatchNorm's
//   f_bits_corr  backward folded in analytically) from the autocorrelation of the shifted bit maps (popcounts), its pooled
//                activation through the forward's table kernel; the backward visits only the windows that see a set bit
//                (x-hat from a table), the empty ones enter through sums of the second layer's dz (f_first_bwd).
//   f_top_point_* the textbook targets (one error per sample): the output convolution, its gradients and the last head layer's g
//                from the ONE heat-map pixel that carries an error.
//   f_out_*      (dense targets) the output convolution (8 -> 1 at 400 x 400 behind the last up-sampling) in PHASE form: a 3 x 3
//                convolution of the 200 x 200 activation with 4 phase channels on f_conv_fwd / f_bw, plus two small
//                kernels that correct the plane's frame (zero padding against the phase form's repeated edge) exactly.
//
// Every reduction has a fixed order (tile -> block assignment by index, ordered combines): two fits from the same state
// give the same bits.  fp32 on the vector ALU (accumulators paired over the channel index: v_pk_fma_f32) with FMA
// contraction; checked against torch autograd in float64 and against the plain form (tests/test_train.py).  Per row of
// the minibatch 9.4 MB of workspace instead of 61 MB; 4096 rows in 42 ms (r03: 66).
#include "ofx_internal.h"
#include "ofx_fit.h"
#include "ofx_diag.h"

namespace {

__device__ __forceinline__ float bn_act(float z, float sc, float sh) { return fmaxf(fmaf(z, sc, sh), 0.f); }

// (a0, a1) += v * (w0, w1) as ONE v_pk_fma_f32 (the splat of v is an op_sel of the instruction): twice the FMAs per
// vector-issue slot of v_fmac_f32, the same roundings.  The inner loops below pair their accumulators over the channel
// index.
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void fma2(float &a0, float &a1, float v, float w0, float w1) {
  const f2v r = __builtin_elementwise_fma((f2v){v, v}, (f2v){w0, w1}, (f2v){a0, a1});
  a0 = r.x;
  a1 = r.y;
}
template <int N>
__device__ __forceinline__ void fma_row(float (&acc)[N], float v, const float (&w)[N]) {   // acc[i] += v * w[i]
  if constexpr (N % 2 == 0) {
#pragma unroll
    for (int i = 0; i < N; i += 2) fma2(acc[i], acc[i + 1], v, w[i], w[i + 1]);
  } else {
#pragma unroll
    for (int i = 0; i < N; i++) acc[i] = fmaf(v, w[i], acc[i]);
  }
}

// x2 bilinear taps of up-res index u over n source cells: half-pixel centres, or (legacy) src = dst / 2 - the same
// conventions as ofx_train.hip's up_taps / the forward's OFX_OPT_BILINEAR_LEGACY
__device__ __forceinline__ void fit_up_taps(int u, int n, int &i0, int &i1, float &w1, int legacy) {
  const int k = u >> 1;
  if (legacy) { i0 = k; i1 = min(k + 1, n - 1); w1 = (u & 1) ? 0.5f : 0.f; }
  else if (u & 1) { i0 = k; i1 = min(k + 1, n - 1); w1 = 0.25f; }
  else { i0 = max(k - 1, 0); i1 = k; w1 = 0.75f; }
}

// The weight of low-res cell i in the up-res cells 2 i - 2 + k, k = 0 .. 4, of a x2 bilinear up-sampling over n cells
// (0 where the up-res cell lies outside the plane): what fit_up_taps gives cell by cell, in closed form - the gather of
// f_b1_up spent ~150 instructions per thread and tile on ten calls of it.  k = 0 never has i among its taps.
__device__ __forceinline__ void fit_up_coef(int i, int n, int legacy, float (&c)[5]) {
  c[0] = 0.f;
  if (legacy) {
    c[1] = i > 0 ? 0.5f : 0.f; c[2] = 1.f; c[3] = i == n - 1 ? 1.f : 0.5f; c[4] = 0.f;
  } else {
    c[1] = i > 0 ? 0.25f : 0.f; c[2] = i == 0 ? 1.f : 0.75f; c[3] = i == n - 1 ? 1.f : 0.75f; c[4] = i < n - 1 ? 0.25f : 0.f;
  }
}

struct FitSrc {
  float *keep;        // POOL in f_conv_fwd: where the pooled activation [n][C][H][W] is written (the weight gradient reads it
                      // back through the PLANE source instead of pooling four times the bytes of z again); may be null
  const void *p;      // bits [n][C][5000] or the producing layer's z [n][C][h][w]
  const float *act;   // the producing layer's {scale, shift} per channel (relu(z * scale + shift)); unused for bits / raw;
                      // PLANE: 16 bytes of zeros in device memory
  int h, w;           // dims of p's planes
  int legacy;
};

// value of the convolution's input (channel c of sample s) at (y, x) of its H x W plane; zero outside
template <int SRC, int C>
__device__ __forceinline__ float src_value(const FitSrc &S, size_t s, int c, int y, int x, int H, int W) {
  if constexpr (SRC == OFX_FIT_SRC_ACTREP) {   // relu(bn(z)) at the plane's own resolution, edge cells repeated outwards
    const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), W - 1);
    return bn_act(reinterpret_cast<const float *>(S.p)[((s * C + c) * (size_t)H + yc) * W + xc], S.act[2 * c], S.act[2 * c + 1]);
  }
  if (y < 0 || y >= H || x < 0 || x >= W) return 0.f;
  if constexpr (SRC == OFX_FIT_SRC_PLANE) return reinterpret_cast<const float *>(S.p)[((s * C + c) * (size_t)H + y) * W + x];
  if constexpr (SRC == OFX_FIT_SRC_BITS) {
    const uint32_t *b = reinterpret_cast<const uint32_t *>(S.p) + (s * C + c) * (size_t)((H * W) >> 5);
    const int p = y * W + x;
    return (float)((b[p >> 5] >> (p & 31)) & 1u);
  } else if constexpr (SRC == OFX_FIT_SRC_POOL) {
    const float *z = reinterpret_cast<const float *>(S.p) + ((s * C + c) * (size_t)S.h + 2 * y) * S.w + 2 * x;
    const float sc = S.act[2 * c], sh = S.act[2 * c + 1];
    const float2 r0 = *reinterpret_cast<const float2 *>(z), r1 = *reinterpret_cast<const float2 *>(z + S.w);
    return fmaxf(fmaxf(bn_act(r0.x, sc, sh), bn_act(r0.y, sc, sh)), fmaxf(bn_act(r1.x, sc, sh), bn_act(r1.y, sc, sh)));
  } else {
    int y0, y1, x0, x1;
    float wy, wx;
    fit_up_taps(y, S.h, y0, y1, wy, S.legacy);
    fit_up_taps(x, S.w, x0, x1, wx, S.legacy);
    const float *q = reinterpret_cast<const float *>(S.p) + (s * C + c) * (size_t)S.h * S.w;
    float v00 = q[y0 * S.w + x0], v01 = q[y0 * S.w + x1], v10 = q[y1 * S.w + x0], v11 = q[y1 * S.w + x1];
    if constexpr (SRC == OFX_FIT_SRC_UP) {
      const float sc = S.act[2 * c], sh = S.act[2 * c + 1];
      v00 = bn_act(v00, sc, sh); v01 = bn_act(v01, sc, sh); v10 = bn_act(v10, sc, sh); v11 = bn_act(v11, sc, sh);
    }
    const float top = v00 * (1.f - wx) + v01 * wx, bot = v10 * (1.f - wx) + v11 * wx;
    return top * (1.f - wy) + bot * wy;
  }
}

// A tile of NROWS rows x PITCH floats in LDS by LDS-direct loads, a ROW per wave and step: lane l of a wave instruction writes
// dword l behind the instruction's base, so a row takes ceil(PITCH / 64) instructions.  row_ptr(r) (wave-uniform: scalar
// registers) is the global address of the plane row under tile row r, or null outside the plane; tile column c <-> plane
// column xlo + c; columns >= NCOL (the pad) and cells outside the plane read *zero.  Every load of the tile is in flight at
// once, no register holds one; per load a lane spends two compares, one select and one address add - enumerating the tile
// flat (element -> row, column by division, 64-bit address per lane) cost ~30 vector instructions per load, 40 % of what the
// conv2 forward issued (SQ_INSTS_VALU, profiles/r04_fit_pmc_sq.txt).  The caller waits (fill_landed) in front of its barrier.
template <int NROWS, int PITCH, int NCOL, int NWAVES, class RowPtr>
__device__ __forceinline__ void lds_direct_rows(float *tile, int wvu, int lane, int xlo, int W, const float *zero, RowPtr row_ptr) {
#pragma unroll 1
  for (int r = wvu; r < NROWS; r += NWAVES) {
    const float *rp = row_ptr(r);
#pragma unroll
    for (int h = 0; h < (PITCH + 63) / 64; h++) {
      const int c = 64 * h + lane, x = xlo + c;
      if (c < PITCH) {
        const float *src = (rp != nullptr && c < NCOL && x >= 0 && x < W) ? rp + x : zero;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(tile + r * PITCH + 64 * h), 4, 0, 0);
      }
    }
  }
}

// Fill loops run in batches of FILL_U elements per thread, all global loads of a batch requested before the first value
// is used: written as a plain `load, compute, store to LDS` loop the compiler waits for every load in turn, and with two
// or three workgroups per CU nothing hides ~2 us of HBM latency per iteration (the first lean build spent 38 000 cycles
// on a tile that holds 4 000 cycles of instructions).
constexpr int FILL_U = 8;

// The input tile of a convolution: in[CI][TR + 2][LP] <- source values of rows y0 - 1 .. y0 + TR, columns x0 - 1 ..
// x0 + TW (zero outside the plane).  Up-sampling sources first stage the low-res activation they interpolate
// (lo[CI][TR / 2 + 3][TW / 2 + 3], coordinates clamped like the taps): every low-res value is read and passed through
// BatchNorm + ReLU once instead of up to 16 times.  Barriers inside: call from all NT threads.
template <int SRC> constexpr bool src_is_up = SRC == OFX_FIT_SRC_UP || SRC == OFX_FIT_SRC_UPRAW;
template <int SRC, int CI, int TR, int TW> constexpr int lo_floats = src_is_up<SRC> ? CI * (TR / 2 + 3) * (TW / 2 + 3) : 1;
// STAGE = false: every tile value straight from global memory (src_value), in a plain loop - what measured faster in
// f_bw, whose 512-thread workgroups lose their second workgroup per CU to the registers of the batched form.
template <int SRC, int CI, int TR, int TW, int LP, int NT, int FU = 8, bool STAGE = true>
__device__ __forceinline__ void fill_input(float (*in)[TR + 2][LP], float *lo, const FitSrc &S, size_t s, int y0, int x0,
                                           int H, int W, int tid) {
  if constexpr (SRC == OFX_FIT_SRC_PLANE) {
    // a stored plane as it is (zero outside) by LDS-direct loads (lds_direct_rows).  (Through registers, 16 bytes per load,
    // the in-order vmcnt left one or two loads per thread in flight in front of their LDS stores: ~24 KB per CU where HBM's
    // latency wants 64.)  Cells outside the plane and the row's pad read a zero word (S.act: 16 bytes of zeros for this source).
    static_assert(NT % 64 == 0, "whole waves");
    const float *zp = reinterpret_cast<const float *>(S.p);
    lds_direct_rows<CI * (TR + 2), LP, TW + 2, NT / 64>(&in[0][0][0], __builtin_amdgcn_readfirstlane(tid >> 6), tid & 63, x0 - 1, W,
                                                         S.act, [&](int r) -> const float * {
      const int c = r / (TR + 2), y = y0 - 1 + (r - c * (TR + 2));
      return (y >= 0 && y < H) ? zp + ((s * CI + c) * (size_t)H + y) * W : nullptr;
    });
    // the caller waits (fill_landed) in front of the barrier that publishes the tile
  } else if constexpr (SRC == OFX_FIT_SRC_ACTREP) {
    // rows of the plane itself: one element at either end of a tile row, TW / 4 aligned 16-byte loads between them
    // (element-wise this fill was 79 % of the phase-form output convolution: tools/fit_ablate.sh)
    static_assert(TW % 4 == 0, "tile width");
    constexpr int ST = TW / 4 + 2;
    const float *zp = reinterpret_cast<const float *>(S.p);
    for (int e = tid; e < CI * (TR + 2) * ST; e += NT) {
      const int c = e / ((TR + 2) * ST), rem = e - c * ((TR + 2) * ST), yy = rem / ST, st = rem - yy * ST;
      const int yc = min(max(y0 - 1 + yy, 0), H - 1);
      const float sc = S.act[2 * c], sh = S.act[2 * c + 1];
      const float *zr = zp + ((s * CI + c) * (size_t)H + yc) * W;
      if (st == 0) in[c][yy][0] = bn_act(zr[max(x0 - 1, 0)], sc, sh);
      else if (st == ST - 1) in[c][yy][TW + 1] = bn_act(zr[min(x0 + TW, W - 1)], sc, sh);
      else {
        const float4 v = *reinterpret_cast<const float4 *>(zr + x0 + 4 * (st - 1));
        float *o = &in[c][yy][1 + 4 * (st - 1)];
        o[0] = bn_act(v.x, sc, sh); o[1] = bn_act(v.y, sc, sh); o[2] = bn_act(v.z, sc, sh); o[3] = bn_act(v.w, sc, sh);
      }
    }
  } else if constexpr (SRC == OFX_FIT_SRC_POOL) {
    // two pooled pixels per step from two 16-byte loads (tile column 0 is an odd plane column: one single step at either
    // end of a row, TW / 2 pairs between them)
    static_assert(TW % 2 == 0, "tile width");
    constexpr int ST = TW / 2 + 2;
    const float *zp = reinterpret_cast<const float *>(S.p);
    for (int e = tid; e < CI * (TR + 2) * ST; e += NT) {
      const int c = e / ((TR + 2) * ST), rem = e - c * ((TR + 2) * ST), yy = rem / ST, st = rem - yy * ST;
      const int y = y0 - 1 + yy, xx = st == 0 ? 0 : 2 * st - 1, x = x0 - 1 + xx;
      const bool pair = st != 0 && st != ST - 1;
      float v0 = 0.f, v1 = 0.f;
      if (y >= 0 && y < H) {
        const float sc = S.act[2 * c], sh = S.act[2 * c + 1];
        const float *zr = zp + ((s * CI + c) * (size_t)S.h + 2 * y) * S.w + 2 * x;
        if (pair) {
          const float4 r0 = *reinterpret_cast<const float4 *>(zr), r1 = *reinterpret_cast<const float4 *>(zr + S.w);
          v0 = fmaxf(fmaxf(bn_act(r0.x, sc, sh), bn_act(r0.y, sc, sh)), fmaxf(bn_act(r1.x, sc, sh), bn_act(r1.y, sc, sh)));
          v1 = fmaxf(fmaxf(bn_act(r0.z, sc, sh), bn_act(r0.w, sc, sh)), fmaxf(bn_act(r1.z, sc, sh), bn_act(r1.w, sc, sh)));
        } else if (x >= 0 && x < W) {
          const float2 r0 = *reinterpret_cast<const float2 *>(zr), r1 = *reinterpret_cast<const float2 *>(zr + S.w);
          v0 = fmaxf(fmaxf(bn_act(r0.x, sc, sh), bn_act(r0.y, sc, sh)), fmaxf(bn_act(r1.x, sc, sh), bn_act(r1.y, sc, sh)));
        }
      }
      in[c][yy][xx] = v0;
      if (pair) in[c][yy][xx + 1] = v1;
    }
  } else if constexpr (!STAGE) {
    for (int e = tid; e < CI * (TR + 2) * (TW + 2); e += NT) {
      const int c = e / ((TR + 2) * (TW + 2)), rem = e - c * ((TR + 2) * (TW + 2));
      const int yy = rem / (TW + 2), xx = rem - yy * (TW + 2);
      in[c][yy][xx] = src_value<SRC, CI>(S, s, c, y0 - 1 + yy, x0 - 1 + xx, H, W);
    }
  } else if constexpr (src_is_up<SRC>) {
    static_assert(TR % 2 == 0 && TW % 2 == 0, "tile origin must be even");
    constexpr int LR = TR / 2 + 3, LC = TW / 2 + 3;
    const int ly0 = y0 / 2 - 1, lx0 = x0 / 2 - 1;
    for (int e0 = tid; e0 < CI * LR * LC; e0 += NT * FU) {
      float v[FU];
#pragma unroll
      for (int u = 0; u < FU; u++) {
        const int e = min(e0 + u * NT, CI * LR * LC - 1);
        const int c = e / (LR * LC), rem = e - c * (LR * LC), i = rem / LC, j = rem - i * LC;
        const int yy = min(max(ly0 + i, 0), S.h - 1), xx = min(max(lx0 + j, 0), S.w - 1);
        v[u] = reinterpret_cast<const float *>(S.p)[((s * CI + c) * (size_t)S.h + yy) * S.w + xx];
      }
#pragma unroll
      for (int u = 0; u < FU; u++) {
        const int e = e0 + u * NT;
        if (e < CI * LR * LC) {
          if constexpr (SRC == OFX_FIT_SRC_UP) { const int c = e / (LR * LC); v[u] = bn_act(v[u], S.act[2 * c], S.act[2 * c + 1]); }
          lo[e] = v[u];
        }
      }
    }
    __syncthreads();
    // one thread per low-res cell (k, m): its 2 x 2 up-res block from the cell's 3 x 3 neighbourhood.  Along an axis
    // up(2 k + a) = wm[a] lo[max(k - 1, 0)] + w0[a] lo[k] + wp[a] lo[min(k + 1, n - 1)] - fit_up_taps written per cell:
    // half-pixel (.25, .75, 0) / (0, .75, .25), legacy (0, 1, 0) / (0, .5, .5).
    constexpr int CR = TR / 2 + 2, CC = TW / 2 + 2;
    const float wm[2] = {S.legacy ? 0.f : 0.25f, 0.f}, w0[2] = {S.legacy ? 1.f : 0.75f, S.legacy ? 0.5f : 0.75f},
                wp[2] = {0.f, S.legacy ? 0.5f : 0.25f};
    for (int e = tid; e < CI * CR * CC; e += NT) {
      const int c = e / (CR * CC), rem = e - c * (CR * CC), i = rem / CC, j = rem - i * CC;
      const int k = ly0 + i, m = lx0 + j;                       // the cell; lo row index of low-res row r is r - ly0
      const int ri[3] = {min(max(k - 1, 0), S.h - 1) - ly0, min(max(k, 0), S.h - 1) - ly0, min(max(k + 1, 0), S.h - 1) - ly0};
      const int cj[3] = {min(max(m - 1, 0), S.w - 1) - lx0, min(max(m, 0), S.w - 1) - lx0, min(max(m + 1, 0), S.w - 1) - lx0};
      const float *q = lo + c * (LR * LC);
      float hb[3][2];
#pragma unroll
      for (int r = 0; r < 3; r++) {
        const float vm = q[max(ri[r], 0) * LC + max(cj[0], 0)], v0 = q[max(ri[r], 0) * LC + cj[1]], vp = q[max(ri[r], 0) * LC + cj[2]];
#pragma unroll
        for (int b = 0; b < 2; b++) hb[r][b] = wm[b] * vm + w0[b] * v0 + wp[b] * vp;
      }
#pragma unroll
      for (int a = 0; a < 2; a++) {
        const int y = 2 * k + a, yy = y - (y0 - 1);
        if (yy < 0 || yy >= TR + 2) continue;
#pragma unroll
        for (int b = 0; b < 2; b++) {
          const int x = 2 * m + b, xx = x - (x0 - 1);
          if (xx < 0 || xx >= TW + 2) continue;
          const float v = wm[a] * hb[0][b] + w0[a] * hb[1][b] + wp[a] * hb[2][b];
          in[c][yy][xx] = (y >= 0 && y < H && x >= 0 && x < W) ? v : 0.f;
        }
      }
    }
  } else {
    constexpr int NE = CI * (TR + 2) * (TW + 2);
    for (int e0 = tid; e0 < NE; e0 += NT * FU) {
      float v[FU];
#pragma unroll
      for (int u = 0; u < FU; u++) {
        const int e = e0 + u * NT;
        const int c = e / ((TR + 2) * (TW + 2)), rem = e - c * ((TR + 2) * (TW + 2));
        const int yy = rem / (TW + 2), xx = rem - yy * (TW + 2);
        v[u] = e < NE ? src_value<SRC, CI>(S, s, c, y0 - 1 + yy, x0 - 1 + xx, H, W) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < FU; u++) {
        const int e = e0 + u * NT;
        if (e < NE) (&in[0][0][0])[(e / (TW + 2)) * LP + e % (TW + 2)] = v[u];
      }
    }
  }
}

// in front of the barrier behind fill_input: LDS-direct loads are published by vmcnt(0), which the memory model does not
// promise at a workgroup fence (k_conv3_stream)
template <int SRC>
__device__ __forceinline__ void fill_landed() {
  if constexpr (SRC == OFX_FIT_SRC_PLANE) __builtin_amdgcn_s_waitcnt(0x0F70);
}

// sum of v over the 64 lanes (every lane gets it)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// K consecutive floats of an LDS array at a wave-uniform offset (a broadcast read).  The kernels keep their weights in
// LDS: read through the kernel argument they are loop-invariant scalar loads, the compiler hoists all 576 of an 8 -> 8
// layer out of the tile loop and parks them in vector registers (351 VGPRs, one wave per SIMD).
template <int K>
__device__ __forceinline__ void lds_vec(const float *p, float (&o)[K]) {
  if constexpr (K % 4 == 0) {
#pragma unroll
    for (int i = 0; i < K / 4; i++) {
      const float4 v = reinterpret_cast<const float4 *>(p)[i];
      o[4 * i] = v.x; o[4 * i + 1] = v.y; o[4 * i + 2] = v.z; o[4 * i + 3] = v.w;
    }
  } else if constexpr (K == 2) {
    const float2 v = *reinterpret_cast<const float2 *>(p);
    o[0] = v.x; o[1] = v.y;
  } else {
#pragma unroll
    for (int i = 0; i < K; i++) o[i] = p[i];
  }
}

// ---------------------------------------------------------------- forward
constexpr int F_TR = 10;  // rows of a forward tile

// PHASE (the output convolution in phase form, see f_out_prep): the CO = 4 outputs of a cell (y, x) are the 2 x 2 block
// (2 y + a, 2 x + b), channel 2 a + b, of ONE plane of 2 H x 2 W.
template <int CI, int CO, int SRC, int TW, bool STATS, bool PHASE = false>
__global__ __launch_bounds__(256, 3) void f_conv_fwd(int n, int H, int W, FitSrc S, const float *__restrict__ w,
                                                  const float *__restrict__ b, float *__restrict__ z,
                                                  double *__restrict__ part) {
  constexpr int TPR = (TW + 3) / 4, LP = 4 * TPR + 4;
  static_assert(F_TR * TPR <= 256, "tile does not fit the block");
  __shared__ __align__(16) float in[CI][F_TR + 2][LP];
  __shared__ float lo[lo_floats<SRC, CI, F_TR, TW>];
  __shared__ double red[4][2 * CO];
  const int tid = threadIdx.x, r = tid / TPR, q = tid - r * TPR;
  const bool active = r < F_TR;
  const int tx_n = W / TW, ty_n = H / F_TR, per_s = tx_n * ty_n;
  const long ntiles = (long)n * per_s;
  double d1[CO], d2[CO];
#pragma unroll
  for (int co = 0; co < CO; co++) { d1[co] = 0.0; d2[co] = 0.0; }
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const size_t s = tile / per_s;
    const int t = (int)(tile - (long)s * per_s), y0 = (t / tx_n) * F_TR, x0 = (t % tx_n) * TW;
    __syncthreads();
    if (OFX_FIT_ABLATE != 1) fill_input<SRC, CI, F_TR, TW, LP, 256>(in, lo, S, s, y0, x0, H, W, tid);
    fill_landed<SRC>();
    __syncthreads();
    if constexpr (SRC == OFX_FIT_SRC_POOL) {
      if (S.keep) {   // block-uniform: the tile's own pixels of the pooled activation, once
        constexpr int V = TW % 4 == 0 ? 4 : 2, TWV = TW / V;
        for (int e = tid; e < CI * F_TR * TWV; e += 256) {
          const int c = e / (F_TR * TWV), rem = e - c * (F_TR * TWV), yy = rem / TWV, xv = rem - yy * TWV;
          const float *q = &in[c][yy + 1][1 + V * xv];
          float *o = S.keep + ((s * CI + c) * (size_t)H + y0 + yy) * W + x0 + V * xv;
          if constexpr (V == 4) *reinterpret_cast<float4 *>(o) = make_float4(q[0], q[1], q[2], q[3]);
          else *reinterpret_cast<float2 *>(o) = make_float2(q[0], q[1]);
        }
      }
    }
    if (!active || OFX_FIT_ABLATE == 2) continue;
    // weights through the scalar cache (s_load, an SGPR pair per v_pk_fma_f32): 18 broadcast ds_read_b128 per input channel
    // kept the LDS return path busier than the vector ALU.  The offset the compiler cannot see through keeps the loads
    // inside the tile loop (hoisted, all 9 CI CO of them would sit in vector registers).
    int zoff;
    asm volatile("s_mov_b32 %0, 0" : "=s"(zoff));
    const float *__restrict__ wt = w + zoff;
    float acc[4][CO];
#pragma unroll
    for (int co = 0; co < CO; co++) {
      const float bv = b[co];
#pragma unroll
      for (int px = 0; px < 4; px++) acc[px][co] = bv;
    }
    // one input channel at a time: unrolled over ci the scheduler requests all 9 CI CO weights up front (512 VGPRs)
#pragma unroll 1
    for (int ci = 0; ci < CI; ci++) {
#pragma unroll
      for (int ky = 0; ky < 3; ky++) {
        const float4 a4 = *reinterpret_cast<const float4 *>(&in[ci][r + ky][4 * q]);
        const float2 a2 = *reinterpret_cast<const float2 *>(&in[ci][r + ky][4 * q + 4]);
        const float v[6] = {a4.x, a4.y, a4.z, a4.w, a2.x, a2.y};
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
          float wv[CO];
#pragma unroll
          for (int co = 0; co < CO; co++) wv[co] = wt[((ky * 3 + kx) * CI + ci) * CO + co];
#pragma unroll
          for (int px = 0; px < 4; px++) fma_row<CO>(acc[px], v[px + kx], wv);
        }
      }
    }
    const int y = y0 + r, x = x0 + 4 * q;
    if constexpr (PHASE) {
      static_assert(CO == 4 && TW % 4 == 0 && !STATS, "phase form");
#pragma unroll
      for (int a = 0; a < 2; a++) {
        float *op = z + (s * 2 * H + 2 * y + a) * (size_t)(2 * W) + 2 * x;
        *reinterpret_cast<float4 *>(op) = make_float4(acc[0][2 * a], acc[0][2 * a + 1], acc[1][2 * a], acc[1][2 * a + 1]);
        *reinterpret_cast<float4 *>(op + 4) = make_float4(acc[2][2 * a], acc[2][2 * a + 1], acc[3][2 * a], acc[3][2 * a + 1]);
      }
      continue;
    }
    float *const zs = z + s * (size_t)(CO * H) * W;   // the sample's planes (block-uniform); 32-bit offsets behind it
    const int pix = y * W + x;
#pragma unroll
    for (int co = 0; co < CO; co++) {
      float *zp = zs + co * H * W + pix;
      float f1 = 0.f, f2 = 0.f;
      if (4 * q + 4 <= TW) {
        *reinterpret_cast<float4 *>(zp) = make_float4(acc[0][co], acc[1][co], acc[2][co], acc[3][co]);
#pragma unroll
        for (int px = 0; px < 4; px++) { f1 += acc[px][co]; f2 = fmaf(acc[px][co], acc[px][co], f2); }
      } else {
#pragma unroll
        for (int px = 0; px < 4; px++)
          if (4 * q + px < TW) { zp[px] = acc[px][co]; f1 += acc[px][co]; f2 = fmaf(acc[px][co], acc[px][co], f2); }
      }
      if (STATS) { d1[co] += (double)f1; d2[co] += (double)f2; }
    }
  }
  if constexpr (STATS) {
#pragma unroll
    for (int co = 0; co < CO; co++) {
      const double a = wave_sum(d1[co]), c = wave_sum(d2[co]);
      if ((tid & 63) == 0) { red[tid >> 6][2 * co] = a; red[tid >> 6][2 * co + 1] = c; }
    }
    __syncthreads();
    if (tid < 2 * CO) part[(size_t)blockIdx.x * 2 * CO + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
  }
}

// part[block][2 C] -> sums[2 C] in block order; with gamma: also the batch statistics {mean, biased variance} and the
// activation's {scale, shift}
// (two levels, both in a fixed order: 16 threads per value take the rows i = j, j + 16, ..., then the 16 shares are added
// in j order - one thread walking 2048 rows took 0.5 ms per layer and pass)
__global__ __launch_bounds__(256) void f_finish(int nblocks, int c_n, double count, const double *part, const float *gamma,
                                                const float *beta, double *sums, float *stat, float *act) {
  __shared__ double share[16][17], sh[16];
  const int k = threadIdx.x & 15, j = threadIdx.x >> 4, nv = 2 * c_n;
  double acc = 0.0;
  if (k < nv)
    for (int i = j; i < nblocks; i += 16) acc += part[(size_t)i * nv + k];
  share[k][j] = acc;
  __syncthreads();
  if (threadIdx.x < 16) {
    double t = 0.0;
    for (int q = 0; q < 16; q++) t += share[threadIdx.x][q];
    sh[threadIdx.x] = t;
    if (sums && threadIdx.x < nv) sums[threadIdx.x] = t;
  }
  __syncthreads();
  if (!stat) return;
  if (threadIdx.x < c_n) {
    const int k = threadIdx.x;
    const double m = sh[2 * k] / count, v = sh[2 * k + 1] / count - m * m;
    const float mean = (float)m, var = (float)(v > 0.0 ? v : 0.0);
    stat[2 * k] = mean;
    stat[2 * k + 1] = var;
    stat[16 + k] = (float)(m - (double)mean);   // what the rounding of the mean dropped (f_bw: BatchNorm's backward)
    const float sc = gamma[k] * rsqrtf(var + 1e-3f);
    act[2 * k] = sc;
    act[2 * k + 1] = beta[k] - mean * sc;
  }
}

// p[n][8][h/2][w/2] = pool2(relu(bn(z))) - the trunk's last layer feeds the dense head through the flatten
__global__ void f_pool_act(int n, int C, int H, int W, const float *z, const float *act, float *p) {
  const int H2 = H / 2, W2 = W / 2;
  const size_t total = (size_t)n * C * H2 * W2;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int x = e % W2, y = (e / W2) % H2;
    const size_t pl = e / ((size_t)W2 * H2);
    const int c = pl % C;
    const float sc = act[2 * c], sh = act[2 * c + 1];
    const float *q = z + (pl * H + 2 * y) * W + 2 * x;
    p[e] = fmaxf(fmaxf(bn_act(q[0], sc, sh), bn_act(q[1], sc, sh)), fmaxf(bn_act(q[W], sc, sh), bn_act(q[W + 1], sc, sh)));
  }
}

// ---------------------------------------------------------------- backward 1: through the pooling
// Layer L of the trunk (8 channels, H x W): dp = conv3x3^T(dz of layer L + 1) on the pooled H/2 x W/2 grid (CONV), or
// dp given (the last trunk layer: the dense head's gradient); g = dp at the first maximum of each 2 x 2 window of
// a = relu(bn(z)) where that maximum is positive, zero elsewhere; part[block] = {sum g, sum g xhat} per channel.
constexpr int P_TR = 10, P_TW = 50;
template <bool CONV>
__global__ __launch_bounds__(256, 3) void f_b1_pool(int n, int H, int W, const float *__restrict__ dzn,
                                                 const float *__restrict__ wn, const float *__restrict__ z,
                                                 const float *__restrict__ stat, const float *__restrict__ act,
                                                 float *__restrict__ g, double *__restrict__ part,
                                                 const float *__restrict__ zero) {
  constexpr int C = 8, LP = P_TW + 4;
  __shared__ __align__(16) float dzt[CONV ? C : 1][P_TR + 2][LP];
  __shared__ double red[4][2 * C];
  const int Hp = H / 2, Wp = W / 2;
  const int tid = threadIdx.x, r = tid / (P_TW / 2), q = tid - r * (P_TW / 2);
  const int tx_n = (Wp + P_TW - 1) / P_TW, ty_n = (Hp + P_TR - 1) / P_TR, per_s = tx_n * ty_n;
  const long ntiles = (long)n * per_s;
  double s1[C], s2[C];
#pragma unroll
  for (int c = 0; c < C; c++) { s1[c] = 0.0; s2[c] = 0.0; }
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const size_t s = tile / per_s;
    const int t = (int)(tile - (long)s * per_s), y0 = (t / tx_n) * P_TR, x0 = (t % tx_n) * P_TW;
    if constexpr (CONV) {
      __syncthreads();
      lds_direct_rows<C * (P_TR + 2), LP, P_TW + 2, 4>(&dzt[0][0][0], __builtin_amdgcn_readfirstlane(tid >> 6), tid & 63, x0 - 1, Wp,
                                                        zero, [&](int rr) -> const float * {
        const int co = rr / (P_TR + 2), y = y0 - 1 + (rr - co * (P_TR + 2));
        return (y >= 0 && y < Hp) ? dzn + ((s * C + co) * (size_t)Hp + y) * Wp : nullptr;
      });
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): landed before the barrier publishes the tile
      __syncthreads();
    }
    const int yp = y0 + r, xp = x0 + 2 * q;
    if (r >= P_TR || yp >= Hp || xp >= Wp) continue;
    float dp[2][C];
    if constexpr (CONV) {
      // the kernel as [tap][co][ci] (wn here: the launcher's transposed copy) through the SCALAR cache, an SGPR pair per
      // v_pk_fma_f32: as 2 broadcast ds_read_b128 per tap and co the weights took 2.3x the LDS time the tile's FMAs take on
      // the vector ALU (r04).  The offset the compiler cannot see through keeps the loads inside the tile loop.
      int zoff;
      asm volatile("s_mov_b32 %0, 0" : "=s"(zoff));
      const float *__restrict__ wt = wn + zoff;
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int ci = 0; ci < C; ci++) dp[j][ci] = 0.f;
#pragma unroll 1
      for (int co = 0; co < C; co++) {
#pragma unroll
        for (int ky = 0; ky < 3; ky++) {
          // dz row yp - (ky - 1) = tile row r - ky + 2; columns xp - 1 .. xp + 2 = tile columns 2 q .. 2 q + 3
          const float2 a = *reinterpret_cast<const float2 *>(&dzt[co][r - ky + 2][2 * q]);
          const float2 c2 = *reinterpret_cast<const float2 *>(&dzt[co][r - ky + 2][2 * q + 2]);
          const float v[4] = {a.x, a.y, c2.x, c2.y};
#pragma unroll
          for (int kx = 0; kx < 3; kx++) {
            float wv[C];
#pragma unroll
            for (int ci = 0; ci < C; ci++) wv[ci] = wt[((ky * 3 + kx) * C + co) * C + ci];
            fma_row<C>(dp[0], v[2 - kx], wv);
            fma_row<C>(dp[1], v[3 - kx], wv);
          }
        }
      }
    }
    const bool both = xp + 1 < Wp;   // the thread's second pooled pixel exists (always, but for the odd-width last layer)
    const float *const zs = z + s * (size_t)(C * H) * W;   // the sample's planes (block-uniform); 32-bit offsets behind them
    float *const gs = g + s * (size_t)(C * H) * W;
#pragma unroll
    for (int ci = 0; ci < C; ci++) {
      const float mean = stat[2 * ci], rs = rsqrtf(stat[2 * ci + 1] + 1e-3f), sc = act[2 * ci], sh = act[2 * ci + 1];
      const int base = (ci * H + 2 * yp) * W + 2 * xp;   // behind the sample's planes zs / gs
      float zr[2][4];   // rows 2 yp, 2 yp + 1; columns 2 xp .. 2 xp + 3
      const bool vec = both && (W & 3) == 0;   // 16-byte accesses: the rows of the 50 x 50 layer are only 8-byte aligned
      if (vec) {
        const float4 r0 = *reinterpret_cast<const float4 *>(zs + base), r1 = *reinterpret_cast<const float4 *>(zs + base + W);
        zr[0][0] = r0.x; zr[0][1] = r0.y; zr[0][2] = r0.z; zr[0][3] = r0.w;
        zr[1][0] = r1.x; zr[1][1] = r1.y; zr[1][2] = r1.z; zr[1][3] = r1.w;
      } else {
        const float2 r0 = *reinterpret_cast<const float2 *>(zs + base), r1 = *reinterpret_cast<const float2 *>(zs + base + W);
        zr[0][0] = r0.x; zr[0][1] = r0.y; zr[1][0] = r1.x; zr[1][1] = r1.y;
        zr[0][2] = zr[0][3] = zr[1][2] = zr[1][3] = 0.f;
        if (both) {
          const float2 q0 = *reinterpret_cast<const float2 *>(zs + base + 2), q1 = *reinterpret_cast<const float2 *>(zs + base + W + 2);
          zr[0][2] = q0.x; zr[0][3] = q0.y; zr[1][2] = q1.x; zr[1][3] = q1.y;
        }
      }
      float o[2][4];
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const float zv[4] = {zr[0][2 * j], zr[0][2 * j + 1], zr[1][2 * j], zr[1][2 * j + 1]};
        float av[4];
#pragma unroll
        for (int i = 0; i < 4; i++) av[i] = bn_act(zv[i], sc, sh);
        int k = 0;
#pragma unroll
        for (int i = 1; i < 4; i++) if (av[i] > av[k]) k = i;
        float d = 0.f;
        if (j == 0 || both) d = CONV ? dp[j][ci] : dzn[((s * C + ci) * (size_t)Hp + yp) * Wp + xp + j];
        const float gv = (av[k] > 0.f && (j == 0 || both)) ? d : 0.f;
        o[0][2 * j] = k == 0 ? gv : 0.f; o[0][2 * j + 1] = k == 1 ? gv : 0.f;
        o[1][2 * j] = k == 2 ? gv : 0.f; o[1][2 * j + 1] = k == 3 ? gv : 0.f;
        s1[ci] += (double)gv;
        s2[ci] += (double)(gv * ((zv[k] - mean) * rs));
      }
      if (vec) {
        *reinterpret_cast<float4 *>(gs + base) = make_float4(o[0][0], o[0][1], o[0][2], o[0][3]);
        *reinterpret_cast<float4 *>(gs + base + W) = make_float4(o[1][0], o[1][1], o[1][2], o[1][3]);
      } else {
        *reinterpret_cast<float2 *>(gs + base) = make_float2(o[0][0], o[0][1]);
        *reinterpret_cast<float2 *>(gs + base + W) = make_float2(o[1][0], o[1][1]);
        if (both) {
          *reinterpret_cast<float2 *>(gs + base + 2) = make_float2(o[0][2], o[0][3]);
          *reinterpret_cast<float2 *>(gs + base + W + 2) = make_float2(o[1][2], o[1][3]);
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < C; c++) {
    const double a = wave_sum(s1[c]), b2 = wave_sum(s2[c]);
    if ((tid & 63) == 0) { red[tid >> 6][2 * c] = a; red[tid >> 6][2 * c + 1] = b2; }
  }
  __syncthreads();
  if (tid < 2 * C) part[(size_t)blockIdx.x * 2 * C + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// ---------------------------------------------------------------- backward 1: through the x2 up-sampling
// Producer P (C channels, h x w: u0 or a head layer's z) feeds conv3x3(up2(act(P))) with CON output channels at
// 2h x 2w.  dU = conv3x3^T(dz of that convolution) on the up-res grid (LDS), gathered back through the bilinear taps
// (fixed order), masked by P's ReLU; with BN the block's {sum g, sum g xhat}.
constexpr int U_TW = 50;
template <int C, int CON, int TRL, bool BN>
__global__ __launch_bounds__(256, 2) void f_b1_up(int n, int h, int w, const float *__restrict__ dzn,
                                               const float *__restrict__ wn, const float *__restrict__ zp,
                                               const float *__restrict__ stat, const float *__restrict__ act, int legacy,
                                               float *__restrict__ g, double *__restrict__ part, const float *__restrict__ zero) {
  constexpr int UR = 2 * TRL + 3, UC = 2 * U_TW + 3, UP_ = UC + 1;   // dU tile: up-res rows 2 y0 - 2 .., pitch
  constexpr int DR = UR + 2, DC = UC + 2, DP = DC + 1;               // dz tile: one more cell all round
  __shared__ float dzt[CON][DR][DP];
  __shared__ float du[C][UR][UP_];
  __shared__ double red[4][2 * C];
  static_assert(sizeof(float) * (CON * DR * DP + C * UR * UP_) <= 62 * 1024, "LDS tiles too large");
  const int H2 = 2 * h, W2 = 2 * w;
  const int tid = threadIdx.x, r = tid / U_TW, q = tid - r * U_TW;
  const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tx_n = (w + U_TW - 1) / U_TW, ty_n = (h + TRL - 1) / TRL, per_s = tx_n * ty_n;
  const long ntiles = (long)n * per_s;
  // few weights (9 C CON <= 72): kept in registers for the whole kernel instead of re-read from LDS per position
  constexpr bool WREG = 9 * C * CON <= 72;
  float wreg[WREG ? 9 * CON : 1][C];
  if constexpr (WREG) {
#pragma unroll
    for (int k = 0; k < 9 * CON; k++)
#pragma unroll
      for (int c = 0; c < C; c++) wreg[k][c] = wn[k * C + c];   // wn: [tap][co][c], the launcher's transposed copy
  }
  double s1[C], s2[C];
  float rsv[C];
#pragma unroll
  for (int c = 0; c < C; c++) { s1[c] = 0.0; s2[c] = 0.0; rsv[c] = BN ? rsqrtf(stat[2 * c + 1] + 1e-3f) : 0.f; }
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const size_t s = tile / per_s;
    const int t = (int)(tile - (long)s * per_s), y0 = (t / tx_n) * TRL, x0 = (t % tx_n) * U_TW;
    __syncthreads();
    // the dz tile by LDS-direct loads (lane l of a wave instruction writes dword l behind the instruction's base): every load
    // of the tile is in flight at once and no register holds one - in batches of 8 loads per thread through registers the
    // fill was five HBM round trips per tile and the kernel latency-bound (6.7 ms per 4096 rows for upconv3, r04).  The
    // pad column and the cells outside the plane read a zero word.
    {
      lds_direct_rows<CON * DR, DP, DC, 4>(&dzt[0][0][0], wvu, tid & 63, 2 * x0 - 3, W2, zero, [&](int r) -> const float * {
        const int co = r / DR, Y = 2 * y0 - 3 + (r - co * DR);
        return (Y >= 0 && Y < H2) ? dzn + ((s * CON + co) * (size_t)H2 + Y) * W2 : nullptr;
      });
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the loads have landed before the barrier publishes them
    }
    __syncthreads();
    int zoff;   // weights [tap][co][c] through the scalar cache, the loads kept inside the tile loop (see f_b1_pool)
    asm volatile("s_mov_b32 %0, 0" : "=s"(zoff));
    const float *__restrict__ wt = wn + zoff;
    for (int e = tid; e < UR * UC; e += 256) {
      const int i = e / UC, j = e - i * UC;
      float acc[C];
#pragma unroll
      for (int c = 0; c < C; c++) acc[c] = 0.f;
      if constexpr (WREG) {
#pragma unroll
        for (int co = 0; co < CON; co++)
#pragma unroll
          for (int t = 0; t < 9; t++) fma_row<C>(acc, dzt[co][i - t / 3 + 2][j - t % 3 + 2], wreg[t * CON + co]);
      } else
#pragma unroll 1
      for (int co = 0; co < CON; co++)
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
          for (int kx = 0; kx < 3; kx++) {
            const float v = dzt[co][i - ky + 2][j - kx + 2];
            float wv[C];
#pragma unroll
            for (int c = 0; c < C; c++) wv[c] = wt[((ky * 3 + kx) * CON + co) * C + c];
            fma_row<C>(acc, v, wv);
          }
#pragma unroll
      for (int c = 0; c < C; c++) du[c][i][j] = acc[c];
    }
    __syncthreads();
    const int y = y0 + r, x = x0 + q;
    if (r >= TRL || y >= h || x >= w) continue;
    const float *const zps = zp + s * (size_t)(C * h) * w;   // the sample's planes (block-uniform); 32-bit offsets behind them
    float *const gs = g + s * (size_t)(C * h) * w;
    float cy[5], cx[5];
    fit_up_coef(y, h, legacy, cy);
    fit_up_coef(x, w, legacy, cx);
#pragma unroll
    for (int c = 0; c < C; c++) {
      float acc = 0.f;
      // up-res row / column 2 y - 2 never has y among its taps (either convention): index 0 of the 5-window is skipped
#pragma unroll
      for (int ky = 1; ky < 5; ky++) {
        float row = 0.f;
#pragma unroll
        for (int kx = 1; kx < 5; kx++) row = fmaf(cx[kx], du[c][2 * r + ky][2 * q + kx], row);
        acc = fmaf(cy[ky], row, acc);
      }
      const int at = (c * h + y) * w + x;   // behind the sample's planes zps / gs
      const float zv = zps[at];
      if constexpr (BN) {
        const float gv = bn_act(zv, act[2 * c], act[2 * c + 1]) > 0.f ? acc : 0.f;
        gs[at] = gv;
        s1[c] += (double)gv;
        s2[c] += (double)(gv * ((zv - stat[2 * c]) * rsv[c]));
      } else {
        gs[at] = zv > 0.f ? acc : 0.f;
      }
    }
  }
  if constexpr (BN) {
#pragma unroll
    for (int c = 0; c < C; c++) {
      const double a = wave_sum(s1[c]), b2 = wave_sum(s2[c]);
      if ((tid & 63) == 0) { red[tid >> 6][2 * c] = a; red[tid >> 6][2 * c + 1] = b2; }
    }
    __syncthreads();
    if (tid < 2 * C) part[(size_t)blockIdx.x * 2 * C + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
  }
}

// ---------------------------------------------------------------- backward 2: dz + the weight gradient
constexpr int W_TR = 8, W_FLUSH = 16;
constexpr int OFX_FIT_BW_NT = 512, OFX_FIT_BW_TR = 8;   // f_bw workgroups (256 threads on 4-row tiles: 4 % slower, r03)
// PHASE: g is ONE plane of 2 H x 2 W per sample (d loss / d heat map) and output channel 2 a + b of cell (y, x) is its
// element (2 y + a, 2 x + b) - the weight gradient of the output convolution in phase form (f_out_prep).
// POINT (the last head layer under the textbook targets): g is zero but for a 4 x 4 patch of cells per sample and channel
// (gp [n][CO][16], f_top_point_bwd; the patch starts at cell ((py - 1) / 2 - 1, (px - 1) / 2 - 1) of the sample's pointer):
// g is not read - 1.28 MB per row that were a memset and a read of zeros - and dz is written to it.
template <int CI, int CO, int SRC, int TW, bool BN, int NT, int TR, bool PHASE = false, bool POINT = false>
__global__ __launch_bounds__(NT, NT == 512 ? 4 : 2) void f_bw(int n, int H, int W, FitSrc S, float *__restrict__ g,
                                            const float *__restrict__ z, const float *__restrict__ stat,
                                            const float *__restrict__ gamma, const double *__restrict__ sums, double count,
                                            double *__restrict__ part, const float *__restrict__ gp,
                                            const ofx_transition *__restrict__ rows) {
  // (the bias gradient of a convolution in front of a BatchNorm is exactly zero: no sums kept for it, written as 0)
  constexpr int LP = TW + 2, NA = 9 * CO + (BN ? 0 : CO), NSUB = (NT / 64) / CI, NPX = TR * TW, NGRP = (NPX + 63) / 64;
  __shared__ float in[CI][TR + 2][LP];
  // up-sampling sources: the low-res activation the tile interpolates is staged first (ONE batch of loads per thread, then
  // LDS reads) - element by element from global memory the fill was 8 dependent HBM round trips and 4 bilinear set-ups per cell
  constexpr bool STG = src_is_up<SRC>;
  constexpr int FU = STG ? (lo_floats<SRC, CI, TR, TW> + NT - 1) / NT : 1;
  __shared__ float lo[STG ? lo_floats<SRC, CI, TR, TW> : 1];
  __shared__ float dzt[CO][TR][TW];
  __shared__ double dacc[NT / 64][NA];
  // BatchNorm's backward per channel: dz = a (g - m0 - (z - mean) c).  Computed once: written out per element, the two
  // double divisions by `count` alone were ~100 instructions for every one of a tile's 6 400 gradient values.
  // m0, c and the batch mean are carried as two floats each (value + what its rounding dropped; stat[16 + c] for the
  // mean): their rounding error is the SAME at every pixel of the batch, so it does not average out in the weight
  // gradient sum in * dz - at 1024 rows fp32 coefficients left conv2's kernel gradient 7e-4 of its scale away from the
  // float64 graph, 1e-3 for conv1's gamma (r04, tools/fit_check64.py; the plain form computes them per element in doubles).
  __shared__ float cf[BN ? CO : 1][7];
  if constexpr (BN)
    if (threadIdx.x < CO) {
      const int co = threadIdx.x;
      const float rs = rsqrtf(stat[2 * co + 1] + 1e-3f);
      const double m0 = sums[2 * co] / count, c = (double)rs * (sums[2 * co + 1] / count);
      cf[co][0] = gamma[co] * rs;
      cf[co][1] = (float)m0;
      cf[co][2] = (float)c;
      cf[co][3] = stat[2 * co];
      cf[co][4] = (float)(m0 - (double)cf[co][1]);
      cf[co][5] = (float)(c - (double)cf[co][2]);
      cf[co][6] = stat[16 + co];
    }
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int ci = wv % CI, sub = wv / CI;
  const int tx_n = W / TW, ty_n = (H + TR - 1) / TR, per_s = tx_n * ty_n;
  const long ntiles = (long)n * per_s;
  for (int e = tid; e < (NT / 64) * NA; e += NT) (&dacc[0][0])[e] = 0.0;
  float acc[NA];
#pragma unroll
  for (int k = 0; k < NA; k++) acc[k] = 0.f;
  auto flush = [&]() {
#pragma unroll
    for (int k = 0; k < NA; k++) {
      const float v = wave_sum(acc[k]);
      if (lane == 0) dacc[wv][k] += (double)v;
      acc[k] = 0.f;
    }
  };
  int since = 0;
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const size_t s = tile / per_s;
    const int t = (int)(tile - (long)s * per_s), y0 = (t / tx_n) * TR, x0 = (t % tx_n) * TW;
    int pk0 = 0, pm0 = 0;   // POINT: first cell of the sample's patch
    if constexpr (POINT) {
      pk0 = (((min(max(rows[s].py, 0), 399)) - 1) >> 1) - 1;
      pm0 = (((min(max(rows[s].px, 0), 399)) - 1) >> 1) - 1;
    }
    __syncthreads();
    fill_input<SRC, CI, TR, TW, LP, NT, FU, STG>(in, lo, S, s, y0, x0, H, W, tid);
    // V gradient values per step (16-byte loads where the rows allow): the index arithmetic is paid once per V values
    constexpr int V = TW % 4 == 0 ? 4 : 2, TWV = TW / V;
    static_assert(TW % V == 0, "tile width");
    if constexpr (PHASE) {
      static_assert(CO == 4 && V == 4 && !BN, "phase form");
      for (int e = tid; e < 2 * TR * TWV; e += NT) {
        const int a = e / (TR * TWV), rem = e - a * (TR * TWV), yy = rem / TWV, xv = rem - yy * TWV;
        const int y = y0 + yy;
        float4 lo4 = make_float4(0.f, 0.f, 0.f, 0.f), hi4 = lo4;
        if (y < H) {
          const float *gp = g + (s * 2 * H + 2 * y + a) * (size_t)(2 * W) + 2 * (x0 + 4 * xv);
          lo4 = *reinterpret_cast<const float4 *>(gp);
          hi4 = *reinterpret_cast<const float4 *>(gp + 4);
        }
        float *d0 = &dzt[2 * a][yy][4 * xv], *d1 = &dzt[2 * a + 1][yy][4 * xv];
        d0[0] = lo4.x; d1[0] = lo4.y; d0[1] = lo4.z; d1[1] = lo4.w;
        d0[2] = hi4.x; d1[2] = hi4.y; d0[3] = hi4.z; d1[3] = hi4.w;
      }
    } else
    for (int e = tid; e < CO * TR * TWV; e += NT) {
      const int co = e / (TR * TWV), rem = e - co * (TR * TWV), yy = rem / TWV, xv = rem - yy * TWV;
      const int y = y0 + yy;
      float d[V];
#pragma unroll
      for (int k = 0; k < V; k++) d[k] = 0.f;
      if (y < H) {
        const size_t at = s * (size_t)(CO * H) * W + (unsigned)((co * H + y) * W + x0 + V * xv);   // uniform 64-bit part + 32-bit offset
        float zz[V];
        if constexpr (POINT) {
          static_assert(!POINT || (V == 4 && BN), "patch form");
          const float4 z4 = *reinterpret_cast<const float4 *>(z + at);
          zz[0] = z4.x; zz[1] = z4.y; zz[2] = z4.z; zz[3] = z4.w;
          if ((unsigned)(y - pk0) < 4u) {
#pragma unroll
            for (int k = 0; k < V; k++) {
              const int xx = x0 + V * xv + k - pm0;
              if ((unsigned)xx < 4u) d[k] = gp[(s * CO + co) * 16 + (y - pk0) * 4 + xx];
            }
          }
        } else if constexpr (V == 4) {
          const float4 g4 = *reinterpret_cast<const float4 *>(g + at);
          d[0] = g4.x; d[1] = g4.y; d[2] = g4.z; d[3] = g4.w;
          if constexpr (BN) { const float4 z4 = *reinterpret_cast<const float4 *>(z + at); zz[0] = z4.x; zz[1] = z4.y; zz[2] = z4.z; zz[3] = z4.w; }
        } else {
          const float2 g2 = *reinterpret_cast<const float2 *>(g + at);
          d[0] = g2.x; d[1] = g2.y;
          if constexpr (BN) { const float2 z2 = *reinterpret_cast<const float2 *>(z + at); zz[0] = z2.x; zz[1] = z2.y; }
        }
        if constexpr (BN) {
#pragma unroll
          for (int k = 0; k < V; k++) {
            const float zm = (zz[k] - cf[co][3]) - cf[co][6];
            d[k] = cf[co][0] * (((d[k] - cf[co][1]) - zm * cf[co][2]) - (zm * cf[co][5] + cf[co][4]));
          }
          if constexpr (SRC != OFX_FIT_SRC_BITS) {   // nothing reads the first layer's dz
            if constexpr (V == 4) *reinterpret_cast<float4 *>(g + at) = make_float4(d[0], d[1], d[2], d[3]);
            else *reinterpret_cast<float2 *>(g + at) = make_float2(d[0], d[1]);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < V; k++) dzt[co][yy][V * xv + k] = d[k];
    }
    fill_landed<SRC>();
    __syncthreads();
    for (int grp = sub; grp < NGRP; grp += NSUB) {
      const int p = 64 * grp + lane;
      const bool ok = p < NPX;
      const int pp = ok ? p : 0, yy = pp / TW, xx = pp - yy * TW;
      float dv[CO];
#pragma unroll
      for (int co = 0; co < CO; co++) dv[co] = ok ? dzt[co][yy][xx] : 0.f;
#pragma unroll
      for (int ky = 0; ky < 3; ky++)
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
          const float v = in[ci][yy + ky][xx + kx];
          if constexpr (CO % 2 == 0) {
#pragma unroll
            for (int co = 0; co < CO; co += 2)
              fma2(acc[(ky * 3 + kx) * CO + co], acc[(ky * 3 + kx) * CO + co + 1], v, dv[co], dv[co + 1]);
          } else {
#pragma unroll
            for (int co = 0; co < CO; co++) acc[(ky * 3 + kx) * CO + co] = fmaf(v, dv[co], acc[(ky * 3 + kx) * CO + co]);
          }
        }
      if constexpr (!BN) {
#pragma unroll
        for (int co = 0; co < CO; co++) acc[9 * CO + co] += dv[co];
      }
    }
    if (++since == W_FLUSH) { flush(); since = 0; }
  }
  flush();
  __syncthreads();
  // part[block] = [tap][ci][co] weights, then [co] bias: the layout of the weight tensors (HWIO)
  for (int k = tid; k < 9 * CI * CO + CO; k += NT) {
    double v = 0.0;
    if (k < 9 * CI * CO) {
      const int co = k % CO, c = (k / CO) % CI, tap = k / (CO * CI);
      for (int sb = 0; sb < NSUB; sb++) v += dacc[sb * CI + c][tap * CO + co];
    } else if constexpr (!BN) {
      const int co = k - 9 * CI * CO;
      for (int sb = 0; sb < NSUB; sb++) v += dacc[sb * CI][9 * CO + co];
    }
    part[(size_t)blockIdx.x * (9 * CI * CO + CO) + k] = v;
  }
}

// The same for layers with few weights (CI x CO <= 8: upconv1, upconv2, the output convolution): a wave per input channel
// would read one LDS value per FMA there.  A thread owns 4 pixels of a row and ALL 9 CI CO sums: 6 input values of a row
// serve 12 FMAs per output channel.
template <int CI, int CO, int SRC, int TW, bool BN>
__global__ __launch_bounds__(256) void f_bw_small(int n, int H, int W, FitSrc S, float *__restrict__ g,
                                                  const float *__restrict__ z, const float *__restrict__ stat,
                                                  const float *__restrict__ gamma, const double *__restrict__ sums,
                                                  double count, double *__restrict__ part) {
  constexpr int TPR = (TW + 3) / 4, LP = 4 * TPR + 4, DP = 4 * TPR, NA = 9 * CI * CO + CO;
  static_assert(W_TR * TPR <= 256 && CI * CO <= 8, "tile / accumulators do not fit");
  __shared__ __align__(16) float in[CI][W_TR + 2][LP];
  __shared__ float lo[lo_floats<SRC, CI, W_TR, TW>];
  __shared__ __align__(16) float dzt[CO][W_TR][DP];
  __shared__ double dacc[4][NA];
  __shared__ float cf[BN ? CO : 1][7];   // see f_bw
  if constexpr (BN)
    if (threadIdx.x < CO) {
      const int co = threadIdx.x;
      const float rs = rsqrtf(stat[2 * co + 1] + 1e-3f);
      const double m0 = sums[2 * co] / count, c = (double)rs * (sums[2 * co + 1] / count);
      cf[co][0] = gamma[co] * rs;
      cf[co][1] = (float)m0;
      cf[co][2] = (float)c;
      cf[co][3] = stat[2 * co];
      cf[co][4] = (float)(m0 - (double)cf[co][1]);
      cf[co][5] = (float)(c - (double)cf[co][2]);
      cf[co][6] = stat[16 + co];
    }
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, r = tid / TPR, q = tid - r * TPR;
  const bool active = r < W_TR;
  const int tx_n = W / TW, ty_n = (H + W_TR - 1) / W_TR, per_s = tx_n * ty_n;
  const long ntiles = (long)n * per_s;
  for (int e = tid; e < 4 * NA; e += 256) (&dacc[0][0])[e] = 0.0;
  float acc[NA];
#pragma unroll
  for (int k = 0; k < NA; k++) acc[k] = 0.f;
  auto flush = [&]() {
#pragma unroll
    for (int k = 0; k < NA; k++) {
      const float v = wave_sum(acc[k]);
      if (lane == 0) dacc[wv][k] += (double)v;
      acc[k] = 0.f;
    }
  };
  int since = 0;
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const size_t s = tile / per_s;
    const int t = (int)(tile - (long)s * per_s), y0 = (t / tx_n) * W_TR, x0 = (t % tx_n) * TW;
    __syncthreads();
    fill_input<SRC, CI, W_TR, TW, LP, 256>(in, lo, S, s, y0, x0, H, W, tid);
    for (int e0 = tid; e0 < CO * W_TR * DP; e0 += 256 * 8) {
      float gv[FILL_U], zv[FILL_U];
#pragma unroll
      for (int u = 0; u < FILL_U; u++) {
        const int e = e0 + u * 256;
        const int co = e / (W_TR * DP), rem = e - co * (W_TR * DP), yy = rem / DP, xx = rem - yy * DP;
        const bool ok = e < CO * W_TR * DP && y0 + yy < H && xx < TW;
        const size_t at = ((s * CO + co) * (size_t)H + y0 + yy) * W + x0 + xx;
        gv[u] = ok ? g[at] : 0.f;
        zv[u] = (BN && ok) ? z[at] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < FILL_U; u++) {
        const int e = e0 + u * 256;
        if (e >= CO * W_TR * DP) continue;
        const int co = e / (W_TR * DP), rem = e - co * (W_TR * DP), yy = rem / DP, xx = rem - yy * DP;
        float d = gv[u];
        if constexpr (BN) {
          if (y0 + yy < H && xx < TW) {
            const float zm = (zv[u] - cf[co][3]) - cf[co][6];
            d = cf[co][0] * (((d - cf[co][1]) - zm * cf[co][2]) - (zm * cf[co][5] + cf[co][4]));
            g[((s * CO + co) * (size_t)H + y0 + yy) * W + x0 + xx] = d;
          }
        }
        dzt[co][yy][xx] = d;
      }
    }
    fill_landed<SRC>();
    __syncthreads();
    if (active) {
      float dv[CO][4];
#pragma unroll
      for (int co = 0; co < CO; co++) {
        const float4 d4 = *reinterpret_cast<const float4 *>(&dzt[co][r][4 * q]);
        dv[co][0] = d4.x; dv[co][1] = d4.y; dv[co][2] = d4.z; dv[co][3] = d4.w;
        acc[9 * CI * CO + co] += (d4.x + d4.y) + (d4.z + d4.w);
      }
#pragma unroll
      for (int ci = 0; ci < CI; ci++)
#pragma unroll
        for (int ky = 0; ky < 3; ky++) {
          const float4 a4 = *reinterpret_cast<const float4 *>(&in[ci][r + ky][4 * q]);
          const float2 a2 = *reinterpret_cast<const float2 *>(&in[ci][r + ky][4 * q + 4]);
          const float v[6] = {a4.x, a4.y, a4.z, a4.w, a2.x, a2.y};
#pragma unroll
          for (int kx = 0; kx < 3; kx++)
#pragma unroll
            for (int co = 0; co < CO; co++)
#pragma unroll
              for (int px = 0; px < 4; px++)
                acc[(ci * CO + co) * 9 + ky * 3 + kx] = fmaf(v[px + kx], dv[co][px], acc[(ci * CO + co) * 9 + ky * 3 + kx]);
        }
    }
    if (++since == W_FLUSH) { flush(); since = 0; }
  }
  flush();
  __syncthreads();
  for (int k = tid; k < NA; k += 256) {
    int a = k;
    if (k < 9 * CI * CO) {
      const int co = k % CO, c = (k / CO) % CI, tap = k / (CO * CI);
      a = (c * CO + co) * 9 + tap;
    }
    part[(size_t)blockIdx.x * NA + k] = (dacc[0][a] + dacc[1][a]) + (dacc[2][a] + dacc[3][a]);
  }
}

// block = 16 values x 16 row shares, combined in share order (fixed)
__global__ __launch_bounds__(256) void f_bw_finish(int nv, int nw, int nblocks, const double *part, float *dw, float *db,
                                                   int c_n, const double *sums, float *dgamma, float *dbeta) {
  __shared__ double share[16][17];
  const int k = blockIdx.x * 16 + (threadIdx.x & 15), j = threadIdx.x >> 4;
  double acc = 0.0;
  if (k < nv)
    for (int i = j; i < nblocks; i += 16) acc += part[(size_t)i * nv + k];
  share[threadIdx.x & 15][j] = acc;
  __syncthreads();
  if (threadIdx.x < 16) {
    const int kk = blockIdx.x * 16 + threadIdx.x;
    double t = 0.0;
    for (int q = 0; q < 16; q++) t += share[threadIdx.x][q];
    if (kk < nw) dw[kk] = (float)t;
    else if (kk < nv) db[kk - nw] = (float)t;
    if (sums && kk < c_n) { dbeta[kk] = (float)sums[2 * kk]; dgamma[kk] = (float)sums[2 * kk + 1]; }
  }
}

// ---------------------------------------------------------------- the output convolution (8 -> 1 at 400 x 400) in phase form
// o2 = conv3x3(up2(a)), a = relu(bn(z)) of the last head layer.  Up-sampling 8 planes to 400 x 400 in every tile cost
// more than the convolution itself (8x the useful instructions, rocprofv3 r03).  Along an axis
//   up(2 k + t) for t = -1, 0, 1, 2  =  R[t + 1][0] a[k - 1] + R[t + 1][1] a[k] + R[t + 1][2] a[k + 1]
// with a[] repeated outwards at the plane's edge, so the 2 x 2 outputs of low-res cell (k, m) are a 3 x 3 convolution
// of `a` with 4 output channels (phase 2 a + b):  Weff[dy][dx][c][2 a + b] = sum_{ky, kx} w[ky][kx][c] R[a + ky][dy] R[b + kx][dx].
// Forward, weight gradient (Q[dy][dx][c][phase] mapped back through R) run on f_conv_fwd / f_bw with the ACTREP source.
// What the phase form gets wrong is the zero padding of the convolution at the plane's frame: it sees the up-sampled plane
// REPEATED outwards (up(-1) = up(0), up(2 h) = up(2 h - 1)).  The two frame kernels take those terms out again.
__device__ __forceinline__ float out_r(int t1, int d, int legacy) {   // R[t + 1][d]
  const float wm0 = legacy ? 0.f : 0.25f, w00 = legacy ? 1.f : 0.75f, w01 = legacy ? 0.5f : 0.75f, wp1 = legacy ? 0.5f : 0.25f;
  const float R[4][3] = {{w01, wp1, 0.f}, {wm0, w00, 0.f}, {0.f, w01, wp1}, {0.f, wm0, w00}};
  return R[t1][d];
}
// weff[3][3][8][4] + bias4[4] from w[3][3][8][1], b[1]
__global__ void f_out_prep(const float *w, const float *b, int legacy, float *weff) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < 288) {
    const int p = e & 3, c = (e >> 2) & 7, dx = (e >> 5) % 3, dy = e / 96, a = p >> 1, bb = p & 1;
    float acc = 0.f;
    for (int ky = 0; ky < 3; ky++)
      for (int kx = 0; kx < 3; kx++) acc += w[(ky * 3 + kx) * 8 + c] * out_r(a + ky, dy, legacy) * out_r(bb + kx, dx, legacy);
    weff[e] = acc;
  } else if (e < 292) {
    weff[e] = b[0];
  }
}
// frame pixel f of a 400 x 400 plane (1596 of them): rows 0 and 399, then columns 0 and 399 without their corners
__device__ __forceinline__ void frame_px(int f, int &Y, int &X) {
  if (f < 400) { Y = 0; X = f; }
  else if (f < 800) { Y = 399; X = f - 400; }
  else if (f < 1198) { Y = f - 799; X = 0; }
  else { Y = f - 1197; X = 399; }
}
constexpr int kFrame = 1596;
// forward: o2[Y][X] -= sum over the taps that fall outside the plane of w U[clamped tap]   (U = up2(a), SRC_UP)
__global__ void f_out_frame_fwd(int n, FitSrc S, const float *w, float *o2) {
  const size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (e >= (size_t)n * kFrame) return;
  const size_t s = e / kFrame;
  int Y, X;
  frame_px((int)(e - s * kFrame), Y, X);
  float acc = 0.f;
  for (int ky = 0; ky < 3; ky++)
    for (int kx = 0; kx < 3; kx++) {
      const int y = Y + ky - 1, x = X + kx - 1;
      if (y >= 0 && y < 400 && x >= 0 && x < 400) continue;
      const int yc = min(max(y, 0), 399), xc = min(max(x, 0), 399);
      for (int c = 0; c < 8; c++) acc += w[(ky * 3 + kx) * 8 + c] * src_value<OFX_FIT_SRC_UP, 8>(S, s, c, yc, xc, 400, 400);
    }
  o2[s * 160000 + (size_t)Y * 400 + X] -= acc;
}
// weight gradient: fpart[sample][72] = sum over the frame pixels and their outside taps of D U[clamped tap]; block = sample
__global__ __launch_bounds__(256) void f_out_frame_bw(int n, FitSrc S, const float *d2, double *fpart) {
  __shared__ double red[4][72];
  const size_t s = blockIdx.x;
  float acc[72];
#pragma unroll
  for (int k = 0; k < 72; k++) acc[k] = 0.f;
  for (int f = threadIdx.x; f < kFrame; f += 256) {
    int Y, X;
    frame_px(f, Y, X);
    const float d = d2[s * 160000 + (size_t)Y * 400 + X];
#pragma unroll
    for (int ky = 0; ky < 3; ky++)
#pragma unroll
      for (int kx = 0; kx < 3; kx++) {
        const int y = Y + ky - 1, x = X + kx - 1;
        if (y >= 0 && y < 400 && x >= 0 && x < 400) continue;
        const int yc = min(max(y, 0), 399), xc = min(max(x, 0), 399);
#pragma unroll
        for (int c = 0; c < 8; c++)
          acc[(ky * 3 + kx) * 8 + c] = fmaf(d, src_value<OFX_FIT_SRC_UP, 8>(S, s, c, yc, xc, 400, 400), acc[(ky * 3 + kx) * 8 + c]);
      }
  }
#pragma unroll
  for (int k = 0; k < 72; k++) {
    const float v = wave_sum(acc[k]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = (double)v;
  }
  __syncthreads();
  if (threadIdx.x < 72) fpart[s * 72 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// out[k] = sum over the rows of part[nrows][nv], two levels in a fixed order (block = 16 values x 16 row shares)
__global__ __launch_bounds__(256) void f_sum_rows(int nv, int nrows, const double *part, double *out) {
  __shared__ double share[16][17];
  const int k = blockIdx.x * 16 + (threadIdx.x & 15), j = threadIdx.x >> 4;
  double acc = 0.0;
  if (k < nv)
    for (int i = j; i < nrows; i += 16) acc += part[(size_t)i * nv + k];
  share[threadIdx.x & 15][j] = acc;
  __syncthreads();
  if (threadIdx.x < 16 && blockIdx.x * 16 + threadIdx.x < nv) {
    double t = 0.0;
    for (int q = 0; q < 16; q++) t += share[threadIdx.x][q];
    out[blockIdx.x * 16 + threadIdx.x] = t;
  }
}
// q[292] = Q[dy][dx][c][phase] + the 4 phase sums of D, fr[72] = the frame terms:
// dw[ky][kx][c] = sum R[a + ky][dy] R[b + kx][dx] Q[dy][dx][c][2 a + b] - fr;  db = sum of D
__global__ void f_out_bw_finish(const double *q, const double *fr, int legacy, float *dw, float *db) {
  const int k = threadIdx.x;
  if (k < 72) {
    const int c = k & 7, kx = (k >> 3) % 3, ky = k / 24;
    double acc = 0.0;
    for (int a = 0; a < 2; a++)
      for (int b = 0; b < 2; b++)
        for (int dy = 0; dy < 3; dy++)
          for (int dx = 0; dx < 3; dx++)
            acc += (double)(out_r(a + ky, dy, legacy) * out_r(b + kx, dx, legacy)) * q[((dy * 3 + dx) * 8 + c) * 4 + 2 * a + b];
    dw[k] = (float)(acc - fr[k]);
  } else if (k == 72) {
    db[0] = (float)((q[288] + q[289]) + (q[290] + q[291]));
  }
}

// ---------------------------------------------------------------- the first layer's weight gradient without z0 and without dz0
// Layer 0 reads 1-bit maps, and BatchNorm's backward is affine in g and z:  dz = a (g - m0 - (z - mean) c)  gives
//   dW[u][co] = sum_p in_u[p] dz[co][p] = a_co ( A[u][co] - m0_co B[u] - c_co ( T[u][co] - mean_co B[u] ) ),  u = (tap, ci),
//   A = sum in_u g,   B[u] = sum in_u,   T[u][co] = sum in_u z[co] = b_co B[u] + sum_u' w[u'][co] Cc[u][u'],
//   Cc[u][u'] = sum_p in_u[p] in_u'[p]  - the autocorrelation of the shifted bit maps (popcounts; B is its diagonal).
// So the 5.12 MB per row of z0 are not read again and dz0 is never formed; A comes out of the kernel that forms g0
// (f_first_bwd, below).  The bias gradient of a convolution in front of BatchNorm is exactly 0.
constexpr int C0_ROWS = 40;   // rows of a correlation chunk
// part[block][324]: Cc over the block's rows; block = 384 threads (all stage the shifted rows, 171 count the pairs ua <= ub)
__global__ __launch_bounds__(384) void f_bits_corr(int n, const uint32_t *__restrict__ bits, double *__restrict__ part) {
  constexpr int RB = 8;              // image rows per barrier pair (one row at a time the kernel was all barrier: 1.54 ms per 4096 rows)
  static_assert(C0_ROWS % RB == 0, "rows per step");
  __shared__ uint32_t ver[RB][18][13];   // the 18 shifted versions of an image row: bit x of version u = in_u at pixel (y, x)
  const int tid = threadIdx.x, nchunks = n * (400 / C0_ROWS);
  // Cc is symmetric: thread q < 171 owns the pair (ua <= ub) and writes both entries (half the popcounts: the kernel is bound
  // by its vector instructions)
  int ua = 0, ub = 0;
  if (tid < 171) {
    int rem = tid;
    while (rem >= 18 - ua) { rem -= 18 - ua; ua++; }
    ub = ua + rem;
  }
  int acc = 0;   // at most (chunks per block) x 40 x 400 < 2^31
  for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const int s = chunk / (400 / C0_ROWS), y0 = (chunk % (400 / C0_ROWS)) * C0_ROWS;
    for (int yb = y0; yb < y0 + C0_ROWS; yb += RB) {
      __syncthreads();
      for (int e = tid; e < RB * 18 * 13; e += 384) {
        const int wd = e % 13, u = (e / 13) % 18, y = yb + e / (18 * 13), tap = u >> 1, ci = u & 1, r = y + tap / 3 - 1, dx = tap % 3 - 1;
        uint32_t v = 0;
        if (r >= 0 && r < 400) {
          const uint32_t *pl = bits + ((size_t)s * 2 + ci) * 5000;
          const int x0 = 32 * wd + dx;                     // plane column of the version's bit 0 of this word
          const long b0 = 400L * r + x0;                   // its bit position in the plane (may be -1 at x0 = -1)
          const long bb = b0 < 0 ? 0 : b0;
          const int wi = (int)(bb >> 5), sh = (int)(bb & 31);
          const uint32_t lo = pl[wi], hi = wi + 1 < 5000 ? pl[wi + 1] : 0u;
          v = sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
          if (b0 < 0) v <<= 1;                             // column -1 does not exist: bit 0 of the word is pixel column 0's left
          // keep the bits whose column x0 + b lies in [0, 400)
          uint32_t m = 0xFFFFFFFFu;
          if (x0 < 0) m &= ~1u;
          const int over = x0 + 32 - 400;
          if (over > 0) m &= over >= 32 ? 0u : (0xFFFFFFFFu >> over);
          if (wd == 12) m &= 0xFFFFu;                      // ... and whose OUTPUT pixel 32 wd + b lies in the row (400 = 12.5 words)
          v &= m;
        }
        ver[e / (18 * 13)][u][wd] = v;
      }
      __syncthreads();
      if (tid < 171) {
#pragma unroll
        for (int k = 0; k < RB; k++)
#pragma unroll
          for (int wd = 0; wd < 13; wd++) acc += __popc(ver[k][ua][wd] & ver[k][ub][wd]);
      }
    }
  }
  if (tid < 171) {
    part[(size_t)blockIdx.x * 324 + ua * 18 + ub] = (double)acc;
    part[(size_t)blockIdx.x * 324 + ub * 18 + ua] = (double)acc;
  }
}

// ---- the first layer is never materialised -------------------------------------------------------------------------------
// z0 = conv3x3(1-bit maps) takes one of 512 values per channel and input map, so (1) its batch statistics follow from the
// autocorrelation Cc of the shifted bit maps: sum z = M b + sum_u w_u B_u, sum z^2 = M b^2 + 2 b sum_u w_u B_u + sum_uu'
// w_u w_u' Cc[u][u'] (exact integers times weights, in doubles); (2) pool(relu(bn(z0))) - all the second layer reads - is the
// forward's table kernel (k_conv1_lut, ofx_policy.hip) on a table that folds the BATCH statistics; (3) the backward
// through the pooling looks x-hat up in a second table.  The 5.12 MB per row of z0 are neither written nor read.
// lut_y[ci][pattern][co] = scale_co sum of the set taps' weights (+ scale b + shift for ci = 0): relu(bn(z0)) = max(y, 0)
// lut_x: the same with rs instead of scale and rs (b - mean) for ci = 0: x-hat
__global__ __launch_bounds__(256) void f_first_prepare(const double *cc, const float *w, const float *b, const float *gamma,
                                                       const float *beta, double count, float *stat, float *act,
                                                       float *lut_y, float *lut_x) {
  __shared__ float sc[8], sh[8], rsv[8], mn[8];
  const int tid = threadIdx.x;
  if (tid < 8) {
    const int co = tid;
    double sw = 0.0, sww = 0.0;
    for (int u = 0; u < 18; u++) {
      sw += (double)w[u * 8 + co] * cc[u * 18 + u];
      for (int v = 0; v < 18; v++) sww += (double)w[u * 8 + co] * (double)w[v * 8 + co] * cc[u * 18 + v];
    }
    const double bb = b[co], mean = bb + sw / count, ez2 = bb * bb + (2.0 * bb * sw + sww) / count, var = ez2 - mean * mean;
    const float meanf = (float)mean, varf = (float)(var > 0.0 ? var : 0.0);
    stat[2 * co] = meanf;
    stat[2 * co + 1] = varf;
    stat[16 + co] = (float)(mean - (double)meanf);
    const float rs = rsqrtf(varf + 1e-3f), s_ = gamma[co] * rs;
    act[2 * co] = s_;
    act[2 * co + 1] = beta[co] - meanf * s_;
    sc[co] = s_; sh[co] = beta[co] - meanf * s_; rsv[co] = rs; mn[co] = meanf;
  }
  __syncthreads();
  for (int e = tid; e < 2 * 512 * 8; e += 256) {
    const int co = e & 7, pat = (e >> 3) & 511, ci = e >> 12;
    float acc = 0.f;
    for (int tap = 0; tap < 9; tap++)
      if ((pat >> tap) & 1) acc += w[(tap * 2 + ci) * 8 + co];
    lut_y[e] = acc * sc[co] + (ci == 0 ? b[co] * sc[co] + sh[co] : 0.f);
    lut_x[e] = acc * rsv[co] + (ci == 0 ? (b[co] - mn[co]) * rsv[co] : 0.f);
  }
}

// ---- the backward of the first TWO layers over the windows that see a set bit ----------------------------------------------
// Layer 0 reads 1-bit maps that are ~1 % set.  A 2 x 2 pooling window whose 4 x 4 bit neighbourhood is EMPTY in both maps
// has z0 = b at its four pixels: the same x-hat xc, the same activation, its first pixel as the maximum, no input under
// any tap of that pixel - and the pooled activation the second layer reads there is the constant K1 = relu(bn(b)).  So
// (1) the first layer: an empty window adds nothing to A = sum in_u g; to BatchNorm's sums it adds [K1 > 0] dp and
//     [K1 > 0] xc dp, with dp the transposed convolution of the second layer's dz at the window;
// (2) the second layer's weight gradient: with in = K1 + delta (delta = 0 on empty windows, in = 0 outside the plane)
//     dW2[t][ci][co] = K1[ci] (sum of dz[co] over the cells p with p + t - 1 inside) + sum over the LISTED windows q of
//     delta[ci][q] dz[co][q - (t - 1)];
// (3) sums of dz over the plane with a border row / column left out are all either needs from the unlisted part: the sum
//     of dp over ALL windows is linear in them, and the batch total of dz itself is ZERO - the second layer's dz comes out
//     of BatchNorm's backward, dz = a (g - mean g - xhat mean(g xhat)), which sums to zero per channel (taking the exact
//     total of stored fp32 values in doubles instead gave the same test results; the float total the weight-gradient kernel
//     has for the bias moved conv1's beta gradient by 7e-4 of its scale - r04);
//     sum over the empty windows of dp = (sum over all) - (sum over the listed windows).
// Only the windows that see a bit are evaluated: ~2.5 % on arena observations (every window on a dense map - the result is
// the same, only the time differs), and dz of the second layer is never stored: it is an element-wise function of g and z
// of that layer (f_bw's formula), formed where it is read - 9 x 8 cells around a listed window, the border lines.  Gone
// per 4096 rows: the dense transposed convolution (94 GMAC), the compact g0 and its kernel (f_b1_first + f_bw_first: 13.6 ms),
// the second layer's f_bw (6.7 ms: 1.28 MB per row written, 4 MB read).
// Block = sample.  Per band of 20 pooled rows the bit rows go to LDS and every wave classifies 5 rows x 200 windows into
// its queue (ballot order: fixed; an entry carries its 4 x 4 bits); whenever 64 are queued the wave works them off, one
// window per lane.  The sums that cross lanes go through LDS records in queue order: A per (tap, map, channel) lane, dW2's
// listed part per (quarter of the records, ci, co of a pair) lane.
constexpr int FS_BR = 20, FS_WR = FS_BR / 4, FS_NW = FS_WR * 200, FS_ROWS = 2 * FS_BR + 2, FS_WORDS = 14;
constexpr int FS_NV = 144 + 24 + 576;   // a block's row of sums: A | sum g, sum g xhat, sum dp (8 each) | dW2's listed part
struct FsEntry {
  float gv[8];          // g of the window per channel (0 behind the ReLU)
  uint32_t kpack;       // window position of the maximum, 2 bits per channel
  uint16_t pat[2][4];   // the 3 x 3 bit pattern around each of the window's four pixels, per map
};
// BatchNorm's backward of the second layer per channel, as f_bw carries it: {a, m0, c, mean, m0 low, c low, mean low}
__device__ __forceinline__ void fs_coef(int co, const float *stat, const float *gamma, const double *sums, double count, float *cf) {
  const float rs = rsqrtf(stat[2 * co + 1] + 1e-3f);
  const double m0 = sums[2 * co] / count, c = (double)rs * (sums[2 * co + 1] / count);
  cf[0] = gamma[co] * rs;
  cf[1] = (float)m0;
  cf[2] = (float)c;
  cf[3] = stat[2 * co];
  cf[4] = (float)(m0 - (double)cf[1]);
  cf[5] = (float)(c - (double)cf[2]);
  cf[6] = stat[16 + co];
}
__device__ __forceinline__ float fs_dz(float g, float z, const float *cf) {
  const float zm = (z - cf[3]) - cf[6];
  return cf[0] * (((g - cf[1]) - zm * cf[2]) - (zm * cf[5] + cf[4]));
}
__global__ __launch_bounds__(256, 2) void f_first_bwd(int n, const uint32_t *__restrict__ bits, const float *__restrict__ g1,
                                                   const float *__restrict__ z1, const float *__restrict__ stat1,
                                                   const float *__restrict__ gamma1, const double *__restrict__ sums1,
                                                   const float *__restrict__ wn, const float *__restrict__ p0,
                                                   const float *__restrict__ lut_y, const float *__restrict__ lut_x,
                                                   const float *__restrict__ gamma, const float *__restrict__ beta,
                                                   double *__restrict__ part) {
  constexpr int C = 8, H = 400, W = 400, Hp = 200, Wp = 200, NSTEP = (Hp / FS_BR) * ((FS_NW + 63) / 64);
  __shared__ __align__(16) float wl[9 * C * C];   // [tap][co][ci]
  __shared__ float cfl[C][8], gbk[3][C];            // the second layer's coefficients; gamma, beta, K1 of the first
  __shared__ uint32_t rows[2][FS_ROWS][FS_WORDS];   // bit i of a staged row <-> image column i - 1 (0 outside the plane)
  __shared__ uint16_t qwin[4][128];                 // a wave's queue: window index 200 yp + xp ...
  __shared__ uint32_t qbits[4][128];                // ... and its 4 x 4 bits, 4 per (map, row): bit 16 map + 4 row + column
  __shared__ __align__(16) FsEntry ent[4][64];
  __shared__ float drec[4][64][8];                  // delta of the record's window per ci
  __shared__ float vrec[4][64][18];                 // dz around the record's window for a pair of co: [co & 1][tap]; behind
                                                    // the pairs: g xhat [8] and dp [8] of the record
  __shared__ double dw2[4][C * 72];                 // a wave's dW2 sums [co][tap][ci]
  __shared__ double red[4][168];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const size_t s = blockIdx.x;
  for (int e = tid; e < 9 * C * C; e += 256) {
    const int ci = e % C, co = (e / C) % C, tap = e / (C * C);
    wl[e] = wn[(tap * C + ci) * C + co];
  }
  if (tid < C) fs_coef(tid, stat1, gamma1, sums1, (double)n * (Hp * Wp), cfl[tid]);
  for (int e = lane; e < C * 72; e += 64) dw2[wv][e] = 0.0;
  if (tid < C) {
    gbk[0][tid] = gamma[tid]; gbk[1][tid] = beta[tid];
    gbk[2][tid] = fmaxf(lut_y[tid] + lut_y[512 * 8 + tid], 0.f);   // the pooled activation of an empty window, as k_conv1_lut forms it
  }
  double accA[3] = {0.0, 0.0, 0.0}, accS = 0.0;   // A: lane <-> (tap, map, channel); lanes 0-23: sum g | sum g xhat | sum dp per channel
  const uint64_t lt = lane ? (~0ull >> (64 - lane)) : 0ull;
  const float *g1s = g1 + s * (size_t)(C * Hp * Wp), *z1s = z1 + s * (size_t)(C * Hp * Wp), *p0s = p0 + s * (size_t)(C * Hp * Wp);
  int cnt = 0;   // queued windows of the wave (wave-uniform)
  for (int step = 0; step <= NSTEP; step++) {
    if (step < NSTEP) {
      const int band = step / ((FS_NW + 63) / 64), it = step - band * ((FS_NW + 63) / 64), y0 = band * FS_BR;
      if (it == 0) {   // block-uniform
        __syncthreads();
        // image rows 2 y0 - 1 .. 2 y0 + 2 FS_BR, columns -1 .. 400 (402 bits -> 13 words + one of zeros)
        for (int e = tid; e < 2 * FS_ROWS * FS_WORDS; e += 256) {
          const int wd = e % FS_WORDS, rr = (e / FS_WORDS) % FS_ROWS, ci = e / (FS_WORDS * FS_ROWS);
          const int gy = 2 * y0 - 1 + rr;
          uint32_t out = 0u;
          if (gy >= 0 && gy < H && wd < 13) {
            const int c0 = 32 * wd - 1;                                   // image column of the word's bit 0
            const long s0 = (long)gy * W + c0;                            // its cell (-1 only at gy = 0, wd = 0)
            const uint32_t *pl = bits + (s * 2 + ci) * 5000;
            const long sw = s0 >> 5;
            const uint32_t lo = (sw >= 0 && sw < 5000) ? pl[sw] : 0u, hi = (sw + 1 >= 0 && sw + 1 < 5000) ? pl[sw + 1] : 0u;
            out = __funnelshift_r(lo, hi, (unsigned)(s0 & 31));
            if (c0 < 0) out &= ~1u;                                       // column -1
            const int over = c0 + 32 - W;                                 // bits past the last column of THIS row
            if (over > 0) out &= 0xFFFFFFFFu >> over;
          }
          rows[ci][rr][wd] = out;
        }
        __syncthreads();
      }
      // window (yl, xp) of the wave's 5 rows: staged rows 2 r .. 2 r + 3 (r = FS_WR wv + yl), staged bits 2 xp .. 2 xp + 3
      const int wi = min(64 * it + lane, FS_NW - 1), yl = wi / Wp, xp = wi - yl * Wp, r = FS_WR * wv + yl, wd = (2 * xp) >> 5;
      const unsigned sh = (unsigned)((2 * xp) & 31);
      uint32_t fb = 0u;
#pragma unroll
      for (int ci = 0; ci < 2; ci++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint32_t *rw = &rows[ci][2 * r + k][wd];
          fb |= (__funnelshift_r(rw[0], rw[1], sh) & 15u) << (16 * ci + 4 * k);
        }
      const bool ne = 64 * it + lane < FS_NW && fb != 0u;
      const uint64_t m = __builtin_amdgcn_ballot_w64(ne);
      if (ne) {
        const int at = cnt + __builtin_popcountll(m & lt);
        qwin[wv][at] = (uint16_t)((y0 + r) * Wp + xp);
        qbits[wv][at] = fb;
      }
      cnt += __builtin_popcountll(m);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if (cnt < 64 && !(step == NSTEP && cnt > 0)) continue;   // wave-uniform
    // ---- work off the first min(64, cnt) queued windows, one per lane
    const int ne = min(64, cnt);
    const bool ok = lane < ne;
    const int qi = ok ? lane : 0;
    const int win = qwin[wv][qi], yp = win / Wp, xp = win - yp * Wp;
    const uint32_t fb = qbits[wv][qi];
    {  // delta = in - K1 of the window per input channel of the second layer
      float dl[C];
#pragma unroll
      for (int ci = 0; ci < C; ci++) dl[ci] = ok ? p0s[(ci * Hp + yp) * Wp + xp] - gbk[2][ci] : 0.f;
#pragma unroll
      for (int ci = 0; ci < C; ci++) drec[wv][lane][ci] = dl[ci];
    }
    float dp[C];
#pragma unroll
    for (int ci = 0; ci < C; ci++) dp[ci] = 0.f;
#pragma unroll 1
    for (int cp = 0; cp < C / 2; cp++) {
      float v[2][9];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int co = 2 * cp + h;
        float gq[9], zq[9];
#pragma unroll
        for (int t = 0; t < 9; t++) {
          const int y = yp - (t / 3 - 1), x = xp - (t % 3 - 1);
          const bool in = y >= 0 && y < Hp && x >= 0 && x < Wp;
          const int at = in ? (co * Hp + y) * Wp + x : 0;
          gq[t] = g1s[at]; zq[t] = z1s[at];
        }
#pragma unroll
        for (int t = 0; t < 9; t++) {
          const int y = yp - (t / 3 - 1), x = xp - (t % 3 - 1);
          v[h][t] = (ok && y >= 0 && y < Hp && x >= 0 && x < Wp) ? fs_dz(gq[t], zq[t], cfl[co]) : 0.f;
        }
      }
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int t = 0; t < 9; t++) {
          float wv8[C];
          lds_vec<C>(&wl[(t * C + 2 * cp + h) * C], wv8);
          fma_row<C>(dp, v[h][t], wv8);
          vrec[wv][lane][9 * h + t] = v[h][t];
        }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // dW2's listed part for this pair of co: lane = (quarter of the records, ci, co of the pair), nine taps each;
      // the four quarters are added in a fixed order
      {
        const int qd = lane >> 4, ci = (lane >> 1) & 7, hh = lane & 1;
        float a[9];   // 16 products per sum in fp32 (the vector ALU's fp64 runs at half rate), the sums themselves in doubles
#pragma unroll
        for (int t = 0; t < 9; t++) a[t] = 0.f;
        for (int e2 = 16 * qd; e2 < min(16 * qd + 16, ne); e2++) {
          const float d = drec[wv][e2][ci];
#pragma unroll
          for (int t = 0; t < 9; t++) a[t] = fmaf(d, vrec[wv][e2][9 * hh + t], a[t]);
        }
#pragma unroll
        for (int t = 0; t < 9; t++) {
          const double a0 = (double)a[t], b16 = a0 + __shfl_xor(a0, 16, 64);
          const double b32 = b16 + __shfl_xor(b16, 32, 64);
          if (qd == 0) dw2[wv][((2 * cp + hh) * 9 + t) * C + ci] += b32;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    // x-hat of the window's four pixels (dy, dx): pattern = bits dx .. dx + 2 of rows dy .. dy + 2 of the 4 x 4 block
    float xh[4][C];
    FsEntry E;
#pragma unroll
    for (int px = 0; px < 4; px++) {
      const int dy = px >> 1, dx = px & 1;
      float4 lo4, hi4;
#pragma unroll
      for (int ci = 0; ci < 2; ci++) {
        const uint32_t f = fb >> (16 * ci);
        const uint32_t pat = ((f >> (4 * dy + dx)) & 7u) | (((f >> (4 * dy + 4 + dx)) & 7u) << 3) | (((f >> (4 * dy + 8 + dx)) & 7u) << 6);
        E.pat[ci][px] = (uint16_t)pat;
        const float4 *e = reinterpret_cast<const float4 *>(&lut_x[(ci * 512 + pat) * 8]);
        const float4 e0 = e[0], e1 = e[1];
        if (ci == 0) { lo4 = e0; hi4 = e1; }
        else { lo4.x += e0.x; lo4.y += e0.y; lo4.z += e0.z; lo4.w += e0.w; hi4.x += e1.x; hi4.y += e1.y; hi4.z += e1.z; hi4.w += e1.w; }
      }
      xh[px][0] = lo4.x; xh[px][1] = lo4.y; xh[px][2] = lo4.z; xh[px][3] = lo4.w;
      xh[px][4] = hi4.x; xh[px][5] = hi4.y; xh[px][6] = hi4.z; xh[px][7] = hi4.w;
    }
    E.kpack = 0u;
#pragma unroll
    for (int c = 0; c < C; c++) {
      float av[4];
#pragma unroll
      for (int i = 0; i < 4; i++) av[i] = fmaxf(fmaf(gbk[0][c], xh[i][c], gbk[1][c]), 0.f);
      int k = 0;
#pragma unroll
      for (int i = 1; i < 4; i++) if (av[i] > av[k]) k = i;
      const float xk = k == 0 ? xh[0][c] : k == 1 ? xh[1][c] : k == 2 ? xh[2][c] : xh[3][c];
      const float gv = (ok && av[k] > 0.f) ? dp[c] : 0.f;
      E.gv[c] = gv;
      E.kpack |= (uint32_t)k << (2 * c);
      vrec[wv][lane][c] = gv * xk;
      vrec[wv][lane][8 + c] = ok ? dp[c] : 0.f;
    }
    ent[wv][lane] = E;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // A[u = (tap, map)][c] += (bit `tap` of the pattern around channel c's maximum) * g: lane <-> (u, c), records in queue order
    {   // lane + 64 j <-> (u, c): c and the map are the lane's own for every j, the tap is (lane >> 4) + 4 j: one read of the
        // record's pattern serves the three; <= 64 terms per sum in fp32, the sums themselves in doubles
      const int c = lane & 7, cin = (lane >> 3) & 1, t0 = lane >> 4;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
      for (int e2 = 0; e2 < ne; e2++) {
        const FsEntry &R = ent[wv][e2];
        const uint32_t pt = R.pat[cin][(R.kpack >> (2 * c)) & 3];
        const float gvc = R.gv[c];
        a0 += ((pt >> t0) & 1u) ? gvc : 0.f;
        a1 += ((pt >> (t0 + 4)) & 1u) ? gvc : 0.f;
        a2 += ((pt >> (t0 + 8)) & 1u) ? gvc : 0.f;
      }
      accA[0] += (double)a0;
      accA[1] += (double)a1;
      if (lane < 16) accA[2] += (double)a2;
    }
    if (lane < 24) {   // BatchNorm's sums over the listed windows, records in queue order
      const int c = lane & 7, what = lane >> 3;
      float a = 0.f;
      for (int e2 = 0; e2 < ne; e2++) a += what == 0 ? ent[wv][e2].gv[c] : vrec[wv][e2][8 * (what - 1) + c];
      accS += (double)a;
    }
    // the queue moves up
    const int rest = cnt - ne;
    uint16_t mw = 0;
    uint32_t mb = 0u;
    if (lane < rest) { mw = qwin[wv][64 + lane]; mb = qbits[wv][64 + lane]; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < rest) { qwin[wv][lane] = mw; qbits[wv][lane] = mb; }
    cnt = rest;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (lane < 24) red[wv][144 + lane] = accS;
#pragma unroll
  for (int j = 0; j < 3; j++)
    if (lane + 64 * j < 144) red[wv][lane + 64 * j] = accA[j];
  __syncthreads();
  // part[sample] = A as [tap][map][channel] (144) | sum g, sum g xhat, sum dp over the listed windows (8 each) | dW2's listed
  // part as [tap][ci][co] (576)
  for (int k = tid; k < FS_NV; k += 256) {
    double v;
    if (k < 168) v = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    else {
      const int j = k - 168, co = j & 7, ci = (j >> 3) & 7, t = j >> 6, a = (co * 9 + t) * C + ci;
      v = (dw2[0][a] + dw2[1][a]) + (dw2[2][a] + dw2[3][a]);
    }
    part[s * FS_NV + k] = v;
  }
}

// bpart[sample][co][8]: sums of the sample's dz plane (200 x 200; dz from g and z of the layer, see above) over its first /
// last row, first / last column, and its four corners (0,0) (0,L) (L,0) (L,L); block = sample, wave = two channels
__global__ __launch_bounds__(256) void f_first_border(int n, const float *__restrict__ g1, const float *__restrict__ z1,
                                                      const float *__restrict__ stat1, const float *__restrict__ gamma1,
                                                      const double *__restrict__ sums1, double *__restrict__ bpart) {
  constexpr int Hp = 200, Wp = 200;
  const int s = blockIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int k = 0; k < 2; k++) {
    const int co = 2 * wv + k;
    float cf[7];
    fs_coef(co, stat1, gamma1, sums1, (double)n * (Hp * Wp), cf);
    const size_t base = ((size_t)s * 8 + co) * (Hp * Wp);
    auto dz = [&](int at) { return (double)fs_dz(g1[base + at], z1[base + at], cf); };
    double r0 = 0.0, r1 = 0.0, c0 = 0.0, c1 = 0.0;
    for (int i = lane; i < 200; i += 64) {
      r0 += dz(i);
      r1 += dz((Hp - 1) * Wp + i);
      c0 += dz(i * Wp);
      c1 += dz(i * Wp + Wp - 1);
    }
    r0 = wave_sum(r0); r1 = wave_sum(r1); c0 = wave_sum(c0); c1 = wave_sum(c1);
    if (lane == 0) {
      double *o = bpart + ((size_t)s * 8 + co) * 8;
      o[0] = r0; o[1] = r1; o[2] = c0; o[3] = c1;
      o[4] = dz(0); o[5] = dz(Wp - 1); o[6] = dz((Hp - 1) * Wp); o[7] = dz(Hp * Wp - 1);
    }
  }
}

// q[744] = the batch's A | sum g, sum g xhat, sum dp over the listed windows | dW2's listed part; bsum[64] = the border sums
// of the second layer's dz; cc[324] = the autocorrelation of the bit maps.  Out: the second layer's dw2 / db2 / dgamma2 /
// dbeta2 (sums1 = its BatchNorm sums), and the first layer's - BatchNorm's sums with the empty windows' share put back, then
// dw, db, dgamma, dbeta (see "the first layer's weight gradient" above)
__global__ __launch_bounds__(576) void f_first_bwd_finish(const double *q, const double *bsum, const float *wn,
                                                          const float *lut_y, const float *lut_x, const double *cc,
                                                          const float *w, const float *b, const float *stat, const float *gamma,
                                                          const float *beta, double count, const double *sums1, float *dw,
                                                          float *db, float *dgamma, float *dbeta, float *dw2, float *db2,
                                                          float *dgamma2, float *dbeta2) {
  __shared__ double sums[16], S[8][9];
  const int k = threadIdx.x;
  if (k < 72) {   // S[co][t]: the sum of dz[co] over the cells p with p + (t - 1) inside the plane; the plane's total is zero
    const int co = k / 9, t = k - 9 * co, dy = t / 3 - 1, dx = t % 3 - 1;
    const double *o = bsum + co * 8;
    double v = 0.0;
    if (dy == 1) v -= o[1]; else if (dy == -1) v -= o[0];
    if (dx == 1) v -= o[3]; else if (dx == -1) v -= o[2];
    if (dy != 0 && dx != 0) v += o[4 + (dy == 1 ? 2 : 0) + (dx == 1 ? 1 : 0)];
    S[co][t] = v;
  }
  __syncthreads();
  if (k < 8) {
    const int c = k;
    double all = 0.0;   // sum over ALL windows of dp[c]
    for (int co = 0; co < 8; co++)
      for (int t = 0; t < 9; t++) all += (double)wn[(t * 8 + c) * 8 + co] * S[co][t];
    const double empty = all - q[160 + c];
    const float xc = lut_x[c] + lut_x[512 * 8 + c];              // x-hat of a pixel with no bit around it, as the kernel forms it
    const bool on = fmaxf(fmaf(gamma[c], xc, beta[c]), 0.f) > 0.f;
    sums[2 * c] = q[144 + c] + (on ? empty : 0.0);
    sums[2 * c + 1] = q[152 + c] + (on ? empty * (double)xc : 0.0);
  }
  __syncthreads();
  {  // the second layer: dw2[t][ci][co], its bias gradient is exactly zero (a convolution in front of BatchNorm)
    const int co = k & 7, ci = (k >> 3) & 7, t = k >> 6;
    const double k1 = (double)fmaxf(lut_y[ci] + lut_y[512 * 8 + ci], 0.f);
    dw2[k] = (float)(k1 * S[co][t] + q[168 + k]);
    if (k < 8) { db2[k] = 0.f; dbeta2[k] = (float)sums1[2 * k]; dgamma2[k] = (float)sums1[2 * k + 1]; }
  }
  if (k < 144) {
    const int co = k % 8, u = k / 8;
    const double rs = 1.0 / sqrt((double)stat[2 * co + 1] + 1e-3), a = (double)gamma[co] * rs;
    const double m0 = sums[2 * co] / count, c = rs * (sums[2 * co + 1] / count), mean = stat[2 * co];
    const double Bu = cc[u * 18 + u];
    double T = (double)b[co] * Bu;
    for (int v = 0; v < 18; v++) T += (double)w[v * 8 + co] * cc[u * 18 + v];
    dw[k] = (float)(a * (q[k] - m0 * Bu - c * (T - mean * Bu)));
  } else if (k < 152) {
    const int co = k - 144;
    db[co] = 0.f;
    dbeta[co] = (float)sums[2 * co];
    dgamma[co] = (float)sums[2 * co + 1];
  }
}


// ---------------------------------------------------------------- the top of head 2 for the textbook targets (ofx_dqn_fit)
// One error per sample on the heat map: the loss reads o2 at ONE pixel (px, py) of a row and d o2 is non-zero there only.
// So the output convolution is evaluated at that pixel (72 taps of the up-sampled activation), its weight gradient is one
// outer product per sample, and the gradient that reaches the last head layer lives in the 4 x 4 low-res cells around the
// pointer: a few hundred operations per sample instead of four passes over 400 x 400 planes.  (The reference's dense
// targets - ofx_dqn_fit_reference - keep the dense kernels above.)
// o2p[s] = b + sum w[ky][kx][c] U_c[py + ky - 1][px + kx - 1]   (U = up2(relu(bn(z))), zero outside the plane)
__global__ void f_top_point_fwd(int n, const ofx_transition *rows, FitSrc S, const float *w, const float *b, float *o2p) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const ofx_transition r = rows[s];
  const int px = min(max(r.px, 0), 399), py = min(max(r.py, 0), 399);
  float acc = b[0];
  for (int ky = 0; ky < 3; ky++)
    for (int kx = 0; kx < 3; kx++)
      for (int c = 0; c < 8; c++)
        acc = fmaf(w[(ky * 3 + kx) * 8 + c], src_value<OFX_FIT_SRC_UP, 8>(S, (size_t)s, c, py + ky - 1, px + kx - 1, 400, 400), acc);
  o2p[s] = acc;
}
// seeds of both heads (t_loss_seed of ofx_train.hip with o2 at the pointer only): do1, d2p[s] = d loss / d o2 at the pointer
__global__ void f_top_point_seed(int n, const ofx_transition *rows, const float *o1, const float *o2p, const float *y_act,
                                 const float *y_ptr, float *do1, float *d2p, float *lpart) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const ofx_transition r = rows[s];
  const int a = r.iaction ? 1 : 0;
  const float e1 = o1[2 * s + a] - y_act[s], e2 = o2p[s] - y_ptr[s];
  do1[2 * s + a] = 2.f * e1 / (2.f * n);
  d2p[s] = 2.f * e2 / (160000.f * n);
  lpart[2 * s] = e1 * e1 / (2.f * n);
  lpart[2 * s + 1] = e2 * e2 / (160000.f * n);
}
// block = sample: pw[s][73] = d U (the output convolution's weight-gradient share, then d for the bias); the 4 x 4 x 8 patch
// of g - all of g that is not zero: gp[s][c][cell], read by f_bw<POINT> - and the sample's {sum g, sum g xhat} per channel
// in part[s][16]
__global__ __launch_bounds__(128) void f_top_point_bwd(int n, const ofx_transition *rows, FitSrc S, const float *w,
                                                       const float *d2p, const float *stat, float *gp, double *pw,
                                                       double *part) {
  __shared__ float gv_[8][16], gx_[8][16];
  const int s = blockIdx.x, tid = threadIdx.x;
  const ofx_transition r = rows[s];
  const int px = min(max(r.px, 0), 399), py = min(max(r.py, 0), 399);
  const float d = d2p[s];
  if (tid < 72) {
    const int c = tid & 7, tap = tid >> 3;
    pw[(size_t)s * 73 + tid] = (double)(d * src_value<OFX_FIT_SRC_UP, 8>(S, (size_t)s, c, py + tap / 3 - 1, px + tap % 3 - 1, 400, 400));
  } else if (tid == 72) {
    pw[(size_t)s * 73 + 72] = (double)d;
  }
  // item = (channel c, cell i of the 4 x 4 window whose first cell is (k0, m0))
  const int c = tid & 7, cell = tid >> 3, k = (((py - 1) >> 1) - 1) + (cell >> 2), m = (((px - 1) >> 1) - 1) + (cell & 3);
  float gv = 0.f, gx = 0.f;
  if (k >= 0 && k < 200 && m >= 0 && m < 200) {
    float da = 0.f;
    for (int ky = 0; ky < 3; ky++) {
      const int Y = py + ky - 1;
      if (Y < 0 || Y >= 400) continue;
      int a0, a1;
      float wt;
      fit_up_taps(Y, 200, a0, a1, wt, S.legacy);
      const float cy = (a0 == k ? 1.f - wt : 0.f) + (a1 == k ? wt : 0.f);
      if (cy == 0.f) continue;
      for (int kx = 0; kx < 3; kx++) {
        const int X = px + kx - 1;
        if (X < 0 || X >= 400) continue;
        fit_up_taps(X, 200, a0, a1, wt, S.legacy);
        const float cx = (a0 == m ? 1.f - wt : 0.f) + (a1 == m ? wt : 0.f);
        // d o2[py][px] / d U[Y][X] = w[tap of (Y, X) as seen from (py, px)] = w[ky][kx]
        da = fmaf(cy * cx, d * w[(ky * 3 + kx) * 8 + c], da);
      }
    }
    const size_t at = (((size_t)s * 8 + c) * 200 + k) * 200 + m;
    const float zv = reinterpret_cast<const float *>(S.p)[at];
    gv = bn_act(zv, S.act[2 * c], S.act[2 * c + 1]) > 0.f ? da : 0.f;
    gx = gv * ((zv - stat[2 * c]) * rsqrtf(stat[2 * c + 1] + 1e-3f));
  }
  gp[((size_t)s * 8 + c) * 16 + cell] = gv;   // cells outside the plane: 0, never looked up
  gv_[c][cell] = gv;
  gx_[c][cell] = gx;
  __syncthreads();
  if (tid < 8) {
    double a = 0.0, b2 = 0.0;
    for (int i = 0; i < 16; i++) { a += (double)gv_[tid][i]; b2 += (double)gx_[tid][i]; }
    part[(size_t)s * 16 + 2 * tid] = a;
    part[(size_t)s * 16 + 2 * tid + 1] = b2;
  }
}

// wt[tap][co][ci] = w[tap][ci][co]: the kernel of a convolution as its transposed convolution reads it
__global__ void f_transpose_w(int ci_n, int co_n, const float *w, float *wt) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 9 * ci_n * co_n) return;
  const int ci = e % ci_n, co = (e / ci_n) % co_n, tap = e / (ci_n * co_n);
  wt[e] = w[(tap * ci_n + ci) * co_n + co];
}

FitSrc dev_src(const ofx_fit_src &s) { return FitSrc{s.keep, s.p, s.act, s.h, s.w, s.legacy}; }
int grid_for(long ntiles, int cap) { return (int)(ntiles < cap ? ntiles : cap); }

}  // namespace

size_t ofx_fit_part_doubles(void) { return (size_t)OFX_FIT_MAX_BLOCKS * (9 * 8 * 8 + 8); }

int ofx_fit_conv_fwd(hipStream_t st, int n, int ci, int co, int H, int W, const ofx_fit_src &src, const float *w,
                     const float *b, float *z, double *part, int *nblocks) {
  const int TW = W >= 100 ? 100 : 50;
  if (W % TW || H % F_TR) { ofx_set_error("ofx_dqn_fit: no forward tiling for %d x %d", H, W); return OFX_ERR_STATE; }
  const long ntiles = (long)n * (H / F_TR) * (W / TW);
  const int grid = grid_for(ntiles, OFX_FIT_MAX_BLOCKS);
  *nblocks = grid;
  const FitSrc S = dev_src(src);
#define FWD(CI_, CO_, SRC_, TW_, ST_) \
  if (ci == CI_ && co == CO_ && src.kind == SRC_ && TW == TW_ && (part != nullptr) == ST_) { \
    hipLaunchKernelGGL((f_conv_fwd<CI_, CO_, SRC_, TW_, ST_>), dim3(grid), dim3(256), 0, st, n, H, W, S, w, b, z, part); \
    OFX_HIP(hipGetLastError()); return OFX_OK; }
  FWD(2, 8, OFX_FIT_SRC_BITS, 100, true)
  FWD(8, 8, OFX_FIT_SRC_POOL, 100, true)
  FWD(8, 8, OFX_FIT_SRC_PLANE, 100, true)
  FWD(8, 8, OFX_FIT_SRC_POOL, 50, true)
  FWD(1, 2, OFX_FIT_SRC_UPRAW, 50, true)
  FWD(2, 4, OFX_FIT_SRC_UP, 100, true)
  FWD(4, 8, OFX_FIT_SRC_UP, 100, true)
#undef FWD
  ofx_set_error("ofx_dqn_fit: no forward kernel for %d -> %d channels, source %d, %d x %d", ci, co, src.kind, H, W);
  return OFX_ERR_STATE;
}

int ofx_fit_finish(hipStream_t st, int nblocks, int c_n, double count, const double *part, const float *gamma,
                   const float *beta, double *sums, float *stat, float *act) {
  hipLaunchKernelGGL(f_finish, dim3(1), dim3(256), 0, st, nblocks, c_n, count, part, gamma, beta, sums, stat, act);
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}

int ofx_fit_pool_act(hipStream_t st, int n, int C, int H, int W, const float *z, const float *act, float *p) {
  const size_t total = (size_t)n * C * (H / 2) * (W / 2);
  hipLaunchKernelGGL(f_pool_act, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, n, C, H, W, z, act, p);
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}

int ofx_fit_b1_pool(hipStream_t st, int n, int H, int W, int conv, const float *dzn, const float *wn, const float *z,
                    const float *stat, const float *act, float *g, double *part, int *nblocks, float *wtr, const float *zero) {
  const int Hp = H / 2, Wp = W / 2;
  const long ntiles = (long)n * ((Hp + P_TR - 1) / P_TR) * ((Wp + P_TW - 1) / P_TW);
  const int grid = grid_for(ntiles, OFX_FIT_MAX_BLOCKS);
  *nblocks = grid;
  if (conv) {
    hipLaunchKernelGGL(f_transpose_w, dim3(3), dim3(192), 0, st, 8, 8, wn, wtr);
    hipLaunchKernelGGL(f_b1_pool<true>, dim3(grid), dim3(256), 0, st, n, H, W, dzn, (const float *)wtr, z, stat, act, g, part, zero);
  }
  else hipLaunchKernelGGL(f_b1_pool<false>, dim3(grid), dim3(256), 0, st, n, H, W, dzn, wn, z, stat, act, g, part, zero);
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}

int ofx_fit_b1_up(hipStream_t st, int n, int c, int con, int h, int w, int bn, const float *dzn, const float *wn,
                  const float *zp, const float *stat, const float *act, int legacy, float *g, double *part, int *nblocks,
                  const float *zero, float *wtr) {
  hipLaunchKernelGGL(f_transpose_w, dim3(3), dim3(192), 0, st, c, con, wn, wtr);
#define B1U(C_, CON_, TRL_, BN_) \
  if (c == C_ && con == CON_ && (bn != 0) == BN_) { \
    const long ntiles = (long)n * ((h + TRL_ - 1) / TRL_) * ((w + U_TW - 1) / U_TW); \
    const int grid = grid_for(ntiles, OFX_FIT_MAX_BLOCKS); \
    *nblocks = grid; \
    hipLaunchKernelGGL((f_b1_up<C_, CON_, TRL_, BN_>), dim3(grid), dim3(256), 0, st, n, h, w, dzn, (const float *)wtr, zp, stat, act, legacy, g, part, zero); \
    OFX_HIP(hipGetLastError()); return OFX_OK; }
  B1U(1, 2, 5, false)
  B1U(2, 4, 5, true)
  B1U(4, 8, 3, true)
  B1U(8, 1, 5, true)
#undef B1U
  ofx_set_error("ofx_dqn_fit: no up-sampling backward kernel for %d -> %d channels", c, con);
  return OFX_ERR_STATE;
}

int ofx_fit_bw(hipStream_t st, int n, int ci, int co, int H, int W, const ofx_fit_src &src, int bn, float *g,
               const float *z, const float *stat, const float *gamma, const double *sums, double *part, float *dw,
               float *db, float *dgamma, float *dbeta, const float *gpatch, const ofx_transition *rows) {
  const int TW = W >= 100 ? 100 : 50;
  if (W % TW) { ofx_set_error("ofx_dqn_fit: no weight-gradient tiling for %d x %d", H, W); return OFX_ERR_STATE; }
  int grid = 0;
  const FitSrc S = dev_src(src);
  const double count = (double)n * H * W;
  bool done = false;
  // NT threads and TR tile rows per workgroup (OFX_FIT_BW_NT / _TR for the layers with CI <= 4: 256-thread workgroups on
  // 4-row tiles - four independent load / compute pipelines per CU instead of two - measured 4 % SLOWER, tools/ab_fit.sh)
#define BWK(CI_, CO_, SRC_, TW_, BN_, NT_, TR_) \
  if (!done && ci == CI_ && co == CO_ && src.kind == SRC_ && TW == TW_ && (bn != 0) == BN_) { \
    grid = grid_for((long)n * ((H + TR_ - 1) / TR_) * (W / TW), OFX_FIT_MAX_BLOCKS / 2); \
    hipLaunchKernelGGL((f_bw<CI_, CO_, SRC_, TW_, BN_, NT_, TR_>), dim3(grid), dim3(NT_), 0, st, n, H, W, S, g, z, stat, gamma, sums, count, part, \
                       (const float *)nullptr, (const ofx_transition *)nullptr); \
    done = true; }
  if (gpatch) {   // g as one 4 x 4 patch per sample and channel (ofx_fit_top_point)
    if (!(ci == 4 && co == 8 && src.kind == OFX_FIT_SRC_UP && TW == 100 && bn && rows)) {
      ofx_set_error("ofx_dqn_fit: the patch form of g is built for the last head layer only");
      return OFX_ERR_STATE;
    }
    grid = grid_for((long)n * ((H + OFX_FIT_BW_TR - 1) / OFX_FIT_BW_TR) * (W / TW), OFX_FIT_MAX_BLOCKS / 2);
    hipLaunchKernelGGL((f_bw<4, 8, OFX_FIT_SRC_UP, 100, true, OFX_FIT_BW_NT, OFX_FIT_BW_TR, false, true>), dim3(grid),
                       dim3(OFX_FIT_BW_NT), 0, st, n, H, W, S, g, z, stat, gamma, sums, count, part, gpatch, rows);
    done = true;
  }
  BWK(2, 8, OFX_FIT_SRC_BITS, 100, true, OFX_FIT_BW_NT, OFX_FIT_BW_TR)
  BWK(8, 8, OFX_FIT_SRC_POOL, 100, true, 512, 8)
  BWK(8, 8, OFX_FIT_SRC_POOL, 50, true, 512, 8)
  BWK(8, 8, OFX_FIT_SRC_PLANE, 100, true, 512, 8)
  BWK(8, 8, OFX_FIT_SRC_PLANE, 50, true, 512, 8)
  BWK(4, 8, OFX_FIT_SRC_UP, 100, true, OFX_FIT_BW_NT, OFX_FIT_BW_TR)
#undef BWK
#define BWS(CI_, CO_, SRC_, TW_, BN_) \
  if (!done && ci == CI_ && co == CO_ && src.kind == SRC_ && TW == TW_ && (bn != 0) == BN_) { \
    grid = grid_for((long)n * ((H + W_TR - 1) / W_TR) * (W / TW), OFX_FIT_MAX_BLOCKS / 2); \
    hipLaunchKernelGGL((f_bw_small<CI_, CO_, SRC_, TW_, BN_>), dim3(grid), dim3(256), 0, st, n, H, W, S, g, z, stat, gamma, sums, count, part); \
    done = true; }
  BWS(1, 2, OFX_FIT_SRC_UPRAW, 50, true)
  BWS(2, 4, OFX_FIT_SRC_UP, 100, true)
#undef BWS
  if (!done) {
    ofx_set_error("ofx_dqn_fit: no weight-gradient kernel for %d -> %d channels, source %d", ci, co, src.kind);
    return OFX_ERR_STATE;
  }
  OFX_HIP(hipGetLastError());
  const int nw = 9 * ci * co, nv = nw + co;
  hipLaunchKernelGGL(f_bw_finish, dim3((nv + 15) / 16), dim3(256), 0, st, nv, nw, grid, part, dw, db, bn ? co : 0,
                     bn ? sums : nullptr, dgamma, dbeta);
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}

size_t ofx_fit_out_floats(void) { return 292; }
size_t ofx_fit_out_doubles(int n) { return (size_t)n * 72 + 292 + 72; }

// o2[n][400][400] = conv3x3(up2(relu(bn(z)))) + b with z [n][8][200][200]: phase form + frame correction
int ofx_fit_out_fwd(hipStream_t st, int n, const ofx_fit_src &src, const float *w, const float *b, float *o2, float *weff) {
  if (src.kind != OFX_FIT_SRC_UP || src.h != 200 || src.w != 200) { ofx_set_error("ofx_dqn_fit: output convolution expects the 200 x 200 head layer"); return OFX_ERR_STATE; }
  hipLaunchKernelGGL(f_out_prep, dim3(2), dim3(256), 0, st, w, b, src.legacy, weff);
  const FitSrc S = dev_src(src);
  FitSrc A = S;                                    // the same planes read at their own resolution
  const long ntiles = (long)n * (200 / F_TR) * (200 / 100);
  const int grid = grid_for(ntiles, OFX_FIT_MAX_BLOCKS);
  hipLaunchKernelGGL((f_conv_fwd<8, 4, OFX_FIT_SRC_ACTREP, 100, false, true>), dim3(grid), dim3(256), 0, st, n, 200, 200, A,
                     weff, weff + 288, o2, (double *)nullptr);
  const size_t fr = (size_t)n * kFrame;
  hipLaunchKernelGGL(f_out_frame_fwd, dim3((unsigned)((fr + 255) / 256)), dim3(256), 0, st, n, S, w, o2);
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}

// dw[3][3][8][1], db[1] of the output convolution from d2 = d loss / d o2
int ofx_fit_out_bw(hipStream_t st, int n, const ofx_fit_src &src, const float *d2, double *part, double *fpart, float *dw,
                   float *db) {
  const FitSrc S = dev_src(src);
  const int grid = grid_for((long)n * ((200 + W_TR - 1) / W_TR) * 2, OFX_FIT_MAX_BLOCKS / 2);
  hipLaunchKernelGGL((f_bw<8, 4, OFX_FIT_SRC_ACTREP, 100, false, 512, 8, true>), dim3(grid), dim3(512), 0, st, n, 200, 200, S,
                     const_cast<float *>(d2), (const float *)nullptr, (const float *)nullptr, (const float *)nullptr,
                     (const double *)nullptr, 1.0, part, (const float *)nullptr, (const ofx_transition *)nullptr);
  hipLaunchKernelGGL(f_out_frame_bw, dim3(n), dim3(256), 0, st, n, S, d2, fpart);
  double *q = fpart + (size_t)n * 72, *fr = q + 292;      // behind the per-sample frame terms
  hipLaunchKernelGGL(f_sum_rows, dim3((292 + 15) / 16), dim3(256), 0, st, 292, grid, part, q);
  hipLaunchKernelGGL(f_sum_rows, dim3((72 + 15) / 16), dim3(256), 0, st, 72, n, fpart, fr);
  hipLaunchKernelGGL(f_out_bw_finish, dim3(1), dim3(128), 0, st, q, fr, src.legacy, dw, db);
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}

size_t ofx_fit_first_doubles(int n) { return (size_t)1024 * 324 + 324 + FS_NV + 64 + (size_t)n * 64; }   // correlation rows, cc, q, border sums + their per-sample rows
size_t ofx_fit_first_floats(void) { return 2 * 8192; }
// The first trunk layer's forward without its tensor: autocorrelation of the bit maps (cpart, kept for the backward) ->
// batch statistics (stat, act) and the two tables (luts: ofx_fit_first_floats() floats) -> p0 = pool(relu(bn(conv1)))
// [n][8][200][200] through the forward's table kernel
int ofx_fit_first_fwd(ofx_handle *h, int n, const void *bits, const float *w, const float *b, const float *gamma,
                      const float *beta, double *cpart, float *stat, float *act, float *luts, float *p0) {
  hipStream_t st = h->stream;
  const int nchunks = n * (400 / C0_ROWS), nc = nchunks < 1024 ? nchunks : 1024;   // persistent: <= 1024 rows of partial counts
  double *cc = cpart + (size_t)1024 * 324;
  hipLaunchKernelGGL(f_bits_corr, dim3(nc), dim3(384), 0, st, n, (const uint32_t *)bits, cpart);
  hipLaunchKernelGGL(f_sum_rows, dim3((324 + 15) / 16), dim3(256), 0, st, 324, nc, cpart, cc);
  hipLaunchKernelGGL(f_first_prepare, dim3(1), dim3(256), 0, st, cc, w, b, gamma, beta, (double)n * 160000.0, stat, act, luts,
                     luts + 8192);
  OFX_HIP(hipGetLastError());
  return ofx_launch_conv1_lut(h, bits, n, luts, p0);
}
// The backward of the first two trunk layers from g1 = d loss / d (BatchNorm output of the second layer) [n][8][200][200]
// (ofx_fit_b1_pool) with sums1 = its {sum g, sum g xhat} per channel, z1 / stat1 / gamma1 of that layer, wn its kernel, p0 its
// input (the pooled activation ofx_fit_first_fwd left), luts / cpart as left by ofx_fit_first_fwd: dw2 .. dbeta2 of the second
// layer, dw .. dbeta of the first (w, b, stat, gamma, beta: its own tensors).  The second layer's dz is never stored.
// part: n x 744 doubles
int ofx_fit_first_bwd(hipStream_t st, int n, const void *bits, const float *g1, const float *z1, const float *stat1,
                      const float *gamma1, const double *sums1, const float *wn, const float *p0, const float *luts,
                      const float *w, const float *b, const float *stat, const float *gamma, const float *beta, double *part,
                      double *cpart, float *dw, float *db, float *dgamma, float *dbeta, float *dw2, float *db2, float *dgamma2,
                      float *dbeta2) {
  double *cc = cpart + (size_t)1024 * 324, *q = cc + 324, *bsum = q + FS_NV, *bpart = bsum + 64;
  hipLaunchKernelGGL(f_first_border, dim3(n), dim3(256), 0, st, n, g1, z1, stat1, gamma1, sums1, bpart);
  hipLaunchKernelGGL(f_sum_rows, dim3((64 + 15) / 16), dim3(256), 0, st, 64, n, bpart, bsum);
  hipLaunchKernelGGL(f_first_bwd, dim3(n), dim3(256), 0, st, n, (const uint32_t *)bits, g1, z1, stat1, gamma1, sums1, wn, p0, luts,
                     luts + 8192, gamma, beta, part);
  hipLaunchKernelGGL(f_sum_rows, dim3((FS_NV + 15) / 16), dim3(256), 0, st, FS_NV, n, part, q);
  hipLaunchKernelGGL(f_first_bwd_finish, dim3(1), dim3(576), 0, st, q, bsum, wn, luts, luts + 8192, cc, w, b, stat, gamma, beta,
                     (double)n * 160000.0, sums1, dw, db, dgamma, dbeta, dw2, db2, dgamma2, dbeta2);
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}
size_t ofx_fit_first_part_doubles(int n) { return (size_t)n * FS_NV; }

size_t ofx_fit_point_doubles(int n) { return (size_t)n * (73 + 16); }
// The top of head 2 for one error per sample (ofx_dqn_fit): o2 at the pointer (o2p [n]), the seeds of both heads (do1, d2p [n],
// lpart [2 n]), the output convolution's dw / db, g of the last head layer as its 4 x 4 patch per sample and channel (gpatch
// [n][8][16], for ofx_fit_bw's gpatch argument; + its BatchNorm sums in `sums`); scratch: ofx_fit_point_doubles(n) doubles
int ofx_fit_top_point(hipStream_t st, int n, const ofx_transition *rows, const ofx_fit_src &src, const float *w, const float *b,
                      const float *o1, const float *y_act, const float *y_ptr, const float *stat, float *o2p, float *do1,
                      float *d2p, float *lpart, float *gpatch, double *scratch, double *sums, float *dw, float *db) {
  const FitSrc S = dev_src(src);
  double *pw = scratch, *part = scratch + (size_t)n * 73;
  hipLaunchKernelGGL(f_top_point_fwd, dim3((n + 63) / 64), dim3(64), 0, st, n, rows, S, w, b, o2p);
  hipLaunchKernelGGL(f_top_point_seed, dim3((n + 255) / 256), dim3(256), 0, st, n, rows, o1, o2p, y_act, y_ptr, do1, d2p, lpart);
  hipLaunchKernelGGL(f_top_point_bwd, dim3(n), dim3(128), 0, st, n, rows, S, w, d2p, stat, gpatch, pw, part);
  hipLaunchKernelGGL(f_bw_finish, dim3((73 + 15) / 16), dim3(256), 0, st, 73, 72, n, pw, dw, db, 0, (const double *)nullptr,
                     (float *)nullptr, (float *)nullptr);
  hipLaunchKernelGGL(f_finish, dim3(1), dim3(256), 0, st, n, 8, 1.0, part, (const float *)nullptr, (const float *)nullptr, sums,
                     (float *)nullptr, (float *)nullptr);
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}
