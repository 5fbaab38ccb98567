// ofx_train.hip - one DQN fit step of the bi-head pointer_model: the model.fit call of
// Trainer.replay (agents/qlearnIA_V2.py:284; model.compile(loss='mse', optimizer=Adam(lr)) :190).
//
// Two forms of the same step.  The default is the LEAN form (dqn_fit_lean below + the kernels of ofx_fit.hip): only the
// pre-activation tensor of every convolution is kept in HBM, the rest is recomputed inside fused tiles (9.4 MB of
// workspace per minibatch row; 4096 rows in 42 ms, 65 ms with the reference's dense targets).  The PLAIN form (OFX_OPT_FIT_PLAIN, dqn_fit_impl) is the layer-by-layer
// original - one fp32 VALU kernel per layer and pass, every tensor of the graph in HBM (61 MB per row) - kept as the
// reference of the lean form.  Both: fixed-order reductions (no atomics: a fit is reproducible to the bit), the dense
// forwards on the f32 MFMA GEMM of the forward, every gradient tensor checked against torch autograd in float64 and the
// two forms against each other (tests/test_train.py).  Keras semantics assumed (parity unpinned: no keras in the image):
//   * fit runs the graph in training mode: BatchNorm normalises with the batch mean / biased variance (eps 1e-3) and
//     moves the stored statistics: moving = 0.99 moving + 0.01 batch;
//   * loss = mse(output1) + mse(output2), each the mean over the batch and the output elements;
//   * the targets equal the current predictions except target[iaction] = y_act and ptr_target[pointer] = y_ptr
//     (qlearnIA_V2.py:279-280), i.e. one non-zero error per head and sample.  The pointer addresses heat[y][x] and
//     the inputs are the transitions' `state` observations (the reference's [x][y] indexing and its use of
//     next_state as input, :280-283, are stated in include/ofx.h, not reproduced);
//   * Adam: beta1 0.9, beta2 0.999, eps 1e-7, bias-corrected step size.
#include "ofx_internal.h"
#include "ofx_fit.h"
#include <string.h>

#define TPS 400

// ---------------------------------------------------------------- elementwise / layout kernels
__global__ void t_bits_to_f32(int n, const uint32_t *bits, float *x) {  // bits [n][2][5000] -> x [n][2][400][400]
  const size_t total = (size_t)n * 2 * TPS * TPS;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const size_t img = e / (TPS * TPS), p = e - img * (TPS * TPS);
    x[e] = (float)((bits[img * ((TPS * TPS) >> 5) + (p >> 5)] >> (p & 31)) & 1u);
  }
}

// z[n][co][H][W] = conv3x3(x[n][ci][H][W], w HWIO [3][3][ci][co]) + b[co], zero padding.  One thread per PIXEL with all
// CO outputs in registers: an input value is loaded once for the CO channels (the weights are wave-uniform reads); per
// output the terms are added in the order (ci, ky, kx), one rounding per operation (-ffp-contract=off).
template <int CO>
__global__ void t_conv_fwd(int n, int ci_n, int H, int W, const float *x, const float *w, const float *b, float *z) {
  const size_t per = (size_t)H * W, total = (size_t)n * per;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int xx = e % W, yy = (e / W) % H;
    const size_t s = e / per;
    float acc[CO];
#pragma unroll
    for (int co = 0; co < CO; co++) acc[co] = b[co];
    for (int ci = 0; ci < ci_n; ci++) {
      const float *xp = x + (s * ci_n + ci) * per;
#pragma unroll
      for (int ky = 0; ky < 3; ky++) {
        const int y = yy + ky - 1;
        if (y < 0 || y >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
          const int xq = xx + kx - 1;
          if (xq < 0 || xq >= W) continue;
          const float v = xp[(size_t)y * W + xq];
          const float *wp = w + ((ky * 3 + kx) * ci_n + ci) * CO;
#pragma unroll
          for (int co = 0; co < CO; co++) acc[co] += v * wp[co];
        }
      }
    }
#pragma unroll
    for (int co = 0; co < CO; co++) z[(s * CO + co) * per + (size_t)yy * W + xx] = acc[co];
  }
}

// dx[n][ci][H][W] = sum_co conv3x3_transposed(dz[n][co], w): one thread per pixel with all CI inputs in registers
template <int CI>
__global__ void t_conv_bwd_data(int n, int co_n, int H, int W, const float *dz, const float *w, float *dx) {
  const size_t per = (size_t)H * W, total = (size_t)n * per;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int xx = e % W, yy = (e / W) % H;
    const size_t s = e / per;
    float acc[CI];
#pragma unroll
    for (int ci = 0; ci < CI; ci++) acc[ci] = 0.f;
    for (int co = 0; co < co_n; co++) {
      const float *dp = dz + (s * co_n + co) * per;
#pragma unroll
      for (int ky = 0; ky < 3; ky++) {
        const int y = yy - (ky - 1);  // output pixel that read this input through tap ky
        if (y < 0 || y >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
          const int xq = xx - (kx - 1);
          if (xq < 0 || xq >= W) continue;
          const float v = dp[(size_t)y * W + xq];
          const float *wp = w + (size_t)(ky * 3 + kx) * CI * co_n + co;
#pragma unroll
          for (int ci = 0; ci < CI; ci++) acc[ci] += v * wp[ci * co_n];
        }
      }
    }
#pragma unroll
    for (int ci = 0; ci < CI; ci++) dx[(s * CI + ci) * per + (size_t)yy * W + xx] = acc[ci];
  }
}

// dw[(ky,kx,ci,co)] = sum_{n,y,x} x[n][ci][y+ky-1][x+kx-1] dz[n][co][y][x] (double accumulators: up to 10^7 terms).
// block = a block of CIB input channels x a block of COB output channels x one of gridDim.y slices of the (n, H, W) sum:
// a pixel's input taps and output gradients are loaded once for the CIB x COB weight columns; db[co] = sum dz comes from
// the blocks of input-channel block 0.  part[slice][wid], combined in slice order by t_conv_bwd_finish.
template <int CIB, int COB>
__global__ __launch_bounds__(256) void t_conv_bwd_weight(int n, int ci_n, int co_n, int H, int W, const float *x,
                                                         const float *dz, double *part) {
  constexpr int NV = CIB * COB * 9 + COB;
  __shared__ double red[4][NV];
  const int cob = co_n / COB;                                      // co_n is a multiple of COB, ci_n of CIB
  const int co0 = (blockIdx.x % cob) * COB, ci0 = (blockIdx.x / cob) * CIB, nw = 9 * ci_n * co_n;
  const size_t per = (size_t)H * W, total = (size_t)n * per, stride = (size_t)gridDim.y * 256;
  double acc[NV];
#pragma unroll
  for (int k = 0; k < NV; k++) acc[k] = 0.0;
  for (size_t e = (size_t)blockIdx.y * 256 + threadIdx.x; e < total; e += stride) {
    const int s = e / per, yy = (e - (size_t)s * per) / W, xx = e % W;
    double g[COB];
#pragma unroll
    for (int c = 0; c < COB; c++) {
      g[c] = dz[((size_t)s * co_n + co0 + c) * per + (size_t)yy * W + xx];
      acc[CIB * COB * 9 + c] += g[c];
    }
#pragma unroll
    for (int i = 0; i < CIB; i++) {
      const float *xs = x + ((size_t)s * ci_n + ci0 + i) * per;
#pragma unroll
      for (int ky = 0; ky < 3; ky++) {
        const int y = yy + ky - 1;
        if (y < 0 || y >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
          const int xq = xx + kx - 1;
          if (xq < 0 || xq >= W) continue;
          const double v = (double)xs[(size_t)y * W + xq];
#pragma unroll
          for (int c = 0; c < COB; c++) acc[(i * COB + c) * 9 + ky * 3 + kx] += v * g[c];
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NV; k++) {
    double v = acc[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    const int k = threadIdx.x;
    const double v = red[0][k] + red[1][k] + red[2][k] + red[3][k];
    double *row = part + (size_t)blockIdx.y * (nw + co_n);
    if (k < CIB * COB * 9) {
      const int t = k % 9, ic = k / 9, i = ic / COB, c = ic % COB;
      row[(t * ci_n + ci0 + i) * co_n + co0 + c] = v;
    } else if (ci0 == 0) {
      row[nw + co0 + (k - CIB * COB * 9)] = v;
    }
  }
}

__global__ void t_conv_bwd_finish(int nw, int nb, int slices, const double *part, float *dw, float *db) {
  const int wid = blockIdx.x * blockDim.x + threadIdx.x;
  if (wid >= nw + nb) return;
  double acc = 0.0;
  for (int k = 0; k < slices; k++) acc += part[(size_t)k * (nw + nb) + wid];
  if (wid < nw) dw[wid] = (float)acc;
  else db[wid - nw] = (float)acc;
}

// per-channel sums over (n, H, W): out[c] = {sum a, sum a*b} (b may be null -> sum a*a); double accumulation,
// gridDim.y slices per channel written to part[c][slice][2] and combined IN SLICE ORDER by t_chan_sums_finish: the same
// bits every run (the fit is reproducible)
__global__ __launch_bounds__(256) void t_chan_sums(int n, int c_n, size_t per, const float *a, const float *b, double *part) {
  __shared__ double r0[256], r1[256];
  const int c = blockIdx.x;
  double s0 = 0.0, s1 = 0.0;
  for (int s = 0; s < n; s++) {                                    // the channel's plane of every sample, slice by slice
    const float *pa = a + ((size_t)s * c_n + c) * per, *pb = b ? b + ((size_t)s * c_n + c) * per : pa;
    for (size_t i = (size_t)blockIdx.y * 256 + threadIdx.x; i < per; i += (size_t)gridDim.y * 256) {
      const double va = pa[i], vb = pb[i];
      s0 += va;
      s1 += va * vb;
    }
  }
  r0[threadIdx.x] = s0; r1[threadIdx.x] = s1;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { r0[threadIdx.x] += r0[threadIdx.x + o]; r1[threadIdx.x] += r1[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[((size_t)c * gridDim.y + blockIdx.y) * 2] = r0[0]; part[((size_t)c * gridDim.y + blockIdx.y) * 2 + 1] = r1[0]; }
}
__global__ void t_chan_sums_finish(int c_n, int slices, const double *part, double *out) {
  const int c = threadIdx.x;
  if (c >= c_n) return;
  double s0 = 0.0, s1 = 0.0;
  for (int k = 0; k < slices; k++) { s0 += part[((size_t)c * slices + k) * 2]; s1 += part[((size_t)c * slices + k) * 2 + 1]; }
  out[2 * c] = s0; out[2 * c + 1] = s1;
}

// batch statistics from the sums: mean, biased variance (stat[c] = {mean, var})
__global__ void t_bn_finish_stats(int c_n, double count, const double *sums, float *stat) {
  const int c = threadIdx.x;
  if (c >= c_n) return;
  const double m = sums[2 * c] / count, v = sums[2 * c + 1] / count - m * m;
  stat[2 * c] = (float)m;
  stat[2 * c + 1] = (float)(v > 0.0 ? v : 0.0);
}

// The three BatchNorm element-wise kernels: blockIdx.y = plane (sample s, channel c) of `per` elements - the channel is
// block-uniform, no division per element.
// a = relu(gamma * (z - mean) / sqrt(var + eps) + beta)
__global__ void t_bn_relu_fwd(int c_n, size_t per, const float *z, const float *stat, const float *gamma, const float *beta,
                              float *a) {
  const int c = blockIdx.y % c_n;
  const size_t base = (size_t)blockIdx.y * per;
  const float mean = stat[2 * c], rs = rsqrtf(stat[2 * c + 1] + 1e-3f), g = gamma[c], bt = beta[c];
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x) {
    const float xh = (z[base + i] - mean) * rs;
    a[base + i] = fmaxf(g * xh + bt, 0.f);
  }
}

// dy (w.r.t. the BN output, ReLU mask applied) and xhat, in place over da / into xh
__global__ void t_bn_relu_bwd_pre(int c_n, size_t per, const float *z, const float *a, const float *stat, float *da, float *xh) {
  const int c = blockIdx.y % c_n;
  const size_t base = (size_t)blockIdx.y * per;
  const float mean = stat[2 * c], rs = rsqrtf(stat[2 * c + 1] + 1e-3f);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x) {
    xh[base + i] = (z[base + i] - mean) * rs;
    if (!(a[base + i] > 0.f)) da[base + i] = 0.f;
  }
}

// dz = gamma / sqrt(var+eps) * (dy - mean(dy) - xhat * mean(dy * xhat)) ; sums[c] = {sum dy, sum dy*xhat}
__global__ void t_bn_bwd(int c_n, size_t per, double count, const float *dy, const float *xh, const float *stat,
                         const float *gamma, const double *sums, float *dz, float *dgamma, float *dbeta) {
  const int c = blockIdx.y % c_n;
  const size_t base = (size_t)blockIdx.y * per;
  const float m0 = (float)(sums[2 * c] / count), m1 = (float)(sums[2 * c + 1] / count);
  const float k = gamma[c] * rsqrtf(stat[2 * c + 1] + 1e-3f);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x)
    dz[base + i] = k * (dy[base + i] - m0 - xh[base + i] * m1);
  if (blockIdx.x == 0 && blockIdx.y < (unsigned)c_n && threadIdx.x == 0) {
    dbeta[blockIdx.y] = (float)sums[2 * blockIdx.y];
    dgamma[blockIdx.y] = (float)sums[2 * blockIdx.y + 1];
  }
}

__global__ void t_pool_fwd(int nc, int H, int W, const float *a, float *p) {  // [nc][H][W] -> [nc][H/2][W/2]
  const int H2 = H / 2, W2 = W / 2;
  const size_t total = (size_t)nc * H2 * W2;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int x = e % W2, y = (e / W2) % H2;
    const size_t c = e / ((size_t)W2 * H2);
    const float *q = a + (c * H + 2 * y) * W + 2 * x;
    p[e] = fmaxf(fmaxf(q[0], q[1]), fmaxf(q[W], q[W + 1]));
  }
}

// routes dp to the first maximum of each window (row-major), zero elsewhere
__global__ void t_pool_bwd(int nc, int H, int W, const float *a, const float *dp, float *da) {
  const int H2 = H / 2, W2 = W / 2;
  const size_t total = (size_t)nc * H2 * W2;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int x = e % W2, y = (e / W2) % H2;
    const size_t c = e / ((size_t)W2 * H2), base = (c * H + 2 * y) * W + 2 * x;
    const float v[4] = {a[base], a[base + 1], a[base + W], a[base + W + 1]};
    int k = 0;
    for (int i = 1; i < 4; i++) if (v[i] > v[k]) k = i;
    const size_t off[4] = {base, base + 1, base + W, base + W + 1};
    for (int i = 0; i < 4; i++) da[off[i]] = i == k ? dp[e] : 0.f;
  }
}

// x2 bilinear, edge clamp: u[nc][2H][2W]; half-pixel centres, or (legacy, OFX_OPT_BILINEAR_LEGACY) src = dst / 2
__device__ inline void up_taps(int u, int n, int &i0, int &i1, float &w1, int legacy) {
  const int k = u >> 1;
  if (legacy) { i0 = k; i1 = min(k + 1, n - 1); w1 = (u & 1) ? 0.5f : 0.f; }
  else if (u & 1) { i0 = k; i1 = min(k + 1, n - 1); w1 = 0.25f; }
  else { i0 = max(k - 1, 0); i1 = k; w1 = 0.75f; }
}
__global__ void t_up_fwd(int nc, int H, int W, const float *x, float *u, int legacy) {
  const int H2 = 2 * H, W2 = 2 * W;
  const size_t total = (size_t)nc * H2 * W2;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int ux = e % W2, uy = (e / W2) % H2;
    const size_t c = e / ((size_t)W2 * H2);
    int y0, y1, x0, x1; float wy, wx;
    up_taps(uy, H, y0, y1, wy, legacy); up_taps(ux, W, x0, x1, wx, legacy);
    const float *q = x + c * H * W;
    const float top = q[(size_t)y0 * W + x0] * (1.f - wx) + q[(size_t)y0 * W + x1] * wx;
    const float bot = q[(size_t)y1 * W + x0] * (1.f - wx) + q[(size_t)y1 * W + x1] * wx;
    u[e] = top * (1.f - wy) + bot * wy;
  }
}
// gather form: dx[y][x] = sum over the up-res cells whose two taps per axis include (y, x), in a fixed order (every run
// gives the same bits; the scatter form needed float atomics).  A source row y is a tap of up-res rows 2y-2 .. 2y+2 only.
__global__ void t_up_bwd(int nc, int H, int W, const float *du, float *dx, int legacy) {
  const int H2 = 2 * H, W2 = 2 * W;
  const size_t total = (size_t)nc * H * W;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int x = e % W, y = (e / W) % H;
    const size_t c = e / ((size_t)W * H);
    const float *q = du + c * H2 * W2;
    float cy[5], cx[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
      const int uy = 2 * y - 2 + k, ux = 2 * x - 2 + k;
      int a0, a1; float w;
      cy[k] = 0.f; cx[k] = 0.f;
      if (uy >= 0 && uy < H2) { up_taps(uy, H, a0, a1, w, legacy); cy[k] = (a0 == y ? 1.f - w : 0.f) + (a1 == y ? w : 0.f); }
      if (ux >= 0 && ux < W2) { up_taps(ux, W, a0, a1, w, legacy); cx[k] = (a0 == x ? 1.f - w : 0.f) + (a1 == x ? w : 0.f); }
    }
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 5; ky++) {
      if (cy[ky] == 0.f) continue;
      const int uy = 2 * y - 2 + ky;
      float row = 0.f;
#pragma unroll
      for (int kx = 0; kx < 5; kx++)
        if (cx[kx] != 0.f) row += cx[kx] * q[(size_t)uy * W2 + 2 * x - 2 + kx];
      acc += cy[ky] * row;
    }
    dx[e] = acc;
  }
}

// dy masked by the ReLU of y (in place) when relu
__global__ void t_relu_mask(size_t total, const float *y, float *dy) {
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x)
    if (!(y[e] > 0.f)) dy[e] = 0.f;
}
// dw[k][o] = sum_s x[s][k] dy[s][o], db[o] = sum_s dy[s][o] (row k = in_n).  A thread owns 4 rows x 4 columns (8 loads per
// 16 FMAs instead of 2 per FMA) of one of 8 interleaved shares of the samples; the 8 shares of a tile are added in share
// order through LDS: the same bits every run.  Block = 32 tiles x 8 shares.
__global__ __launch_bounds__(256) void t_dense_bwd_w(int n, int in_n, int out_n, const float *x, const float *dy, float *dw, float *db) {
  __shared__ float red[8][32][17];
  const int oq = (out_n + 3) / 4, kq = (in_n + 1 + 3) / 4;
  const int tl = threadIdx.x & 31, share = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + tl;
  const bool live = e < oq * kq;
  const int o0 = live ? (e % oq) * 4 : 0, k0 = live ? (e / oq) * 4 : 0;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = 0.f;
  if (live)
    for (int s = share; s < n; s += 8) {
      float xv[4], dv[4];
#pragma unroll
      for (int i = 0; i < 4; i++) xv[i] = k0 + i < in_n ? x[(size_t)s * in_n + k0 + i] : (k0 + i == in_n ? 1.f : 0.f);
#pragma unroll
      for (int j = 0; j < 4; j++) dv[j] = o0 + j < out_n ? dy[(size_t)s * out_n + o0 + j] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] += xv[i] * dv[j];
    }
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) red[share][tl][4 * i + j] = acc[i][j];
  __syncthreads();
  if (share != 0 || !live) return;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 8; q++) t += red[q][tl][4 * i + j];
      if (o0 + j >= out_n) continue;
      if (k0 + i < in_n) dw[(size_t)(k0 + i) * out_n + o0 + j] = t;
      else if (k0 + i == in_n) db[o0 + j] = t;
    }
}
// dx[s][k] = sum_o dy[s][o] w[k][o] (+ dx when accumulate): 4 samples x 4 inputs per thread
__global__ __launch_bounds__(256) void t_dense_bwd_x(int n, int in_n, int out_n, const float *dy, const float *w, float *dx, int accumulate) {
  const int kq = (in_n + 3) / 4, sq = (n + 3) / 4;
  const size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (e >= (size_t)kq * sq) return;
  const int k0 = (int)(e % kq) * 4, s0 = (int)(e / kq) * 4;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = 0.f;
  for (int o = 0; o < out_n; o++) {
    float dv[4], wv[4];
#pragma unroll
    for (int i = 0; i < 4; i++) dv[i] = s0 + i < n ? dy[(size_t)(s0 + i) * out_n + o] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) wv[j] = k0 + j < in_n ? w[(size_t)(k0 + j) * out_n + o] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] += dv[i] * wv[j];
  }
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (s0 + i >= n || k0 + j >= in_n) continue;
      const size_t at = (size_t)(s0 + i) * in_n + k0 + j;
      dx[at] = accumulate ? dx[at] + acc[i][j] : acc[i][j];
    }
}

// out[c][r] = in[r][c] (a weight matrix or a batch of activations for the MFMA GEMM's row-major operands)
__global__ void t_transpose(int rows, int cols, const float *in, float *out) {
  __shared__ float tile[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8)
    if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = in[(size_t)(r0 + i) * cols + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < cols && r0 + tx < rows) out[(size_t)(c0 + i) * rows + r0 + tx] = tile[tx][i];
}

// f[n][5008] = concat(vec8, flatten_hwc(x4 [n][8][25][25])) and its transpose for the gradient
__global__ void t_concat_fwd(int n, const ofx_transition *rows, const float *x4, float *f, int next_head) {
  const size_t total = (size_t)n * 5008;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int k = e % 5008, s = e / 5008;
    if (k < 8) f[e] = next_head ? rows[s].head_next[k] : rows[s].head_prev[k];
    else { const int j = k - 8, c = j % 8, p = j / 8; f[e] = x4[((size_t)s * 8 + c) * 625 + p]; }
  }
}
__global__ void t_concat_bwd(int n, const float *df, float *dx4) {
  const size_t total = (size_t)n * 5000;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int j = e % 5000, s = e / 5000, c = j % 8, p = j / 8;
    dx4[((size_t)s * 8 + c) * 625 + p] = df[(size_t)s * 5008 + 8 + j];
  }
}

// loss seeds: one non-zero error per head and sample; loss[0] += mse(out1) share, loss[1] += mse(out2) share
__global__ void t_loss_seed(int n, const ofx_transition *rows, const float *o1, const float *o2, const float *y_act,
                            const float *y_ptr, float *do1, float *do2, float *lpart) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const ofx_transition r = rows[s];
  if (r.ship < 0) { lpart[2 * s] = lpart[2 * s + 1] = 0.f; return; }  // padding row: no error
  const int a = r.iaction ? 1 : 0, px = min(max(r.px, 0), TPS - 1), py = min(max(r.py, 0), TPS - 1);
  const float e1 = o1[2 * s + a] - y_act[s];
  const size_t k = (size_t)s * TPS * TPS + (size_t)py * TPS + px;
  const float e2 = o2[k] - y_ptr[s];
  do1[2 * s + a] = 2.f * e1 / (2.f * n);
  do2[k] = 2.f * e2 / ((float)(TPS * TPS) * n);
  lpart[2 * s] = e1 * e1 / (2.f * n);          // summed in sample order by t_sum_ordered
  lpart[2 * s + 1] = e2 * e2 / ((float)(TPS * TPS) * n);
}
// out[j] = sum_i part[i * stride + j] for j < stride, in index order (one thread per j: tiny)
__global__ void t_sum_ordered(int count, int stride, const float *part, float *out) {
  const int j = threadIdx.x;
  if (j >= stride) return;
  double acc = 0.0;
  for (int i = 0; i < count; i++) acc += (double)part[(size_t)i * stride + j];
  out[j] = (float)acc;
}

// Trainer.replay as written (ofx_dqn_fit_reference): the targets are whole predictions of `state` with one entry
// replaced per head, so every output carries an error.  t1 [n][2], t2 [n][TPS][TPS].
__global__ void t_reference_targets(int n, const ofx_transition *rows, float gamma, const float *act_next,
                                    const float *max_next, float *t1, float *t2) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const ofx_transition r = rows[s];
  const float live = r.done ? 0.f : 1.f;                                     // int(not done)
  const int px = min(max(r.px, 0), TPS - 1), py = min(max(r.py, 0), TPS - 1);
  t1[2 * s + (r.iaction ? 1 : 0)] = (float)r.reward + gamma * fmaxf(act_next[2 * s], act_next[2 * s + 1]) * live;  // :279
  // ptr_target[ipointer] with ipointer = (x, y) on the (400, 400, 1) prediction = [row][col][0]: row x, column y (:280)
  t2[(size_t)s * TPS * TPS + (size_t)px * TPS + py] = (float)r.reward + gamma * max_next[s] * live;
}
__global__ void t_unpack_heads(int n, const ofx_transition *rows, float *vec_prev, float *vec_next) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  for (int k = 0; k < 8; k++) {
    vec_prev[(size_t)s * 8 + k] = rows[s].head_prev[k];
    vec_next[(size_t)s * 8 + k] = rows[s].head_next[k];
  }
}
// dense targets: d = 2 (o - t) scale; block b's share of the loss goes to lpart[b] (summed in block order afterwards)
__global__ __launch_bounds__(256) void t_loss_dense(size_t total, float scale, const float *o, const float *t, float *d, float *lpart) {
  __shared__ float red[256];
  float acc = 0.f;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const float err = o[e] - t[e];
    d[e] = 2.f * err * scale;
    acc += err * err * scale;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) lpart[blockIdx.x] = red[0];
}

__global__ void t_adam(size_t cnt, float *w, const float *g, float *m, float *v, float lr_t, float b1, float b2, float eps) {
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < cnt; e += (size_t)gridDim.x * blockDim.x) {
    const float gi = g[e];
    const float mi = b1 * m[e] + (1.f - b1) * gi, vi = b2 * v[e] + (1.f - b2) * gi * gi;
    m[e] = mi; v[e] = vi;
    w[e] -= lr_t * mi / (sqrtf(vi) + eps);
  }
}
__global__ void t_moving(int c_n, float *mean, float *var, const float *stat) {
  const int c = threadIdx.x;
  if (c >= c_n) return;
  mean[c] = 0.99f * mean[c] + 0.01f * stat[2 * c];
  var[c] = 0.99f * var[c] + 0.01f * stat[2 * c + 1];
}

// ---------------------------------------------------------------- host orchestration
#define GRID(total) dim3((unsigned)(((total) + 255) / 256 > 65535 * 16 ? 65535 * 16 : ((total) + 255) / 256)), dim3(256)
#define K(kern, total, ...) do { hipLaunchKernelGGL(kern, GRID(total), 0, st, __VA_ARGS__); OFX_HIP(hipGetLastError()); } while (0)

struct Arena {  // bump allocator over one hipMalloc
  char *base; size_t used, cap;
  bool over = false;
  char *take(size_t bytes) {
    char *p = base + used;
    used += (bytes + 255) & ~(size_t)255;
    if (used > cap) { over = true; used = 0; p = base; }   // never hand out memory past the block; the caller checks `over`
    return p;
  }
  float *f(size_t n) { return (float *)take(n * 4); }
  double *d(size_t n) { return (double *)take(n * 8); }
};

static const int kTI[4] = {2, 8, 8, 8}, kUI[4] = {1, 2, 4, 8}, kUO[4] = {2, 4, 8, 1};

static const int kWSlices = 128;

static void conv_bwd_weight(hipStream_t st, int n, int ci, int co, int H, int W, const float *x, const float *dz,
                            double *part, float *dw, float *db) {
  const int nw = 9 * ci * co;
  // few weights over many pixels (out2, upconv3) need the slices to fill the chip; small maps do not
  const size_t total = (size_t)n * H * W;
  // the partial-sum buffer holds kWSlices * 600 doubles: a layer with few weights can afford more, shorter slices
  const int cap = (int)((size_t)kWSlices * 600 / (size_t)(nw + co));
  int slices = (int)((total + 8191) / 8192);
  slices = slices < 1 ? 1 : slices > cap ? cap : slices;
  if (slices > 1024) slices = 1024;
#define BWW(CIB, COB) hipLaunchKernelGGL((t_conv_bwd_weight<CIB, COB>), dim3((ci / CIB) * (co / COB), slices), dim3(256), 0, st, n, ci, co, H, W, x, dz, part)
  if (co % 4 == 0) BWW(1, 4);
  else if (co % 2 == 0 && ci % 2 == 0) BWW(2, 2);
  else if (co % 2 == 0) BWW(1, 2);
  else if (ci % 4 == 0) BWW(4, 1);
  else BWW(1, 1);
#undef BWW
  hipLaunchKernelGGL(t_conv_bwd_finish, dim3((nw + co + 255) / 256), dim3(256), 0, st, nw, co, slices, part, dw, db);
}

// one blockIdx.y per (sample, channel) plane of `per` elements, up to 64 blocks of 256 threads along it
#define PLANES(per, planes) dim3((unsigned)(((per) + 255) / 256 > 64 ? 64 : ((per) + 255) / 256), (unsigned)(planes)), dim3(256)
#define GRIDP(total) dim3((unsigned)(((total) + 255) / 256 > 65535 * 16 ? 65535 * 16 : ((total) + 255) / 256)), dim3(256)
static int conv_fwd(hipStream_t st, int n, int ci, int co, int H, int W, const float *x, const float *w, const float *b, float *z) {
  const size_t px = (size_t)n * H * W;
  switch (co) {
    case 8: hipLaunchKernelGGL(t_conv_fwd<8>, GRIDP(px), 0, st, n, ci, H, W, x, w, b, z); break;
    case 4: hipLaunchKernelGGL(t_conv_fwd<4>, GRIDP(px), 0, st, n, ci, H, W, x, w, b, z); break;
    case 2: hipLaunchKernelGGL(t_conv_fwd<2>, GRIDP(px), 0, st, n, ci, H, W, x, w, b, z); break;
    case 1: hipLaunchKernelGGL(t_conv_fwd<1>, GRIDP(px), 0, st, n, ci, H, W, x, w, b, z); break;
    default: ofx_set_error("ofx_dqn_fit: no conv kernel for %d output channels", co); return OFX_ERR_STATE;
  }
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}
static int conv_bwd_data(hipStream_t st, int n, int ci, int co, int H, int W, const float *dz, const float *w, float *dx) {
  const size_t px = (size_t)n * H * W;
  switch (ci) {
    case 8: hipLaunchKernelGGL(t_conv_bwd_data<8>, GRIDP(px), 0, st, n, co, H, W, dz, w, dx); break;
    case 4: hipLaunchKernelGGL(t_conv_bwd_data<4>, GRIDP(px), 0, st, n, co, H, W, dz, w, dx); break;
    case 2: hipLaunchKernelGGL(t_conv_bwd_data<2>, GRIDP(px), 0, st, n, co, H, W, dz, w, dx); break;
    case 1: hipLaunchKernelGGL(t_conv_bwd_data<1>, GRIDP(px), 0, st, n, co, H, W, dz, w, dx); break;
    default: ofx_set_error("ofx_dqn_fit: no conv kernel for %d input channels", ci); return OFX_ERR_STATE;
  }
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}

__global__ void t_count_pads(int n, const ofx_transition *rows, int32_t *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && rows[i].ship < 0) atomicAdd(out, 1);
}

// a workspace the handle keeps between calls: grown on demand, given back when a call needs less than a quarter of it (one
// 4096-row fit leaves 40 GB behind - a handle that then trains on 256 rows should not hold them), freed by ofx_destroy
static int keep_workspace(ofx_handle *h, void **buf, size_t *have, size_t need) {
  if (*have >= need && *have / 4 <= need) return OFX_OK;
  OFX_HIP(hipStreamSynchronize(h->stream));
  if (*buf) (void)hipFree(*buf);
  *buf = nullptr; *have = 0;
  OFX_HIP(hipMalloc(buf, need));
  *have = need;
  return OFX_OK;
}

// padding rows (ship < 0) would enter the BatchNorm batch statistics and the loss scale: refused before any work is done
static int refuse_pads(ofx_handle *h, int n, const ofx_transition *rows, const char *who) {
  hipStream_t st = h->stream;
  if (!h->counter) OFX_HIP(hipMalloc((void **)&h->counter, 4 * sizeof(int32_t)));
  int32_t pads = 0;
  OFX_HIP(hipMemsetAsync(h->counter, 0, sizeof(int32_t), st));
  hipLaunchKernelGGL(t_count_pads, dim3((n + 255) / 256), dim3(256), 0, st, n, rows, h->counter);
  OFX_HIP(hipMemcpyAsync(&pads, h->counter, sizeof(pads), hipMemcpyDeviceToHost, st));
  OFX_HIP(hipStreamSynchronize(st));
  if (pads) {
    ofx_set_error("%s: %d of %d rows are padding (ship < 0); gather with ofx_replay_gather_valid", who, pads, n);
    return OFX_ERR_INVALID;
  }
  return OFX_OK;
}

// The lean form of one fit step (default; ofx_fit.hip): the same graph, loss and update as dqn_fit_impl below with only
// z of every convolution in HBM.  The dense layers, the loss and Adam are the plain form's kernels.
static int dqn_fit_lean(ofx_handle *h, float *weights, float *adam_m, float *adam_v, int32_t step, float lr, int32_t n,
                        const ofx_transition *rows, const void *bits_prev, const float *y_act, const float *y_ptr,
                        const float *t1, const float *t2, float *grad_out, float *loss_host) {
  const bool dense = t1 != nullptr;
  OFX_HIP(hipSetDevice(h->cfg.device));
  hipStream_t st = h->stream;
  ofx_policy_desc L;
  int rc = ofx_policy_layout(h, &L);
  if (rc) return rc;
  if (!dense && (rc = refuse_pads(h, n, rows, "ofx_dqn_fit"))) return rc;   // (dense targets: ofx_dqn_fit_reference has checked)
  const size_t N = (size_t)n;
  const int legacy = h->opt_bilinear_legacy;
  static const int tS[4] = {400, 200, 100, 50}, uS[3] = {50, 100, 200};
  size_t need = 0;
  auto sz = [&](size_t bytes) { need += (bytes + 255) & ~(size_t)255; };
  for (int i = 1; i < 4; i++) { sz(4 * N * 8 * tS[i] * tS[i]); sz(4 * N * 8 * tS[i] * tS[i]); }  // z, g (layer 0 has neither)
  sz(4 * ofx_fit_first_floats()); sz(4 * 5008 * 100);
  sz(8 * ofx_fit_first_doubles(n)); sz(8 * ofx_fit_first_part_doubles(n));                                 // correlation, first-layers backward
  for (int i = 0; i < 3; i++) sz(4 * N * 8 * tS[i + 1] * tS[i + 1]);                                       // pooled activation
  for (int j = 0; j < 3; j++) { sz(4 * N * kUO[j] * uS[j] * uS[j]); sz(4 * N * kUO[j] * uS[j] * uS[j]); }
  sz(4 * N * 160000); sz(4 * N * 160000);                                                                  // o2, do2
  sz(4 * N * 5000); sz(4 * N * 5008); sz(4 * N * 100); sz(4 * N * 50); sz(4 * N * 2); sz(4 * N * 625);      // p3 f d1 d2 o1 u0
  sz(4 * N * 625); sz(4 * N * 2); sz(4 * N * 100); sz(4 * N * 50); sz(4 * N * 5008); sz(4 * N * 5000);      // gu0 do1 dd1 dd2 df dp3
  sz(4 * (size_t)L.n_floats); sz(4 * 64); sz(4 * 64); sz(4 * 576); sz(4 * (2 * N + 4096)); sz(8 * ofx_fit_part_doubles()); sz(8 * 32);
  sz(4 * ofx_fit_out_floats()); sz(8 * ofx_fit_out_doubles(n)); sz(8 * ofx_fit_point_doubles(n)); sz(4 * N * 128);
  for (int k = 0; k < 7; k++) { sz(4 * 32); sz(4 * 16); }   // stat (+ the mean's low parts), act
  if ((rc = keep_workspace(h, &h->fitws, &h->fitws_bytes, need))) return rc;
  Arena A{(char *)h->fitws, 0, need};
  auto T = [&](int t) { return weights + L.offset[t]; };
  float *grad = A.f(L.n_floats);
  OFX_HIP(hipMemsetAsync(grad, 0, sizeof(float) * L.n_floats, st));
  auto G = [&](int t) { return grad + L.offset[t]; };
  float *loss = A.f(64), *zero = A.f(64), *wtr = A.f(576);
  OFX_HIP(hipMemsetAsync(loss, 0, 2 * 256, st));   // loss and zero: adjacent 256-byte slots of the arena
  float *lpart = A.f(2 * N + 4096);
  double *part = A.d(ofx_fit_part_doubles()), *sums = A.d(32);
  float *weff = A.f(ofx_fit_out_floats());
  double *fpart = A.d(ofx_fit_out_doubles(n)), *pscratch = A.d(ofx_fit_point_doubles(n));
  float *gpatch = A.f(N * 128);   // textbook targets: all of the last head layer's g that is not zero
  float *tz[4], *tg[4], *tstat[4], *tact[4], *uz[3], *ug[3], *ustat[3], *uact[3], *tp[3];
  for (int i = 0; i < 3; i++) tp[i] = A.f(N * 8 * tS[i + 1] * tS[i + 1]);
  for (int i = 0; i < 4; i++) { tz[i] = i ? A.f(N * 8 * tS[i] * tS[i]) : nullptr; tg[i] = i ? A.f(N * 8 * tS[i] * tS[i]) : nullptr; tstat[i] = A.f(32); tact[i] = A.f(16); }
  float *luts = A.f(ofx_fit_first_floats()), *w1t = A.f(5008 * 100);
  double *cpart = A.d(ofx_fit_first_doubles(n)), *fpart2 = A.d(ofx_fit_first_part_doubles(n));
  for (int j = 0; j < 3; j++) { uz[j] = A.f(N * kUO[j] * uS[j] * uS[j]); ug[j] = A.f(N * kUO[j] * uS[j] * uS[j]); ustat[j] = A.f(32); uact[j] = A.f(16); }
  float *o2 = A.f(N * 160000), *do2 = A.f(N * 160000);
  float *p3 = A.f(N * 5000), *f = A.f(N * 5008), *d1 = A.f(N * 100), *d2 = A.f(N * 50), *o1 = A.f(N * 2), *u0 = A.f(N * 625);
  float *gu0 = A.f(N * 625), *do1 = A.f(N * 2), *dd1 = A.f(N * 100), *dd2 = A.f(N * 50), *df = A.f(N * 5008), *dp3 = A.f(N * 5000);
  if (A.over) { ofx_set_error("ofx_dqn_fit: internal workspace sized too small"); return OFX_ERR_STATE; }
  int nb = 0;
  // input of trunk layer i: the forward pools the previous layer's z on the fly and KEEPS the pooled activation (a quarter
  // of z's bytes); the weight gradient reads that plane back instead of pooling z a second time
  auto trunk_src = [&](int i, bool backward) {
    if (i == 0) return ofx_fit_src{OFX_FIT_SRC_BITS, nullptr, bits_prev, nullptr, 400, 400, legacy};
    if (backward || i == 1) return ofx_fit_src{OFX_FIT_SRC_PLANE, nullptr, tp[i - 1], zero, tS[i], tS[i], legacy};
    return ofx_fit_src{OFX_FIT_SRC_POOL, tp[i - 1], tz[i - 1], tact[i - 1], tS[i - 1], tS[i - 1], legacy};
  };
  auto head_src = [&](int j) {   // input of head-2 layer j (3 = the output convolution)
    return j == 0 ? ofx_fit_src{OFX_FIT_SRC_UPRAW, nullptr, u0, nullptr, 25, 25, legacy}
                  : ofx_fit_src{OFX_FIT_SRC_UP, nullptr, uz[j - 1], uact[j - 1], uS[j - 1], uS[j - 1], legacy};
  };

  // ---- forward (training mode) ----
  // the first layer is never materialised: statistics from the autocorrelation of the bit maps, p0 through the table kernel
  if ((rc = ofx_fit_first_fwd(h, n, bits_prev, T(0), T(1), T(2), T(3), cpart, tstat[0], tact[0], luts, tp[0]))) return rc;
  for (int i = 1; i < 4; i++) {
    const int s = tS[i];
    if ((rc = ofx_fit_conv_fwd(st, n, kTI[i], 8, s, s, trunk_src(i, false), T(6 * i), T(6 * i + 1), tz[i], part, &nb))) return rc;
    if ((rc = ofx_fit_finish(st, nb, 8, (double)N * s * s, part, T(6 * i + 2), T(6 * i + 3), nullptr, tstat[i], tact[i]))) return rc;
  }
  if ((rc = ofx_fit_pool_act(st, n, 8, 50, 50, tz[3], tact[3], p3))) return rc;
  K(t_concat_fwd, N * 5008, n, rows, p3, f, dense ? 1 : 0);
  if ((rc = ofx_launch_gemm(h, f, 5008, T(24), 100, T(25), d1, 100, n, 100, 5008, 1))) return rc;
  if ((rc = ofx_launch_gemm(h, d1, 100, T(26), 50, T(27), d2, 50, n, 50, 100, 1))) return rc;
  if ((rc = ofx_launch_gemm(h, d2, 50, T(28), 2, T(29), o1, 2, n, 2, 50, 0))) return rc;
  if ((rc = ofx_launch_gemm(h, d1, 100, T(30), 625, T(31), u0, 625, n, 625, 100, 1))) return rc;
  for (int j = 0; j < 3; j++) {
    const int s = uS[j];
    if ((rc = ofx_fit_conv_fwd(st, n, kUI[j], kUO[j], s, s, head_src(j), T(32 + 6 * j), T(33 + 6 * j), uz[j], part, &nb))) return rc;
    if ((rc = ofx_fit_finish(st, nb, kUO[j], (double)N * s * s, part, T(34 + 6 * j), T(35 + 6 * j), nullptr, ustat[j], uact[j]))) return rc;
  }
  if (dense && (rc = ofx_fit_out_fwd(st, n, head_src(3), T(50), T(51), o2, weff))) return rc;

  // ---- loss seeds ----
  OFX_HIP(hipMemsetAsync(do1, 0, N * 2 * 4, st));
  if (!dense) {
    // one error per sample on the heat map: the output convolution at the pointer only, its gradients and the last head
    // layer's g from that one pixel (ofx_fit.hip, "the top of head 2 for the textbook targets")
    float *o2p = o2, *d2p = o2 + N;                          // [n] each: the dense planes are not used on this path
    if ((rc = ofx_fit_top_point(st, n, rows, head_src(3), T(50), T(51), o1, y_act, y_ptr, ustat[2], o2p, do1, d2p, lpart, gpatch,
                                pscratch, sums, G(50), G(51)))) return rc;
    hipLaunchKernelGGL(t_sum_ordered, dim3(1), dim3(64), 0, st, n, 2, lpart, loss);
    OFX_HIP(hipGetLastError());
  } else if (dense) {
    const int nb1 = (int)((N * 2 + 255) / 256), nb2 = (int)(N * 160000 / 256 > 2048 ? 2048 : (N * 160000 + 255) / 256);
    hipLaunchKernelGGL(t_loss_dense, dim3(nb1), dim3(256), 0, st, N * 2, 1.f / (2.f * n), o1, t1, do1, lpart);
    hipLaunchKernelGGL(t_sum_ordered, dim3(1), dim3(64), 0, st, nb1, 1, lpart, loss);
    hipLaunchKernelGGL(t_loss_dense, dim3(nb2), dim3(256), 0, st, N * 160000, 1.f / (160000.f * n), o2, t2, do2, lpart + 2048);
    hipLaunchKernelGGL(t_sum_ordered, dim3(1), dim3(64), 0, st, nb2, 1, lpart + 2048, loss + 1);
    OFX_HIP(hipGetLastError());
  }

  // ---- backward: head 2 ----
  if (dense && (rc = ofx_fit_out_bw(st, n, head_src(3), do2, part, fpart, G(50), G(51)))) return rc;
  const float *dzn = do2;
  for (int j = 2; j >= 0; j--) {
    const int s = uS[j];
    if (j < 2 || dense) {   // (the textbook path has the last layer's g and its sums already)
      if ((rc = ofx_fit_b1_up(st, n, kUO[j], kUO[j + 1], s, s, 1, dzn, T(32 + 6 * (j + 1)), uz[j], ustat[j], uact[j], legacy, ug[j], part, &nb, zero, wtr))) return rc;
      if ((rc = ofx_fit_finish(st, nb, kUO[j], 1.0, part, nullptr, nullptr, sums, nullptr, nullptr))) return rc;
    }
    const bool patch = j == 2 && !dense;   // (g of the last layer is a 4 x 4 patch per sample there: not read, dz written)
    if ((rc = ofx_fit_bw(st, n, kUI[j], kUO[j], s, s, head_src(j), 1, ug[j], uz[j], ustat[j], T(34 + 6 * j), sums, part,
                         G(32 + 6 * j), G(33 + 6 * j), G(34 + 6 * j), G(35 + 6 * j), patch ? gpatch : nullptr,
                         patch ? rows : nullptr))) return rc;
    dzn = ug[j];
  }
  if ((rc = ofx_fit_b1_up(st, n, 1, 2, 25, 25, 0, dzn, T(32), u0, nullptr, nullptr, legacy, gu0, part, &nb, zero, wtr))) return rc;
  // gu0 = d u0 [n][625], already behind u0's ReLU mask
  K(t_dense_bwd_w, (size_t)26 * 157 * 8, n, 100, 625, d1, gu0, G(30), G(31));
  K(t_dense_bwd_x, ((N + 3) / 4) * 25, n, 100, 625, gu0, T(30), dd1, 0);
  // ---- backward: head 1 ----
  K(t_dense_bwd_w, (size_t)32 * 8, n, 50, 2, d2, do1, G(28), G(29));
  K(t_dense_bwd_x, ((N + 3) / 4) * 13, n, 50, 2, do1, T(28), dd2, 0);
  K(t_relu_mask, N * 50, N * 50, d2, dd2);
  K(t_dense_bwd_w, (size_t)26 * 13 * 8 + 255, n, 100, 50, d1, dd2, G(26), G(27));
  K(t_dense_bwd_x, ((N + 3) / 4) * 25, n, 100, 50, dd2, T(26), dd1, 1);
  // ---- dense1 + trunk ----
  K(t_relu_mask, N * 100, N * 100, d1, dd1);
  K(t_dense_bwd_w, (size_t)1253 * 25 * 8, n, 5008, 100, f, dd1, G(24), G(25));
  // d f = dd1 x W1^T on the f32 MFMA GEMM (the 5008 x 100 kernel transposed first): 2 x 5008 x 100 FLOP per row
  hipLaunchKernelGGL(t_transpose, dim3((100 + 31) / 32, (5008 + 31) / 32), dim3(256), 0, st, 5008, 100, T(24), w1t);
  OFX_HIP(hipGetLastError());
  if ((rc = ofx_launch_gemm(h, dd1, 100, w1t, 5008, nullptr, df, 5008, n, 5008, 100, 0))) return rc;
  K(t_concat_bwd, N * 5000, n, df, dp3);
  dzn = dp3;
  for (int i = 3; i >= 0; i--) {
    const int s = tS[i];
    if ((rc = ofx_fit_b1_pool(st, n, s, s, i < 3, dzn, i < 3 ? T(6 * (i + 1)) : nullptr, tz[i], tstat[i], tact[i], tg[i], part, &nb, wtr, zero))) return rc;
    if ((rc = ofx_fit_finish(st, nb, 8, 1.0, part, nullptr, nullptr, sums, nullptr, nullptr))) return rc;
    if (i == 1) {
      // the first two layers: the first has neither z, g nor dz, the second's dz is never stored - only the windows that see
      // a set bit are visited (ofx_fit.hip, f_first_bwd)
      if ((rc = ofx_fit_first_bwd(st, n, bits_prev, tg[1], tz[1], tstat[1], T(8), sums, T(6), tp[0], luts, T(0), T(1), tstat[0],
                                  T(2), T(3), fpart2, cpart, G(0), G(1), G(2), G(3), G(6), G(7), G(8), G(9)))) return rc;
      break;
    }
    if ((rc = ofx_fit_bw(st, n, kTI[i], 8, s, s, trunk_src(i, true), 1, tg[i], tz[i], tstat[i], T(6 * i + 2), sums, part,
                         G(6 * i), G(6 * i + 1), G(6 * i + 2), G(6 * i + 3)))) return rc;
    dzn = tg[i];
  }
  if (grad_out) OFX_HIP(hipMemcpyAsync(grad_out, grad, sizeof(float) * L.n_floats, hipMemcpyDeviceToDevice, st));

  // ---- Adam + moving statistics ----
  const float b1 = 0.9f, b2 = 0.999f;
  const float lr_t = lr * sqrtf(1.f - powf(b2, (float)step)) / (1.f - powf(b1, (float)step));
  for (int t = 0; t < L.n_tensors; t++) {
    const bool conv_bn = t < 24 || (t >= 32 && t < 50);
    const int k = conv_bn ? (t < 24 ? t % 6 : (t - 32) % 6) : -1;
    if (k == 4 || k == 5) continue;  // moving mean / variance: not trained
    K(t_adam, (size_t)L.count[t], (size_t)L.count[t], weights + L.offset[t], grad + L.offset[t], adam_m + L.offset[t],
      adam_v + L.offset[t], lr_t, b1, b2, 1e-7f);
  }
  for (int i = 0; i < 4; i++) hipLaunchKernelGGL(t_moving, dim3(1), dim3(64), 0, st, 8, weights + L.offset[6 * i + 4], weights + L.offset[6 * i + 5], tstat[i]);
  for (int j = 0; j < 3; j++) hipLaunchKernelGGL(t_moving, dim3(1), dim3(64), 0, st, kUO[j], weights + L.offset[32 + 6 * j + 4], weights + L.offset[32 + 6 * j + 5], ustat[j]);
  OFX_HIP(hipGetLastError());
  float lh[2] = {0.f, 0.f};
  OFX_HIP(hipMemcpyAsync(lh, loss, sizeof(lh), hipMemcpyDeviceToHost, st));
  OFX_HIP(hipStreamSynchronize(st));
  if (loss_host) { loss_host[0] = lh[0]; loss_host[1] = lh[1]; }
  if ((rc = ofx_policy_weights_updated(h, weights))) return rc;  // a pinned blob is prepared again
  return OFX_OK;
}

// One fit step.  Two forms of the targets: the sparse one of ofx_dqn_fit (one error per head and sample: y_act / y_ptr,
// inputs = `state`) and the dense one of ofx_dqn_fit_reference (t1 [n][2] / t2 [n][400][400] whole target tensors,
// inputs = bits_in + the rows' next_state head).
static int dqn_fit_impl(ofx_handle *h, float *weights, float *adam_m, float *adam_v, int32_t step, float lr, int32_t n,
                        const ofx_transition *rows, const void *bits_prev, const float *y_act, const float *y_ptr,
                        const float *t1, const float *t2, float *grad_out, float *loss_host) {
  if (!h->opt_fit_plain)
    return dqn_fit_lean(h, weights, adam_m, adam_v, step, lr, n, rows, bits_prev, y_act, y_ptr, t1, t2, grad_out, loss_host);
  const bool dense = t1 != nullptr;
  OFX_HIP(hipSetDevice(h->cfg.device));
  hipStream_t st = h->stream;
  ofx_policy_desc L;
  int rc = ofx_policy_layout(h, &L);
  if (rc) return rc;
  if (!dense && (rc = refuse_pads(h, n, rows, "ofx_dqn_fit"))) return rc;   // (dense targets: ofx_dqn_fit_reference has checked)
  const size_t N = (size_t)n;
  const int legacy = h->opt_bilinear_legacy;
  // activations: trunk sizes 400,200,100,50 (z, a per layer + pooled), head-2 sizes 50,100,200 (+ upsampled inputs)
  size_t need = 0;
  auto sz = [&](size_t fl) { need += (fl * 4 + 255) & ~(size_t)255; };
  for (int pass = 0; pass < 1; pass++) {
    sz(N * 2 * 160000);
    for (int i = 0, s = 400; i < 4; i++, s /= 2) { sz(N * 8 * s * s); sz(N * 8 * s * s); sz(N * 8 * (s / 2) * (s / 2)); sz(N * 8 * s * s); sz(N * 8 * s * s); }
    sz(N * 5008); sz(N * 100); sz(N * 50); sz(N * 2); sz(N * 625);
    for (int j = 0, s = 50; j < 3; j++, s *= 2) { sz(N * kUI[j] * s * s); sz(N * kUO[j] * s * s); sz(N * kUO[j] * s * s); sz(N * kUO[j] * s * s); sz(N * kUO[j] * s * s); }
    sz(N * 8 * 160000); sz(N * 160000);               // up4, o2
    sz(N * 8 * 160000); sz(N * 8 * 160000);           // two gradient scratch planes of the largest size
    sz(N * 5008); sz(N * 100); sz(N * 100); sz(N * 50); sz(N * 2); sz(N * 625); sz(N * 160000);
    sz(L.n_floats); sz(64); sz(2 * N + 4096); need += 65536 + 16 * 64 * 8 + 16 * 64 * 2 * 8 + (size_t)kWSlices * 600 * 8;
  }
  if ((rc = keep_workspace(h, &h->fitws, &h->fitws_bytes, need))) return rc;
  void *raw = h->fitws;
  Arena A{(char *)raw, 0, need};
  const float *W_ = weights;
  auto T = [&](int t) { return weights + L.offset[t]; };
  float *grad = A.f(L.n_floats);
  OFX_HIP(hipMemsetAsync(grad, 0, sizeof(float) * L.n_floats, st));
  float *loss = A.f(64);
  OFX_HIP(hipMemsetAsync(loss, 0, 64 * sizeof(float), st));
  float *lpart = A.f(2 * N + 4096);                   // per-sample / per-block loss shares, summed in order
  double *sums = A.d(16 * 16);
  double *spart = A.d(16 * 64 * 2);                  // t_chan_sums partials [channel][slice][2]
  double *wpart = A.d((size_t)kWSlices * 600);
  auto G = [&](int t) { return grad + L.offset[t]; };

  // ---- forward (training mode) ----
  float *x0 = A.f(N * 2 * 160000);
  K(t_bits_to_f32, N * 2 * 160000, n, (const uint32_t *)bits_prev, x0);
  float *tz[4], *ta[4], *tp[4], *tstat[4];
  const float *tin = x0;
  for (int i = 0, s = 400; i < 4; i++, s /= 2) {
    const size_t per = (size_t)s * s;
    tz[i] = A.f(N * 8 * per); ta[i] = A.f(N * 8 * per); tp[i] = A.f(N * 8 * per / 4); tstat[i] = A.f(16);
    if ((rc = conv_fwd(st, n, kTI[i], 8, s, s, tin, T(6 * i), T(6 * i + 1), tz[i]))) return rc;
    hipLaunchKernelGGL(t_chan_sums, dim3(8, 64), dim3(256), 0, st, n, 8, per, tz[i], (const float *)nullptr, spart);
    hipLaunchKernelGGL(t_chan_sums_finish, dim3(1), dim3(64), 0, st, 8, 64, spart, sums);
    hipLaunchKernelGGL(t_bn_finish_stats, dim3(1), dim3(64), 0, st, 8, (double)N * (double)per, sums, tstat[i]);
    hipLaunchKernelGGL(t_bn_relu_fwd, PLANES(per, n * 8), 0, st, 8, per, tz[i], tstat[i], T(6 * i + 2), T(6 * i + 3), ta[i]);
    K(t_pool_fwd, N * 8 * per / 4, n * 8, s, s, ta[i], tp[i]);
    tin = tp[i];
  }
  float *f = A.f(N * 5008), *d1 = A.f(N * 100), *d2 = A.f(N * 50), *o1 = A.f(N * 2), *u0 = A.f(N * 625);
  K(t_concat_fwd, N * 5008, n, rows, tp[3], f, dense ? 1 : 0);
  // the four dense layers on the f32 MFMA (k_gemm_f32: exact fp32 products, fp32 accumulation in MFMA order)
  if ((rc = ofx_launch_gemm(h, f, 5008, T(24), 100, T(25), d1, 100, n, 100, 5008, 1))) return rc;
  if ((rc = ofx_launch_gemm(h, d1, 100, T(26), 50, T(27), d2, 50, n, 50, 100, 1))) return rc;
  if ((rc = ofx_launch_gemm(h, d2, 50, T(28), 2, T(29), o1, 2, n, 2, 50, 0))) return rc;
  if ((rc = ofx_launch_gemm(h, d1, 100, T(30), 625, T(31), u0, 625, n, 625, 100, 1))) return rc;
  float *uu[3], *uz[3], *ua[3], *ustat[3];
  const float *uin = u0;
  for (int j = 0, s = 50; j < 3; j++, s *= 2) {
    const size_t per = (size_t)s * s;
    uu[j] = A.f(N * kUI[j] * per); uz[j] = A.f(N * kUO[j] * per); ua[j] = A.f(N * kUO[j] * per); ustat[j] = A.f(16);
    K(t_up_fwd, N * kUI[j] * per, n * kUI[j], s / 2, s / 2, uin, uu[j], legacy);
    if ((rc = conv_fwd(st, n, kUI[j], kUO[j], s, s, uu[j], T(32 + 6 * j), T(33 + 6 * j), uz[j]))) return rc;
    hipLaunchKernelGGL(t_chan_sums, dim3(kUO[j], 64), dim3(256), 0, st, n, kUO[j], per, uz[j], (const float *)nullptr, spart);
    hipLaunchKernelGGL(t_chan_sums_finish, dim3(1), dim3(64), 0, st, kUO[j], 64, spart, sums);
    hipLaunchKernelGGL(t_bn_finish_stats, dim3(1), dim3(64), 0, st, kUO[j], (double)N * (double)per, sums, ustat[j]);
    hipLaunchKernelGGL(t_bn_relu_fwd, PLANES(per, n * kUO[j]), 0, st, kUO[j], per, uz[j], ustat[j], T(34 + 6 * j), T(35 + 6 * j), ua[j]);
    uin = ua[j];
  }
  float *up4 = A.f(N * 8 * 160000), *o2 = A.f(N * 160000);
  K(t_up_fwd, N * 8 * 160000, n * 8, 200, 200, ua[2], up4, legacy);
  if ((rc = conv_fwd(st, n, 8, 1, 400, 400, up4, T(50), T(51), o2))) return rc;

  // ---- loss seeds ----
  float *do1 = A.f(N * 2), *do2 = A.f(N * 160000);
  OFX_HIP(hipMemsetAsync(do1, 0, N * 2 * 4, st));
  OFX_HIP(hipMemsetAsync(do2, 0, N * 160000 * 4, st));
  if (dense) {
    {
      const int nb1 = (int)((N * 2 + 255) / 256), nb2 = (int)(N * 160000 / 256 > 2048 ? 2048 : (N * 160000 + 255) / 256);
      hipLaunchKernelGGL(t_loss_dense, dim3(nb1), dim3(256), 0, st, N * 2, 1.f / (2.f * n), o1, t1, do1, lpart);
      hipLaunchKernelGGL(t_sum_ordered, dim3(1), dim3(64), 0, st, nb1, 1, lpart, loss);
      hipLaunchKernelGGL(t_loss_dense, dim3(nb2), dim3(256), 0, st, N * 160000, 1.f / (160000.f * n), o2, t2, do2, lpart + 2048);
      hipLaunchKernelGGL(t_sum_ordered, dim3(1), dim3(64), 0, st, nb2, 1, lpart + 2048, loss + 1);
      OFX_HIP(hipGetLastError());
    }
  } else {
    K(t_loss_seed, N, n, rows, o1, o2, y_act, y_ptr, do1, do2, lpart);
    hipLaunchKernelGGL(t_sum_ordered, dim3(1), dim3(64), 0, st, n, 2, lpart, loss);
  }

  // ---- backward: head 2 ----
  float *gA = A.f(N * 8 * 160000), *gB = A.f(N * 8 * 160000);  // gradient scratch (largest tensors)
  conv_bwd_weight(st, n, 8, 1, 400, 400, up4, do2, wpart, G(50), G(51));
  if ((rc = conv_bwd_data(st, n, 8, 1, 400, 400, do2, T(50), gA))) return rc;   // d up4
  float *dcur = gB;                                                          // d ua[2]
  K(t_up_bwd, N * 8 * 40000, n * 8, 200, 200, gA, dcur, legacy);
  for (int j = 2, s = 200; j >= 0; j--, s /= 2) {
    const size_t per = (size_t)s * s;
    float *xh = gA;                                                          // reuse as xhat
    hipLaunchKernelGGL(t_bn_relu_bwd_pre, PLANES(per, n * kUO[j]), 0, st, kUO[j], per, uz[j], ua[j], ustat[j], dcur, xh);
    hipLaunchKernelGGL(t_chan_sums, dim3(kUO[j], 64), dim3(256), 0, st, n, kUO[j], per, dcur, xh, spart);
    hipLaunchKernelGGL(t_chan_sums_finish, dim3(1), dim3(64), 0, st, kUO[j], 64, spart, sums);
    float *dz = ua[j];                                                       // the activation is dead now: holds dz
    hipLaunchKernelGGL(t_bn_bwd, PLANES(per, n * kUO[j]), 0, st, kUO[j], per, (double)N * (double)per, dcur, xh, ustat[j], T(34 + 6 * j), sums, dz, G(34 + 6 * j), G(35 + 6 * j));
    conv_bwd_weight(st, n, kUI[j], kUO[j], s, s, uu[j], dz, wpart, G(32 + 6 * j), G(33 + 6 * j));
    float *duu = gA;                                                         // d (upsampled input)
    if ((rc = conv_bwd_data(st, n, kUI[j], kUO[j], s, s, dz, T(32 + 6 * j), duu))) return rc;
    float *dprev = gB;                                                       // d (previous activation / u0)
    K(t_up_bwd, N * kUI[j] * per / 4, n * kUI[j], s / 2, s / 2, duu, dprev, legacy);
    dcur = dprev;
  }
  // dcur = d u0 [n][625] (pre-mask)
  float *dd1 = A.f(N * 100), *dd1b = A.f(N * 100), *dd2 = A.f(N * 50), *df = A.f(N * 5008);
  K(t_relu_mask, N * 625, N * 625, u0, dcur);
  K(t_dense_bwd_w, (size_t)26 * 157 * 8, n, 100, 625, d1, dcur, G(30), G(31));
  K(t_dense_bwd_x, ((N + 3) / 4) * 25, n, 100, 625, dcur, T(30), dd1, 0);
  // ---- backward: head 1 ----
  K(t_dense_bwd_w, (size_t)32 * 8, n, 50, 2, d2, do1, G(28), G(29));
  K(t_dense_bwd_x, ((N + 3) / 4) * 13, n, 50, 2, do1, T(28), dd2, 0);
  K(t_relu_mask, N * 50, N * 50, d2, dd2);
  K(t_dense_bwd_w, (size_t)26 * 13 * 8 + 255, n, 100, 50, d1, dd2, G(26), G(27));
  K(t_dense_bwd_x, ((N + 3) / 4) * 25, n, 100, 50, dd2, T(26), dd1, 1);
  (void)dd1b;
  // ---- dense1 + trunk ----
  K(t_relu_mask, N * 100, N * 100, d1, dd1);
  K(t_dense_bwd_w, (size_t)1253 * 25 * 8, n, 5008, 100, f, dd1, G(24), G(25));
  K(t_dense_bwd_x, ((N + 3) / 4) * 1252, n, 5008, 100, dd1, T(24), df, 0);
  float *dp = gB;
  K(t_concat_bwd, N * 5000, n, df, dp);                                      // d tp[3] [n][8][25][25]
  for (int i = 3, s = 50; i >= 0; i--, s *= 2) {
    const size_t per = (size_t)s * s, tot = N * 8 * per;
    float *da = gA;
    K(t_pool_bwd, tot / 4, n * 8, s, s, ta[i], dp, da);
    float *xh = tp[i];                                                       // pooled output is dead: reuse? too small -> use gB tail
    xh = gB + N * 8 * 40000;                                                 // second half of gB (>= N*8*per for s <= 200)
    if (s == 400) xh = up4;                                                  // the 400^2 layer: up4 (8 x 400^2) is dead by now
    hipLaunchKernelGGL(t_bn_relu_bwd_pre, PLANES(per, n * 8), 0, st, 8, per, tz[i], ta[i], tstat[i], da, xh);
    hipLaunchKernelGGL(t_chan_sums, dim3(8, 64), dim3(256), 0, st, n, 8, per, da, xh, spart);
    hipLaunchKernelGGL(t_chan_sums_finish, dim3(1), dim3(64), 0, st, 8, 64, spart, sums);
    float *dz = ta[i];
    hipLaunchKernelGGL(t_bn_bwd, PLANES(per, n * 8), 0, st, 8, per, (double)N * (double)per, da, xh, tstat[i], T(6 * i + 2), sums, dz, G(6 * i + 2), G(6 * i + 3));
    const float *xin = i == 0 ? x0 : tp[i - 1];
    conv_bwd_weight(st, n, kTI[i], 8, s, s, xin, dz, wpart, G(6 * i), G(6 * i + 1));
    if (i > 0) {
      dp = gB;                                                               // d tp[i-1] [n][8][s][s]
      if ((rc = conv_bwd_data(st, n, 8, 8, s, s, dz, T(6 * i), dp))) return rc;
    }
  }
  if (A.over) {  // sizing bug guard: nothing has touched the weights yet
    ofx_set_error("ofx_dqn_fit: internal workspace sized too small");
    return OFX_ERR_STATE;
  }
  if (grad_out) OFX_HIP(hipMemcpyAsync(grad_out, grad, sizeof(float) * L.n_floats, hipMemcpyDeviceToDevice, st));

  // ---- Adam + moving statistics ----
  const float b1 = 0.9f, b2 = 0.999f;
  const float lr_t = lr * sqrtf(1.f - powf(b2, (float)step)) / (1.f - powf(b1, (float)step));
  for (int t = 0; t < L.n_tensors; t++) {
    const bool conv_bn = t < 24 || (t >= 32 && t < 50);
    const int k = conv_bn ? (t < 24 ? t % 6 : (t - 32) % 6) : -1;
    if (k == 4 || k == 5) continue;  // moving mean / variance: not trained
    K(t_adam, (size_t)L.count[t], (size_t)L.count[t], weights + L.offset[t], grad + L.offset[t], adam_m + L.offset[t],
      adam_v + L.offset[t], lr_t, b1, b2, 1e-7f);
  }
  for (int i = 0; i < 4; i++) hipLaunchKernelGGL(t_moving, dim3(1), dim3(64), 0, st, 8, weights + L.offset[6 * i + 4], weights + L.offset[6 * i + 5], tstat[i]);
  for (int j = 0; j < 3; j++) hipLaunchKernelGGL(t_moving, dim3(1), dim3(64), 0, st, kUO[j], weights + L.offset[32 + 6 * j + 4], weights + L.offset[32 + 6 * j + 5], ustat[j]);
  OFX_HIP(hipGetLastError());
  float lh[2] = {0.f, 0.f};
  OFX_HIP(hipMemcpyAsync(lh, loss, sizeof(lh), hipMemcpyDeviceToHost, st));
  OFX_HIP(hipStreamSynchronize(st));
  if (loss_host) { loss_host[0] = lh[0]; loss_host[1] = lh[1]; }
  (void)W_;
  if ((rc = ofx_policy_weights_updated(h, weights))) return rc;  // a pinned blob is prepared again
  return OFX_OK;
}

extern "C" int ofx_dqn_fit(ofx_handle *h, float *weights, float *adam_m, float *adam_v, int32_t step, float lr, int32_t n,
                           const ofx_transition *rows, const void *bits_prev, const float *y_act, const float *y_ptr,
                           float *grad_out, float *loss_host) {
  if (!h || !weights || !adam_m || !adam_v || !rows || !bits_prev || !y_act || !y_ptr || n < 1 || step < 1) {
    ofx_set_error("ofx_dqn_fit: bad argument");
    return OFX_ERR_INVALID;
  }
  return dqn_fit_impl(h, weights, adam_m, adam_v, step, lr, n, rows, bits_prev, y_act, y_ptr, nullptr, nullptr, grad_out,
                      loss_host);
}

// Trainer.replay's loop body and fit exactly as written (agents/qlearnIA_V2.py:251-285), quirks included:
//   [target, ptr_target]         = predict(state)                                   (:266, inference-mode BatchNorm)
//   [prediction, ptr_prediction] = predict(next_state)                              (:274)
//   target[iaction]      = reward + gamma max(prediction)     int(not done)         (:279)
//   ptr_target[ipointer] = reward + gamma max(ptr_prediction) int(not done)         (:280) - ipointer = (x, y) indexes
//                          the (400, 400, 1) prediction as [x][y]: row x, column y, the transpose of the pixel that
//                          get_best_action's unravel_index(order='F') named
//   inputs1[i] = img_input (re-bound to NEXT_state's maps at :273), inputs2[i] = next_obs.vector[:8]     (:282-283)
//   fit(x = inputs, y = [target, ptr_target]): training-mode forward on next_state against targets built from state
// so every output element carries an error, not one per head.
extern "C" int ofx_dqn_fit_reference(ofx_handle *h, float *weights, float *adam_m, float *adam_v, int32_t step, float lr,
                                     int32_t n, const ofx_transition *rows, const void *bits_prev, const void *bits_next,
                                     float gamma, float *grad_out, float *loss_host) {
  if (!h || !weights || !adam_m || !adam_v || !rows || !bits_prev || !bits_next || n < 1 || step < 1) {
    ofx_set_error("ofx_dqn_fit_reference: bad argument");
    return OFX_ERR_INVALID;
  }
  OFX_HIP(hipSetDevice(h->cfg.device));
  hipStream_t st = h->stream;
  int rc;
  if ((rc = refuse_pads(h, n, rows, "ofx_dqn_fit_reference"))) return rc;  // before the two predict passes
  const size_t N = (size_t)n;
  // vec_prev [n][8], vec_next [n][8], act_next [n][2], max_next [n], t1 [n][2], t2 [n][160000]
  if ((rc = keep_workspace(h, &h->fitws2, &h->fitws2_bytes, sizeof(float) * N * (8 + 8 + 2 + 1 + 2 + 160000)))) return rc;
  float *buf = (float *)h->fitws2;
  float *vec_prev = buf, *vec_next = vec_prev + 8 * N, *act_next = vec_next + 8 * N, *max_next = act_next + 2 * N;
  float *t1 = max_next + N, *t2 = t1 + 2 * N;
  K(t_unpack_heads, N, n, rows, vec_prev, vec_next);
  if ((rc = ofx_policy_predict_obs(h, weights, n, bits_prev, vec_prev, t1, t2, nullptr))) return rc;
  if ((rc = ofx_policy_predict_obs(h, weights, n, bits_next, vec_next, act_next, nullptr, max_next))) return rc;
  K(t_reference_targets, N, n, rows, gamma, act_next, max_next, t1, t2);
  return dqn_fit_impl(h, weights, adam_m, adam_v, step, lr, n, rows, bits_next, nullptr, nullptr, t1, t2, grad_out, loss_host);
}
