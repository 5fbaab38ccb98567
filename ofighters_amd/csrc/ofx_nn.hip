// ofx_nn.hip - the from-scratch sigmoid MLP forward
// (agents/neural_network.py:396-420 Neural_network.feed, sigmoid :49-53,
//  max_sol_index :423-429; lib/renforcement_learning_neural_network.py:30-60
//  take_action = feed + argmax).  float64 like the reference.
//
// Generic layer: one wavefront per (sample, output neuron); lanes stride the
// input dimension (coalesced rows of W), shuffle-reduce, sigmoid.
// Observation-fed first layer: the input is toVector's
// [8 scalars | ship_map.ravel() | laser_map.ravel()] (observation.py:119-125)
// whose 2*W*H tail is binary and identical for every ship of an arena, so it
// is consumed as a sparse gather-sum of W_1 columns at the set cells of the
// arena's bit maps, once per arena; only the 8-scalar head is per ship.
#include "ofx_internal.h"

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ inline double sigmoid64(double v) { return 1.0 / (1.0 + exp(-v)); }

// y[b][o] = sigmoid(W[o][:] . x[b][:] + B[o])
__global__ __launch_bounds__(256) void k_mlp_layer(const double *W, const double *B, const double *x, double *y,
                                                   int batch, int nin, int nout) {
  const int lane = threadIdx.x & 63;
  const long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wv >= (long)batch * nout) return;
  const int b = (int)(wv / nout), o = (int)(wv - (long)b * nout);
  const double *w = W + (size_t)o * nin, *xi = x + (size_t)b * nin;
  double acc = 0.0;
  for (int k = lane; k < nin; k += 64) acc += w[k] * xi[k];
  acc = wave_sum(acc);
  if (lane == 0) y[(size_t)b * nout + o] = sigmoid64(acc + B[o]);
}

__global__ void k_argmax_rows(const double *y, int batch, int n, int32_t *out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  const double *r = y + (size_t)b * n;
  int best = 0;
  double bv = r[0];
  for (int k = 1; k < n; k++)
    if (r[k] > bv) { bv = r[k]; best = k; }  // first maximum, like np.argmax
  out[b] = best;
}

// per arena: tail[a][o] = sum over set cells c of W1[o][8 + c] (ship map) +
// W1[o][8 + cells + c] (laser map).  One block per arena, thread per bitmap word.
__global__ __launch_bounds__(256) void k_obs_tail(const unsigned *ship_bits, const unsigned *laser_bits, int words,
                                                  const double *W1, int nin, int n1, double *tail) {
  extern __shared__ double red[];  // [4 waves][n1]
  const int a = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const size_t cells = (size_t)words * 32;
  for (int o0 = 0; o0 < n1; o0 += 8) {  // 8 output neurons per pass keeps accumulators in registers
    double acc[8];
#pragma unroll
    for (int q = 0; q < 8; q++) acc[q] = 0.0;
    for (int which = 0; which < 2; which++) {
      const unsigned *bits = (which ? laser_bits : ship_bits) + (size_t)a * words;
      const size_t col0 = 8 + (which ? cells : 0);
      for (int w = tid; w < words; w += 256) {
        unsigned v = bits[w];
        while (v) {
          const int bpos = __ffs((int)v) - 1;
          v &= v - 1;
          const size_t col = col0 + (size_t)w * 32 + bpos;
#pragma unroll
          for (int q = 0; q < 8; q++)
            if (o0 + q < n1) acc[q] += W1[(size_t)(o0 + q) * nin + col];
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const double s = wave_sum(acc[q]);
      if (lane == 0 && o0 + q < n1) red[wv * n1 + o0 + q] = s;
    }
  }
  __syncthreads();
  for (int o = tid; o < n1; o += 256)
    tail[(size_t)a * n1 + o] = red[o] + red[n1 + o] + red[2 * n1 + o] + red[3 * n1 + o];
}

// first layer for every (arena, ship): head (8 scalars) + shared tail
__global__ void k_obs_first(int N, int M, int W, int H, ofx_state st, const double *W1, const double *B1, int nin,
                            int n1, const double *tail, double *y) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * M * n1) return;
  const int o = t % n1, s = t / n1, a = s / M;
  const double *w = W1 + (size_t)o * nin;
  double acc = w[0] * (double)st.reward[s];
  acc += w[1] * 1.0;
  acc += w[2] * (double)st.ship_px[s];
  acc += w[3] * (double)st.ship_py[s];
  acc += w[4] * (double)W;
  acc += w[5] * (double)H;
  acc += w[6] * (double)st.ship_x[s];
  acc += w[7] * (double)st.ship_y[s];
  acc += tail[(size_t)a * n1 + o];
  y[t] = sigmoid64(acc + B1[o]);
}

static int check_layers(const int32_t *layers, int n_layers) {
  if (!layers || n_layers < 2 || n_layers > 64) { ofx_set_error("scratch feed: need 2..64 layers"); return OFX_ERR_INVALID; }
  for (int i = 0; i < n_layers; i++)
    if (layers[i] < 1) { ofx_set_error("scratch feed: layer %d has size %d", i, layers[i]); return OFX_ERR_INVALID; }
  return OFX_OK;
}

// runs layers [first .. n_layers-1) from activations `cur` (batch x layers[first]); result in y
static int run_layers(ofx_handle *h, const int32_t *layers, int n_layers, int first, const double *weights,
                      const double *biases, const double *cur, int batch, double *y, double *ping, double *pong) {
  size_t woff = 0, boff = 0;
  for (int l = 0; l < first; l++) { woff += (size_t)layers[l] * layers[l + 1]; boff += layers[l + 1]; }
  for (int l = first; l + 1 < n_layers; l++) {
    const int nin = layers[l], nout = layers[l + 1];
    double *dst = (l + 2 == n_layers) ? y : (((l - first) & 1) ? pong : ping);
    const long waves = (long)batch * nout;
    hipLaunchKernelGGL(k_mlp_layer, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, h->stream, weights + woff,
                       biases + boff, cur, dst, batch, nin, nout);
    OFX_HIP(hipGetLastError());
    cur = dst;
    woff += (size_t)nin * nout;
    boff += nout;
  }
  return OFX_OK;
}

extern "C" int ofx_scratch_feed(ofx_handle *h, const int32_t *layers, int32_t n_layers, const double *weights,
                                const double *biases, const double *x, int32_t batch, double *y, int32_t *argmax) {
  if (!h || !weights || !biases || !x || !y || batch < 1) { ofx_set_error("ofx_scratch_feed: bad argument"); return OFX_ERR_INVALID; }
  int rc = check_layers(layers, n_layers);
  if (rc) return rc;
  OFX_HIP(hipSetDevice(h->cfg.device));
  int maxw = 0;
  for (int i = 1; i + 1 < n_layers; i++) maxw = layers[i] > maxw ? layers[i] : maxw;
  const size_t act = (size_t)batch * (maxw ? maxw : 1) * sizeof(double);
  if ((rc = ofx_ensure_scratch(h, 2 * act))) return rc;
  double *ping = (double *)h->scratch, *pong = (double *)((char *)h->scratch + act);
  if ((rc = run_layers(h, layers, n_layers, 0, weights, biases, x, batch, y, ping, pong))) return rc;
  if (argmax) {
    hipLaunchKernelGGL(k_argmax_rows, dim3((batch + 255) / 256), dim3(256), 0, h->stream, y, batch,
                       layers[n_layers - 1], argmax);
    OFX_HIP(hipGetLastError());
  }
  return OFX_OK;
}

extern "C" int ofx_scratch_feed_obs(ofx_handle *h, const int32_t *layers, int32_t n_layers, const double *weights,
                                    const double *biases, double *y, int32_t *argmax) {
  if (!h || !weights || !biases || !y) { ofx_set_error("ofx_scratch_feed_obs: bad argument"); return OFX_ERR_INVALID; }
  int rc = check_layers(layers, n_layers);
  if (rc) return rc;
  if (!h->spawned) { ofx_set_error("You must execute analyse_battleground first."); return OFX_ERR_STATE; }
  const ofx_config &c = h->cfg;
  const size_t cells = (size_t)c.width * c.height;
  if ((size_t)layers[0] != 8 + 2 * cells) {
    ofx_set_error("ofx_scratch_feed_obs: layers[0] must be 8 + 2*W*H = %zu (got %d)", 8 + 2 * cells, layers[0]);
    return OFX_ERR_INVALID;
  }
  OFX_HIP(hipSetDevice(c.device));
  if ((rc = ofx_launch_raster(h, OFX_MAP_BITS_LSB, nullptr, nullptr))) return rc;
  const int batch = c.n_arenas * c.n_ships, n1 = layers[1];
  int maxw = n1;
  for (int i = 1; i + 1 < n_layers; i++) maxw = layers[i] > maxw ? layers[i] : maxw;
  const size_t act = (size_t)batch * maxw * sizeof(double), tailb = (size_t)c.n_arenas * n1 * sizeof(double);
  if ((rc = ofx_ensure_scratch(h, 3 * act + tailb))) return rc;
  double *first = (double *)h->scratch;
  double *ping = (double *)((char *)h->scratch + act), *pong = (double *)((char *)h->scratch + 2 * act);
  double *tail = (double *)((char *)h->scratch + 3 * act);
  const int words = (int)(cells >> 5);
  hipLaunchKernelGGL(k_obs_tail, dim3(c.n_arenas), dim3(256), sizeof(double) * 4 * n1, h->stream,
                     (const unsigned *)h->maps[OFX_MAP_BITS_LSB][0], (const unsigned *)h->maps[OFX_MAP_BITS_LSB][1],
                     words, weights, layers[0], n1, tail);
  OFX_HIP(hipGetLastError());
  double *dst1 = (n_layers == 2) ? y : first;
  const int T = batch * n1;
  hipLaunchKernelGGL(k_obs_first, dim3((T + 255) / 256), dim3(256), 0, h->stream, c.n_arenas, c.n_ships, c.width,
                     c.height, h->st, weights, biases, layers[0], n1, tail, dst1);
  OFX_HIP(hipGetLastError());
  if (n_layers > 2 && (rc = run_layers(h, layers, n_layers, 1, weights, biases, first, batch, y, ping, pong))) return rc;
  if (argmax) {
    hipLaunchKernelGGL(k_argmax_rows, dim3((batch + 255) / 256), dim3(256), 0, h->stream, y, batch,
                       layers[n_layers - 1], argmax);
    OFX_HIP(hipGetLastError());
  }
  return OFX_OK;
}
