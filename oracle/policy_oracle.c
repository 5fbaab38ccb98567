/*
 * policy_oracle.c - CPU restatement of the bi-head "pointer_model" forward.
 * TEST INFRASTRUCTURE ONLY (same rules as ofx_oracle.c).
 *
 * PARITY UNPINNED: the arithmetic of this path lives in keras + tensorflow
 * (both unpinned in the reference's requirements.txt:1-2 / setup.py:15, absent
 * from every interpreter of the build image, no weights shipped, no reference
 * test at this boundary).  This file restates the GRAPH of
 * agents/qlearnIA_V2.py:123-190 and the inference glue of :206-220 with the
 * Keras layer defaults written out, and is cross-checked against torch CPU ops
 * (tests/test_policy_oracle.py) - an independent check of the restatement, not
 * a reference oracle.
 *
 * Declared conventions (Keras defaults unless noted):
 *   image input  (400,400,2) NHWC: [row=y][col=x][c], c0 ship_map, c1 laser_map
 *                (np.stack((ship_map, laser_map), axis=2), qlearnIA_V2.py:208)
 *   vector input (8,) = obs.vector[:8] raw, un-normalised (qlearnIA_V2.py:210)
 *   Conv2D       kernel HWIO [3][3][Cin][Cout], padding 'same' (zero pad 1), stride 1
 *   BatchNorm    inference: y = x*inv + (beta - mean*inv), inv = gamma / sqrt(var + 1e-3)
 *   MaxPooling2D 2x2 stride 2 'valid'
 *   Flatten      (h, w, c) order ; Concatenate([vector, flat]) : vector FIRST
 *   Dense        kernel (in, out) ; y = act(x @ K + b)
 *   Reshape      (25,25,1) row-major
 *   UpSampling2D (2,2) bilinear.  The reference's unpinned keras/tensorflow range (requirements.txt:1-2) admits two
 *                conventions for UpSampling2D(interpolation='bilinear') (qlearnIA_V2.py:166,172,178,184):
 *                  legacy_bilinear = 0 (declared default): HALF-PIXEL centres, src = (dst + 0.5)/2 - 0.5
 *                      (tf.image.resize of TF2 / tf.keras = torch interpolate(align_corners=False));
 *                  legacy_bilinear = 1: TF1 resize_bilinear(align_corners=False) without half-pixel centres,
 *                      src = dst/2 (standalone Keras 2.x on TF 1.x): out[2k] = in[k],
 *                      out[2k+1] = (in[k] + in[min(k+1, n-1)])/2.
 *                Both clamp at the edges.  Neither can be pinned here (no keras/tensorflow in the image).
 *   outputs      act_values (2,), ptr_values (400,400)
 *   post         iaction = argmax(act) ; ipointer = unravel_index(argmax(ptr),
 *                (400,400), order='F') = (k % 400, k // 400) = (x, y)
 *
 * Weight blob (float32), tensors in this order (n = tensor index):
 *   for i in 1..4 (trunk, Cin = 2,8,8,8 ; Cout = 8):
 *       conv_i.kernel[3][3][Cin][8], conv_i.bias[8], bn_i.gamma[8], bn_i.beta[8], bn_i.mean[8], bn_i.var[8]
 *   dense1.kernel[5008][100], dense1.bias[100]
 *   dense2.kernel[100][50],   dense2.bias[50]
 *   output1.kernel[50][2],    output1.bias[2]
 *   updense1.kernel[100][625], updense1.bias[625]
 *   for i in 1..3 (head, Cin = 1,2,4 ; Cout = 2,4,8):
 *       upconv_i.kernel[3][3][Cin][Cout], upconv_i.bias, upbn_i.gamma, .beta, .mean, .var
 *   upconv4.kernel[3][3][8][1], upconv4.bias[1]
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BN_EPS 1e-3f

static const int TRUNK_CIN[4] = {2, 8, 8, 8};
static const int UP_CIN[4] = {1, 2, 4, 8};
static const int UP_COUT[4] = {2, 4, 8, 1};

/* number of tensors = 4*6 + 8 + 3*6 + 2 = 52 */
int orc_policy_layout(int32_t *offset, int32_t *count) {
  int n = 0, off = 0;
#define T(c) do { offset[n] = off; count[n] = (c); off += (c); n++; } while (0)
  for (int i = 0; i < 4; i++) { T(9 * TRUNK_CIN[i] * 8); T(8); T(8); T(8); T(8); T(8); }
  T(5008 * 100); T(100);
  T(100 * 50); T(50);
  T(50 * 2); T(2);
  T(100 * 625); T(625);
  for (int i = 0; i < 3; i++) { int co = UP_COUT[i]; T(9 * UP_CIN[i] * co); T(co); T(co); T(co); T(co); T(co); }
  T(9 * 8 * 1); T(1);
#undef T
  offset[n] = off; /* total */
  return n;
}

/* conv 3x3 'same' on HWC float image; out[h][w][co] */
static void conv3x3(const float *in, int H, int W, int cin, const float *k, const float *b, int cout, float *out) {
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++)
      for (int co = 0; co < cout; co++) {
        float acc = 0.0f;
        for (int dy = 0; dy < 3; dy++) {
          int yy = y + dy - 1;
          if (yy < 0 || yy >= H) continue;
          for (int dx = 0; dx < 3; dx++) {
            int xx = x + dx - 1;
            if (xx < 0 || xx >= W) continue;
            const float *ip = in + ((size_t)yy * W + xx) * cin;
            const float *kp = k + ((size_t)(dy * 3 + dx) * cin) * cout + co;
            for (int ci = 0; ci < cin; ci++) acc += ip[ci] * kp[(size_t)ci * cout];
          }
        }
        out[((size_t)y * W + x) * cout + co] = acc + b[co];
      }
}

static void bn_relu(float *x, size_t npix, int c, const float *g, const float *be, const float *mu, const float *var) {
  for (int ch = 0; ch < c; ch++) {
    float inv = g[ch] / sqrtf(var[ch] + BN_EPS);
    float sh = be[ch] - mu[ch] * inv;
    for (size_t p = 0; p < npix; p++) {
      float v = x[p * c + ch] * inv + sh;
      x[p * c + ch] = v > 0.0f ? v : 0.0f;
    }
  }
}

static void maxpool2(const float *in, int H, int W, int c, float *out) {
  int Ho = H / 2, Wo = W / 2;
  for (int y = 0; y < Ho; y++)
    for (int x = 0; x < Wo; x++)
      for (int ch = 0; ch < c; ch++) {
        float m = in[((size_t)(2 * y) * W + 2 * x) * c + ch];
        float v;
        v = in[((size_t)(2 * y) * W + 2 * x + 1) * c + ch]; if (v > m) m = v;
        v = in[((size_t)(2 * y + 1) * W + 2 * x) * c + ch]; if (v > m) m = v;
        v = in[((size_t)(2 * y + 1) * W + 2 * x + 1) * c + ch]; if (v > m) m = v;
        out[((size_t)y * Wo + x) * c + ch] = m;
      }
}

/* bilinear x2, edge clamp; half-pixel centres, or (legacy) the TF1 mapping src = dst / 2 */
static void upsample2(const float *in, int H, int W, int c, float *out, int legacy) {
  int Ho = 2 * H, Wo = 2 * W;
  for (int y = 0; y < Ho; y++) {
    float sy = legacy ? (float)y * 0.5f : ((float)y + 0.5f) * 0.5f - 0.5f;
    float fy = floorf(sy);
    int y0 = (int)fy, y1 = y0 + 1;
    float ly = sy - fy;
    if (y0 < 0) y0 = 0;
    if (y1 > H - 1) y1 = H - 1;
    for (int x = 0; x < Wo; x++) {
      float sx = legacy ? (float)x * 0.5f : ((float)x + 0.5f) * 0.5f - 0.5f;
      float fx = floorf(sx);
      int x0 = (int)fx, x1 = x0 + 1;
      float lx = sx - fx;
      if (x0 < 0) x0 = 0;
      if (x1 > W - 1) x1 = W - 1;
      for (int ch = 0; ch < c; ch++) {
        float a = in[((size_t)y0 * W + x0) * c + ch], b = in[((size_t)y0 * W + x1) * c + ch];
        float d = in[((size_t)y1 * W + x0) * c + ch], e = in[((size_t)y1 * W + x1) * c + ch];
        float top = a + (b - a) * lx, bot = d + (e - d) * lx;
        out[((size_t)y * Wo + x) * c + ch] = top + (bot - top) * ly;
      }
    }
  }
}

static void dense(const float *x, int nin, const float *k, const float *b, int nout, int relu, float *y) {
  for (int o = 0; o < nout; o++) {
    float acc = 0.0f;
    for (int i = 0; i < nin; i++) acc += x[i] * k[(size_t)i * nout + o];
    acc += b[o];
    y[o] = (relu && acc < 0.0f) ? 0.0f : acc;
  }
}

/* One forward.  ship_map / laser_map: uint8 [400][400] ([row=y][col=x]);
 * vec8: the 8-scalar head; weights: the blob above.
 * Outputs: act_values[2], heat[400*400] (may be NULL), iaction, ipointer[2]=(x,y). */
void orc_policy_forward2(const uint8_t *ship_map, const uint8_t *laser_map, const float *vec8, const float *w,
                         float *act_values, float *heat, int32_t *iaction, int32_t *ipointer, int legacy_bilinear) {
  int32_t off[64], cnt[64];
  orc_policy_layout(off, cnt);
  const int S = 400;
  size_t big = (size_t)S * S * 8;
  float *a = (float *)malloc(sizeof(float) * big), *b = (float *)malloc(sizeof(float) * big);
  for (size_t p = 0; p < (size_t)S * S; p++) {
    a[2 * p] = (float)ship_map[p];
    a[2 * p + 1] = (float)laser_map[p];
  }
  int H = S, t = 0;
  for (int i = 0; i < 4; i++) {
    conv3x3(a, H, H, TRUNK_CIN[i], w + off[t], w + off[t + 1], 8, b);
    bn_relu(b, (size_t)H * H, 8, w + off[t + 2], w + off[t + 3], w + off[t + 4], w + off[t + 5]);
    maxpool2(b, H, H, 8, a);
    H /= 2;
    t += 6;
  }
  /* a = pool4 (25,25,8) == Flatten order (h,w,c) */
  float concat[5008], d1[100], d2[50], ud[625];
  memcpy(concat, vec8, 8 * sizeof(float));
  memcpy(concat + 8, a, 5000 * sizeof(float));
  dense(concat, 5008, w + off[t], w + off[t + 1], 100, 1, d1); t += 2;
  dense(d1, 100, w + off[t], w + off[t + 1], 50, 1, d2); t += 2;
  dense(d2, 50, w + off[t], w + off[t + 1], 2, 0, act_values); t += 2;
  dense(d1, 100, w + off[t], w + off[t + 1], 625, 1, ud); t += 2;
  memcpy(a, ud, 625 * sizeof(float));
  H = 25;
  for (int i = 0; i < 4; i++) {
    upsample2(a, H, H, UP_CIN[i], b, legacy_bilinear);
    H *= 2;
    conv3x3(b, H, H, UP_CIN[i], w + off[t], w + off[t + 1], UP_COUT[i], a);
    if (i < 3) {
      bn_relu(a, (size_t)H * H, UP_COUT[i], w + off[t + 2], w + off[t + 3], w + off[t + 4], w + off[t + 5]);
      t += 6;
    } else {
      t += 2;
    }
  }
  /* a = ptr_values (400,400) */
  *iaction = act_values[1] > act_values[0] ? 1 : 0; /* np.argmax: first maximum */
  size_t best = 0;
  for (size_t p = 1; p < (size_t)S * S; p++)
    if (a[p] > a[best]) best = p;
  ipointer[0] = (int32_t)(best % S); /* unravel_index(order='F') on a C-order flat index */
  ipointer[1] = (int32_t)(best / S);
  if (heat) memcpy(heat, a, sizeof(float) * (size_t)S * S);
  free(a);
  free(b);
}

/* the declared default convention (half-pixel centres) */
void orc_policy_forward(const uint8_t *ship_map, const uint8_t *laser_map, const float *vec8, const float *w,
                        float *act_values, float *heat, int32_t *iaction, int32_t *ipointer) {
  orc_policy_forward2(ship_map, laser_map, vec8, w, act_values, heat, iaction, ipointer, 0);
}
