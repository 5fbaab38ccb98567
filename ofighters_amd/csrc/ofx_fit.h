// ofx_fit.h - launchers of the lean fit's convolution kernels (ofx_fit.hip), called by ofx_train.hip's orchestration
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "../../include/ofx.h"

// how a convolution's input is produced from what is kept in HBM
#define OFX_FIT_SRC_BITS 0   /* 1-bit maps -> 0.f / 1.f */
#define OFX_FIT_SRC_POOL 1   /* pool2(relu(bn(z_prev)))          (trunk) */
#define OFX_FIT_SRC_UP 2     /* up2(relu(bn(z_prev)))            (head 2) */
#define OFX_FIT_SRC_UPRAW 3  /* up2(u0), u0 already behind its ReLU */
#define OFX_FIT_SRC_PLANE 5  /* a stored plane as it is, zero outside (the pooled activation kept by the forward) */
#define OFX_FIT_SRC_ACTREP 4 /* relu(bn(z_prev)) at its own resolution, edge cells repeated outwards (phase form) */

#ifndef OFX_FIT_MAX_BLOCKS
#define OFX_FIT_MAX_BLOCKS 2048 /* persistent grids: at most this many blocks, each with one row of partial sums */
#endif

struct ofx_fit_src {
  int kind;
  float *keep;        // POOL in ofx_fit_conv_fwd: also store the pooled activation [n][C][H][W] here (null: do not)
  const void *p;      // bits [n][C][5000] or the producing layer's z [n][C][h][w]
  const float *act;   // the producing layer's {scale, shift} per channel; null for bits / raw; PLANE: 16 bytes of zeros in
                      // device memory (what an LDS-direct load reads for a cell outside the plane)
  int h, w;           // dims of p's planes
  int legacy;         // OFX_OPT_BILINEAR_LEGACY
};

size_t ofx_fit_part_doubles(void);  // size of the partial-sum buffer every launcher below may use
// z = conv3x3(src) + b; part != null: the block sums of z, z*z per output channel (part[block][2 co]), *nblocks rows
int ofx_fit_conv_fwd(hipStream_t st, int n, int ci, int co, int H, int W, const ofx_fit_src &src, const float *w,
                     const float *b, float *z, double *part, int *nblocks);
// ordered combine of part[nblocks][2 c] -> sums[2 c] (may be null); with stat != null also {mean, var} and {scale, shift}
int ofx_fit_finish(hipStream_t st, int nblocks, int c_n, double count, const double *part, const float *gamma,
                   const float *beta, double *sums, float *stat, float *act);
int ofx_fit_pool_act(hipStream_t st, int n, int C, int H, int W, const float *z, const float *act, float *p);
// g = d loss / d (BN output of the layer), ReLU-masked, through the pooling in front of the next convolution
// (conv: dzn = that convolution's dz [n][8][H/2][W/2], wn its kernel; else dzn = d pooled output)
// wtr: 576 floats of scratch (the kernel transposed for the scalar cache); zero: 16 bytes of zeros (LDS-direct loads)
int ofx_fit_b1_pool(hipStream_t st, int n, int H, int W, int conv, const float *dzn, const float *wn, const float *z,
                    const float *stat, const float *act, float *g, double *part, int *nblocks, float *wtr,
                    const float *zero);
size_t ofx_fit_first_doubles(int n);
size_t ofx_fit_first_floats(void);
struct ofx_handle;
// the first trunk layer without its tensor z0 (see ofx_fit.hip "the first layer is never materialised")
int ofx_fit_first_fwd(ofx_handle *h, int n, const void *bits, const float *w, const float *b, const float *gamma,
                      const float *beta, double *cpart, float *stat, float *act, float *luts, float *p0);
// the backward of the first TWO layers (ofx_fit.hip "... over the windows that see a set bit"): g1 / sums1 from
// ofx_fit_b1_pool + ofx_fit_finish of the second layer; its dz is never stored.  part: ofx_fit_first_part_doubles(n) doubles
size_t ofx_fit_first_part_doubles(int n);
int ofx_fit_first_bwd(hipStream_t st, int n, const void *bits, const float *g1, const float *z1, const float *stat1,
                      const float *gamma1, const double *sums1, const float *wn, const float *p0, const float *luts,
                      const float *w, const float *b, const float *stat, const float *gamma, const float *beta, double *part,
                      double *cpart, float *dw, float *db, float *dgamma, float *dbeta, float *dw2, float *db2, float *dgamma2,
                      float *dbeta2);
// the same through the x2 up-sampling in front of a convolution with `con` output channels at 2h x 2w; zero: 16 bytes of
// zeros in device memory (what an LDS-direct load reads for a cell outside the plane); wtr: 576 floats of scratch
int ofx_fit_b1_up(hipStream_t st, int n, int c, int con, int h, int w, int bn, const float *dzn, const float *wn,
                  const float *zp, const float *stat, const float *act, int legacy, float *g, double *part, int *nblocks,
                  const float *zero, float *wtr);
// dz over g in place (bn), dw / db (/ dgamma, dbeta from sums); gpatch + rows (the last head layer under the textbook
// targets): g is not read - it is zero but for the 4 x 4 patch per sample and channel ofx_fit_top_point left in gpatch
int ofx_fit_bw(hipStream_t st, int n, int ci, int co, int H, int W, const ofx_fit_src &src, int bn, float *g,
               const float *z, const float *stat, const float *gamma, const double *sums, double *part, float *dw,
               float *db, float *dgamma, float *dbeta, const float *gpatch = nullptr, const ofx_transition *rows = nullptr);
// the output convolution (8 -> 1 at 400 x 400 behind the last x2 up-sampling) in phase form: forward + frame correction;
// weff: ofx_fit_out_floats() floats of scratch; the weight gradient needs part (ofx_fit_part_doubles) and fpart
// (ofx_fit_out_doubles(n) doubles)
size_t ofx_fit_out_floats(void);
size_t ofx_fit_out_doubles(int n);
int ofx_fit_out_fwd(hipStream_t st, int n, const ofx_fit_src &src, const float *w, const float *b, float *o2, float *weff);
int ofx_fit_out_bw(hipStream_t st, int n, const ofx_fit_src &src, const float *d2, double *part, double *fpart, float *dw,
                   float *db);
// the top of head 2 when d o2 has one non-zero per sample (ofx_dqn_fit's targets): see ofx_fit.hip
size_t ofx_fit_point_doubles(int n);
int ofx_fit_top_point(hipStream_t st, int n, const ofx_transition *rows, const ofx_fit_src &src, const float *w, const float *b,
                      const float *o1, const float *y_act, const float *y_ptr, const float *stat, float *o2p, float *do1,
                      float *d2p, float *lpart, float *gpatch, double *scratch, double *sums, float *dw, float *db);
