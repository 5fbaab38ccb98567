// ofx_raster.hip - Observation.analyse_battleground for N arenas
// (lib/observation.py:79-95): two fresh zero maps per arena, a disc per playable
// ship (ship_map) and per listed laser, destroyed-this-tick included
// (laser_map), drawn by Circle.binary_draw -> skimage.draw.disk((y, x), r,
// shape) (lib/form.py:222-228).
//
// Disc arithmetic restated from scikit-image 0.18.3 draw.py:11-43,46-143 with
// rotation = 0 (the only value disk() passes): for the clipped bounding box
//   ul = max(ceil(c - r), 0), lr = min(floor(c + r), shape - 1), sc = c - ul,
// pixel (i, j) of the box is set iff ((i - sc_r)/r)^2 + ((j - sc_c)/r)^2 < 1
// in fp64 (numpy squares with x*x; the sin(0)=0 terms add exact zeros).
//
// Mapping: one 256-thread workgroup per (arena, map).  The map is built as a
// 1-bit-per-cell image in LDS (W*H/8 bytes = 20 KB at 400x400): zero, barrier,
// (entity, box-row) tasks OR their row spans in with LDS atomics, barrier, then
// the workgroup streams the image out expanding bits to the requested element
// type with 16-byte stores (1 KiB per wave instruction, fully coalesced).  HBM
// sees each output byte exactly once and no read-modify-write; the kernel is
// bound by HBM write bandwidth (algorithmic bytes = N * 2 * W*H * sizeof(elem)).
#include "ofx_internal.h"

struct RasterParams {
  int N, M, L, W, H;   // map is [W rows = y][H cols = x]  (np.zeros((dim.x, dim.y)))
  int ship_radius, laser_radius;
  ofx_state st;
  void *out[2];        // [which] base pointers, arena stride = map bytes
  int words;           // W*H/32
};

__device__ inline void draw_row(unsigned *bm, int rows, int cols, double cr, double cc, double rad, int box_row) {
  // bounding box (clipped) - recomputed per task, a handful of fp64 ops
  long ulr = (long)ceil(cr - rad), ulc = (long)ceil(cc - rad);
  long lrr = (long)floor(cr + rad), lrc = (long)floor(cc + rad);
  if (ulr < 0) ulr = 0;
  if (ulc < 0) ulc = 0;
  if (lrr > rows - 1) lrr = rows - 1;
  if (lrc > cols - 1) lrc = cols - 1;
  const long nr = lrr - ulr + 1, nc = lrc - ulc + 1;
  if (box_row >= nr || nc <= 0) return;
  const double sc_r = cr - (double)ulr, sc_c = cc - (double)ulc;
  const double t1 = ((double)box_row - sc_r) / rad;
  const double t1sq = t1 * t1;
  const long p0 = (ulr + box_row) * (long)cols + ulc;  // cell index of the span start
  unsigned acc = 0;
  long cur_word = p0 >> 5;
  for (long j = 0; j < nc; j++) {
    const double t2 = ((double)j - sc_c) / rad;  // sign is irrelevant once squared
    const double d = t1sq + t2 * t2;
    const long pcell = p0 + j;
    const long w = pcell >> 5;
    if (w != cur_word) {
      if (acc) atomicOr(&bm[cur_word], acc);
      acc = 0;
      cur_word = w;
    }
    if (d < 1.0) acc |= 1u << (pcell & 31);
  }
  if (acc) atomicOr(&bm[cur_word], acc);
}

template <int FMT>
__global__ __launch_bounds__(256) void k_raster(RasterParams p) {
  extern __shared__ __align__(16) unsigned bm[];
  const int a = blockIdx.x >> 1, which = blockIdx.x & 1;
  const int tid = threadIdx.x;
  // ---- zero the LDS bit image (16 B per lane per iteration) ----
  {
    uint4 z = make_uint4(0, 0, 0, 0);
    uint4 *b4 = reinterpret_cast<uint4 *>(bm);
    const int n4 = (p.words + 3) >> 2;
    for (int i = tid; i < n4; i += 256) b4[i] = z;
  }
  __syncthreads();
  // ---- draw ----
  const int rows = p.W, cols = p.H;
  if (which == 0) {
    const int span = 2 * p.ship_radius + 1;
    const int tasks = p.M * span;
    for (int t = tid; t < tasks; t += 256) {
      const int e = t / span, br = t - e * span;
      const size_t si = (size_t)a * p.M + e;
      if (p.st.alive[si])  // only playable ships are drawn (observation.py:88)
        draw_row(bm, rows, cols, (double)p.st.ship_y[si], (double)p.st.ship_x[si], (double)p.ship_radius, br);
    }
  } else {
    const int n = p.st.n_lasers[a];
    const int span = 2 * p.laser_radius + 1;
    const int tasks = n * span;
    for (int t = tid; t < tasks; t += 256) {
      const int e = t / span, br = t - e * span;
      const size_t li = (size_t)a * p.L + e;  // no state filter (observation.py:92-93)
      draw_row(bm, rows, cols, p.st.laser_y[li], p.st.laser_x[li], (double)p.laser_radius, br);
    }
  }
  __syncthreads();
  // ---- stream out ----
  const size_t cells = (size_t)p.W * p.H;
  if (FMT == OFX_MAP_U8) {
    uint4 *o = reinterpret_cast<uint4 *>(static_cast<uint8_t *>(p.out[which]) + (size_t)a * cells);
    const int groups = (int)(cells >> 4);  // 16 cells -> 16 bytes
    for (int g = tid; g < groups; g += 256) {
      const unsigned bits = (bm[g >> 1] >> ((g & 1) * 16)) & 0xFFFFu;
      uint4 v;  // nibble b3b2b1b0 -> bytes b0,b1,b2,b3 (little endian = ascending cell index)
      v.x = ((bits & 0xFu) * 0x00204081u) & 0x01010101u;
      v.y = (((bits >> 4) & 0xFu) * 0x00204081u) & 0x01010101u;
      v.z = (((bits >> 8) & 0xFu) * 0x00204081u) & 0x01010101u;
      v.w = (((bits >> 12) & 0xFu) * 0x00204081u) & 0x01010101u;
      o[g] = v;
    }
  } else if (FMT == OFX_MAP_F32) {
    uint4 *o = reinterpret_cast<uint4 *>(static_cast<float *>(p.out[which]) + (size_t)a * cells);
    const int groups = (int)(cells >> 2);  // 4 cells -> 16 bytes
    for (int g = tid; g < groups; g += 256) {
      const unsigned bits = bm[g >> 3] >> ((g & 7) * 4);
      uint4 v;
      v.x = (bits & 1u) ? 0x3F800000u : 0u;
      v.y = (bits & 2u) ? 0x3F800000u : 0u;
      v.z = (bits & 4u) ? 0x3F800000u : 0u;
      v.w = (bits & 8u) ? 0x3F800000u : 0u;
      o[g] = v;
    }
  } else if (FMT == OFX_MAP_F64) {
    uint4 *o = reinterpret_cast<uint4 *>(static_cast<double *>(p.out[which]) + (size_t)a * cells);
    const int groups = (int)(cells >> 1);  // 2 cells -> 16 bytes
    for (int g = tid; g < groups; g += 256) {
      const unsigned bits = bm[g >> 4] >> ((g & 15) * 2);
      uint4 v;  // 1.0 = 0x3FF0000000000000
      v.x = 0u; v.y = (bits & 1u) ? 0x3FF00000u : 0u;
      v.z = 0u; v.w = (bits & 2u) ? 0x3FF00000u : 0u;
      o[g] = v;
    }
  } else {  // OFX_MAP_BITS (numpy.packbits: MSB first per byte) / OFX_MAP_BITS_LSB (raw)
    unsigned *o = reinterpret_cast<unsigned *>(static_cast<uint8_t *>(p.out[which]) + (size_t)a * (cells >> 3));
    for (int w = tid; w < p.words; w += 256) {
      const unsigned v = bm[w];
      o[w] = (FMT == OFX_MAP_BITS) ? __builtin_bswap32(__brev(v)) : v;
    }
  }
}

int ofx_launch_raster(ofx_handle *h, int map_type, void *ship_map, void *laser_map) {
  const ofx_config &c = h->cfg;
  const size_t per_map = ofx_map_bytes(h, map_type);
  if (!ship_map) {
    for (int w = 0; w < 2; w++) {
      if (!h->maps[map_type][w]) {
        OFX_HIP(hipMalloc(&h->maps[map_type][w], per_map * (size_t)c.n_arenas));
      }
    }
    ship_map = h->maps[map_type][0];
    laser_map = h->maps[map_type][1];
  }
  RasterParams p;
  p.N = c.n_arenas; p.M = c.n_ships; p.L = c.laser_cap; p.W = c.width; p.H = c.height;
  p.ship_radius = c.ship_radius; p.laser_radius = c.laser_radius;
  p.st = h->st;
  p.out[0] = ship_map; p.out[1] = laser_map;
  p.words = (int)(((size_t)c.width * c.height) >> 5);
  const size_t lds = ((size_t)p.words * 4 + 15) & ~(size_t)15;
  const dim3 grid(2 * c.n_arenas), block(256);
  switch (map_type) {
    case OFX_MAP_U8: hipLaunchKernelGGL(k_raster<OFX_MAP_U8>, grid, block, lds, h->stream, p); break;
    case OFX_MAP_F32: hipLaunchKernelGGL(k_raster<OFX_MAP_F32>, grid, block, lds, h->stream, p); break;
    case OFX_MAP_F64: hipLaunchKernelGGL(k_raster<OFX_MAP_F64>, grid, block, lds, h->stream, p); break;
    case OFX_MAP_BITS: hipLaunchKernelGGL(k_raster<OFX_MAP_BITS>, grid, block, lds, h->stream, p); break;
    default: hipLaunchKernelGGL(k_raster<OFX_MAP_BITS_LSB>, grid, block, lds, h->stream, p); break;
  }
  OFX_HIP(hipGetLastError());
  return OFX_OK;
}
