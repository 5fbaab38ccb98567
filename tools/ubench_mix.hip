// Micro-benchmark (diagnostic tool, not part of the product): do the matrix pipe and the VALU of a gfx950 SIMD run
// concurrently?  Each wave issues NM MFMAs and NV VALU FMAs per loop body, interleaved; the report gives the rate
// of either pipe and the sum.
//   hipcc -O3 --offload-arch=gfx950 -o ubench_mix tools/ubench_mix.hip && ./ubench_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MF 0: v_mfma_f32_4x4x1_16b_f32 cbsz:4 ; 1: v_mfma_f32_16x16x4_f32
// VF 0: v_fmac_f32 VGPR weight ; 1: v_fmac_f32 SGPR weight ; 2: v_pk_fma_f32 (VGPR)
template <int MF, int VF, int NM, int NV>
__global__ __launch_bounds__(256) void k(float *out, const float *w, int iters) {
  const float s0 = w[0];
  const float x = (float)threadIdx.x * 1e-3f, v0 = s0 + x;
  f32x4 m[8];
  float a[16];
  f32x2 pa[8], xv = {x, x + 1}, wv = {v0, v0};
  for (int i = 0; i < 8; i++) { m[i] = (f32x4){x, x + i, x - i, x}; pa[i] = (f32x2){x + i, x - i}; }
  for (int i = 0; i < 16; i++) a[i] = x + i;
  constexpr int PER = NM ? (NV + NM - 1) / NM : NV;
  for (int it = 0; it < iters; it++) {
    int vi = 0;
#pragma unroll
    for (int i = 0; i < (NM ? NM : 1); i++) {
      if (NM) {
        if (MF == 0) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:4" : "+v"(m[i & 7]) : "v"(x), "v"(v0));
        else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(m[i & 7]) : "v"(x), "v"(v0));
      }
#pragma unroll
      for (int j = 0; j < PER; j++, vi++) {
        if (vi >= NV) break;
        if (VF == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[vi & 15]) : "v"(v0), "v"(x));
        else if (VF == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[vi & 15]) : "s"(s0), "v"(x));
        else asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pa[vi & 7]) : "v"(wv), "v"(xv));
      }
    }
  }
  float sum = 0;
  for (int i = 0; i < 8; i++) sum += m[i][0] + m[i][1] + m[i][2] + m[i][3] + pa[i].x + pa[i].y;
  for (int i = 0; i < 16; i++) sum += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int MF, int VF, int NM, int NV>
void run(int blocks_per_cu) {
  const int iters = 2000;
  float *out, *w;
  const int blocks = 256 * blocks_per_cu;
  CHECK(hipMalloc(&out, sizeof(float) * blocks * 256));
  CHECK(hipMalloc(&w, 64));
  CHECK(hipMemset(w, 0, 64));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MF, VF, NM, NV>), dim3(blocks), dim3(256), 0, 0, out, w, iters);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MF, VF, NM, NV>), dim3(blocks), dim3(256), 0, 0, out, w, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double waves = (double)blocks * 4 * iters;
  const double mflop = waves * NM * (MF == 0 ? 16.0 * 4 * 4 * 2 : 16.0 * 16 * 4 * 2);
  const double vflop = waves * NV * (VF == 2 ? 256.0 : 128.0);
  static const char *mn[] = {"4x4x1_16b", "16x16x4"}, *vn[] = {"fmac(vgpr)", "fmac(sgpr)", "pk_fma"};
  printf("%-10s x%-2d + %-10s x%-2d  waves/SIMD=%d  %7.3f ms  mfma %6.1f  valu %6.1f  sum %6.1f TFLOP/s\n", mn[MF], NM,
         vn[VF], NV, blocks_per_cu, ms, mflop / ms / 1e9, vflop / ms / 1e9, (mflop + vflop) / ms / 1e9);
  CHECK(hipFree(out)); CHECK(hipFree(w));
}

int main() {
  for (int b = 1; b <= 4; b *= 2) {
    run<0, 0, 8, 0>(b);
    run<0, 0, 8, 8>(b);
    run<0, 0, 8, 16>(b);
    run<0, 0, 8, 32>(b);
    run<0, 1, 8, 16>(b);
    run<0, 1, 8, 32>(b);
    run<0, 2, 8, 8>(b);
    run<0, 2, 8, 16>(b);
    run<1, 0, 8, 0>(b);
    run<1, 0, 8, 8>(b);
    run<1, 0, 8, 16>(b);
    run<1, 0, 8, 24>(b);
    run<1, 0, 8, 32>(b);
    run<1, 0, 8, 40>(b);
    run<1, 0, 8, 48>(b);
    run<1, 0, 8, 64>(b);
    run<1, 1, 8, 40>(b);
    run<1, 1, 8, 32>(b);
    run<1, 1, 8, 64>(b);
    run<1, 2, 8, 16>(b);
    run<1, 2, 8, 32>(b);
    run<0, 0, 0, 32>(b);
    run<0, 1, 0, 32>(b);
    run<0, 2, 0, 32>(b);
  }
  return 0;
}
