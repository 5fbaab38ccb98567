// Micro-benchmark (diagnostic tool, not part of the product): issue rate of fp32 FMA forms on gfx950.
//   hipcc -O3 --offload-arch=gfx950 -o ubench_fma tools/ubench_fma.hip && ./ubench_fma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE 0: v_fmac_f32 with an SGPR weight, 16 independent accumulators
// MODE 1: v_pk_fma_f32, 8 independent packed accumulators
// MODE 2: v_mfma_f32_32x32x2_f32, 2 independent accumulators
// MODE 3: v_fmac_f32 with VGPR weight
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, const float *w, int iters) {
  const float s0 = w[0], s1 = w[1], s2 = w[2], s3 = w[3];  // uniform -> SGPRs
  const float x = (float)threadIdx.x * 1e-3f;
  if (MODE == 0 || MODE == 3) {
    float a[16];
    for (int i = 0; i < 16; i++) a[i] = x + i;
    const float v0 = MODE == 3 ? s0 + x : 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
          if (MODE == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(s0), "v"(x));
          else asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(v0), "v"(x));
        }
      }
    }
    float sum = 0;
    for (int i = 0; i < 16; i++) sum += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  } else if (MODE == 1) {
    f32x2 a[8], xv = {x, x + 1}, wv = {s0, s1};
    for (int i = 0; i < 8; i++) a[i] = (f32x2){x + i, x - i};
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(wv), "v"(xv));
      }
    }
    float sum = 0;
    for (int i = 0; i < 8; i++) sum += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  } else if (MODE == 4 || MODE == 5) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 a[8];
    for (int i = 0; i < 8; i++) a[i] = (f32x4){x, x + i, x - i, x};
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 2; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
          if (MODE == 4) a[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, s2 + x, a[i], 0, 0, 0);
          else a[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, s3 + x, a[i], 4, 0, 0);
        }
      }
    }
    float sum = 0;
    for (int i = 0; i < 8; i++) sum += a[i][0] + a[i][1] + a[i][2] + a[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  } else {
    f32x16 a0, a1;
    for (int i = 0; i < 16; i++) { a0[i] = x; a1[i] = x + 1; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 2; r++) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, s2 + x, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, s3 + x, a1, 0, 0, 0);
      }
    }
    float sum = 0;
    for (int i = 0; i < 16; i++) sum += a0[i] + a1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  }
}

template <int MODE>
void run(const char *name, int blocks_per_cu, int iters, double flop_per_thread_iter) {
  float *out, *w;
  const int blocks = 256 * blocks_per_cu;
  CHECK(hipMalloc(&out, sizeof(float) * blocks * 256));
  CHECK(hipMalloc(&w, 64));
  CHECK(hipMemset(w, 0, 64));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, w, iters);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, w, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)blocks * 256 * iters * flop_per_thread_iter;
  printf("%-28s waves/SIMD=%d  %8.3f ms  %8.1f TFLOP/s\n", name, blocks_per_cu, ms, flops / ms / 1e9);
  CHECK(hipFree(out)); CHECK(hipFree(w));
}

int main() {
  for (int b = 1; b <= 8; b *= 2) {
    run<0>("v_fmac_f32 (SGPR weight)", b, 4000, 64 * 2.0);
    run<3>("v_fmac_f32 (VGPR weight)", b, 4000, 64 * 2.0);
    run<1>("v_pk_fma_f32", b, 4000, 32 * 4.0);
    run<2>("v_mfma_f32_32x32x2_f32", b, 1000, 4 * 32.0 * 32 * 2 * 2 / 64);
    run<4>("v_mfma_f32_16x16x4_f32", b, 1000, 16 * 16.0 * 16 * 4 * 2 / 64);
    run<5>("v_mfma_f32_4x4x1_16B_f32", b, 1000, 16 * 16.0 * 4 * 4 * 1 * 2 / 64);
  }
  return 0;
}
