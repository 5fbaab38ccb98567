// Micro-benchmark (diagnostic, not product): how close do k_head_stream's two inner loops get to the matrix pipe's rate
// when ALL 16 waves of a CU run the same one (a phase-separated schedule), against the producer / consumer mix?
//   mode 0: stage C pass - 144 v_mfma_f32_4x4x1 (cbsz 4) + 48 8-byte LDS reads, explicit 3-deep pipeline
//   mode 1: stage B tile pair - 36 v_mfma_f32_16x16x4 + 18 LDS gathers + 16 LDS writes
//   hipcc -O3 --offload-arch=gfx950 -o ubench_phase tools/ubench_phase.hip && ./ubench_phase
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const volatile __attribute__((address_space(3))) f32x2 lds_v2;
constexpr int P3 = 212, PL3 = 16 * P3 + 4;   // ring row pitch / channel plane stride (floats)

template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float *out, const float *w, int iters) {
  extern __shared__ __align__(16) float sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int e = tid; e < 8 * PL3; e += THREADS) sm[e] = (float)(e & 255) * 1e-3f;
  __syncthreads();
  float sum = 0.f;
  if (MODE == 0) {
    float wreg[5];
    for (int r = 0; r < 5; r++) wreg[r] = w[r * 64 + lane];
    const int task = tid % 1000, r_in = task / 100, jx = task % 100;
    unsigned q[3];
    for (int dy = 0; dy < 3; dy++) q[dy] = (unsigned)(((r_in + dy) & 15) * P3 + 2 * jx) * 4;
    const char *u3b = reinterpret_cast<const char *>(sm);
    for (int it = 0; it < iters; it++) {
      f32x4 acc[2][2];
      for (int px = 0; px < 2; px++) { acc[px][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[px][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      f32x2 V[3][2][2];
      auto ldv = [&](int t) {
        const int c = t / 3, dy = t % 3;
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
          const char *qq = u3b + q[dy] + (c + 4 * hf) * (PL3 * 4);
          V[t % 3][hf][0] = *(lds_v2 *)(qq);
          V[t % 3][hf][1] = *(lds_v2 *)(qq + 8);
        }
      };
      ldv(0); ldv(1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 12; t++) {
        if (t + 2 < 12) ldv(t + 2);
        const int c = t / 3, dy = t % 3;
#pragma unroll
        for (int dx = 0; dx < 3; dx++)
#pragma unroll
          for (int hf = 0; hf < 2; hf++) {
            const int kk = (c + 4 * hf) * 9 + dy * 3 + dx;
#pragma unroll
            for (int px = 0; px < 2; px++) {
              const float v = V[t % 3][hf][(dx + px) >> 1][(dx + px) & 1];
              switch (kk & 15) {
#define CASE(B) case B: acc[px][hf] = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[kk >> 4], v, acc[px][hf], 4, B, 0); break;
                CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15)
#undef CASE
              }
            }
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      const f32x4 o0 = acc[0][0] + acc[0][1], o1 = acc[1][0] + acc[1][1];
      sum += fmaxf(fmaxf(o0[0], o0[1]), fmaxf(o0[2], o0[3])) + fmaxf(fmaxf(o1[0], o1[1]), fmaxf(o1[2], o1[3]));
      for (int dy = 0; dy < 3; dy++) { const unsigned n = q[dy] + 5 * P3 * 4; q[dy] = n >= 16 * P3 * 4 ? n - 16 * P3 * 4 : n; }
      __syncthreads();
    }
  } else {
    const int n16 = lane & 15, kq = lane >> 4;
    float bw[2][9];
    for (int hf = 0; hf < 2; hf++) for (int j = 0; j < 9; j++) bw[hf][j] = w[(hf * 9 + j) * 64 + lane];
    const float *a_lane = &sm[kq * 2000 + (n16 & 3)];
    float *w_lane = &sm[4 * PL3 + (n16 & 3) * PL3 + (n16 >> 2) * 2];
    const int wv = tid >> 6;
    for (int it = 0; it < iters; it++) {
      const float *a0p = a_lane + ((it + wv) & 7) * 108 + 4 * (n16 >> 2), *a1p = a0p + 16;
      float a0[9], a1[9];
#pragma unroll
      for (int j = 0; j < 9; j++) { a0[j] = a0p[(j / 3) * 108 + j % 3]; a1[j] = a1p[(j / 3) * 108 + j % 3]; }
      f32x4 d00 = {0, 0, 0, 0}, d01 = d00, d10 = d00, d11 = d00;
#pragma unroll
      for (int j = 0; j < 9; j++) {
        d00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], bw[0][j], d00, 0, 0, 0);
        d01 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], bw[1][j], d01, 0, 0, 0);
      }
      float *wp = w_lane + ((it + wv) & 7) * P3 + 8 * kq;
#pragma unroll
      for (int j = 0; j < 9; j++) {
        d10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], bw[0][j], d10, 0, 0, 0);
        d11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], bw[1][j], d11, 0, 0, 0);
        if (j == 1) {
#pragma unroll
          for (int i = 0; i < 4; i++) { wp[2 * i] = fmaxf(d00[i], 0.f); wp[2 * PL3 + 2 * i] = fmaxf(d01[i], 0.f); }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; i++) { wp[64 + 2 * i] = fmaxf(d10[i], 0.f); wp[2 * PL3 + 64 + 2 * i] = fmaxf(d11[i], 0.f); }
      if ((it & 1) == 1) __syncthreads();
    }
    sum = sm[tid];
  }
  out[blockIdx.x * THREADS + tid] = sum;
}

template <int MODE, int THREADS>
void run(int wgs_per_cu, const char *name) {
  const int iters = 2000, blocks = 256 * wgs_per_cu;
  float *out, *w;
  CHECK(hipMalloc(&out, sizeof(float) * blocks * THREADS));
  CHECK(hipMalloc(&w, 64 * 64 * 4));
  CHECK(hipMemset(w, 0, 64 * 64 * 4));
  const size_t lds = sizeof(float) * 8 * PL3;
  CHECK(hipFuncSetAttribute((const void *)k<MODE, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE, THREADS>), dim3(blocks), dim3(THREADS), lds, 0, out, w, iters);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE, THREADS>), dim3(blocks), dim3(THREADS), lds, 0, out, w, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double waves = (double)blocks * THREADS / 64, per_simd = waves / 1024;
  // MFMA pipe cycles per wave and iteration: mode 0: 144 x 8 (2 passes) ; mode 1: 36 x 32
  const double cyc = MODE == 0 ? 144.0 * 8 : 36.0 * 32;
  const double ideal_ms = per_simd * iters * cyc / 2.4e6;
  printf("%-44s %7.3f ms  pipe-cycle floor %7.3f ms  = %.0f %% busy\n", name, ms, ideal_ms, 100.0 * ideal_ms / ms);
  CHECK(hipFree(out)); CHECK(hipFree(w));
}

int main() {
  run<0, 1024>(1, "stage C pass, 16 waves / CU in phase");
  run<0, 512>(1, "stage C pass, 8 waves / CU");
  run<0, 256>(1, "stage C pass, 4 waves / CU");
  run<1, 1024>(1, "stage B tile pair, 16 waves / CU in phase");
  run<1, 512>(1, "stage B tile pair, 8 waves / CU");
  run<1, 256>(1, "stage B tile pair, 4 waves / CU");
  return 0;
}
