// Micro-benchmark (diagnostic tool, not part of the product): what do waves of DIFFERENT kinds cost each other when they
// share a gfx950 SIMD?  tools/ubench_mix.hip interleaves matrix and vector instructions inside every wave; this one
// gives each wave ONE kind of work and places the kinds side by side on a SIMD - the question behind a producer /
// consumer split of k_trunk12 (table phase = ds_read_b128 + v_max3 beside the 16x16x4 GEMM phase) and behind
// k_head_stream's schedule (4x4x1 consumers beside 16x16x4 producers).
//
// A 1024-thread workgroup has 16 waves; the waves of a workgroup go to the SIMDs in the cyclic order 0 2 1 3, so waves
// w, w + 4, w + 8, w + 12 share a SIMD: "slot" k = w / 4 is the k-th wave of its SIMD.  Every slot gets a role; every wave
// repeats its role's loop body for the same WINDOW of cycles (s_memtime) and counts the bodies.  One workgroup per CU.
//   hipcc -O3 --offload-arch=gfx950 -o ubench_coexec tools/ubench_coexec.hip && ./ubench_coexec
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum Role { IDLE = 0, M16 = 1, M4 = 2, VF = 3, LT = 4, LR = 5, M16V = 6, PK = 7, PKS = 8, PKC = 9 };
static const char *role_name[] = {"-", "mfma16x16x4", "mfma4x4x1", "v_fmac", "ds_b128+max3", "ds_read_b64", "16x16x4+relu/lds", "v_pk_fma", "v_pk_fma sgpr/bc", "pk_fma+lds (C)"};
// instructions per loop body (the unit the report divides by)
static const int role_n[] = {1, 16, 64, 128, 32, 64, 16, 128, 128, 96};

template <int ROLE>
__device__ __forceinline__ void body(f32x4 *m, float *a, float x, float v0, const float *lds, unsigned laddr) {
  if (ROLE == M16) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(m[i & 3]) : "v"(x), "v"(v0));
  } else if (ROLE == M4) {
#pragma unroll
    for (int i = 0; i < 64; i++) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:4" : "+v"(m[i & 3]) : "v"(x), "v"(v0));
  } else if (ROLE == VF) {
#pragma unroll
    for (int i = 0; i < 128; i++) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i & 15]) : "v"(v0), "v"(x));
  } else if (ROLE == LT) {  // k_trunk12's table phase in miniature: 16-byte LDS reads + v_max3 on them
#pragma unroll
    for (int i = 0; i < 32; i++) {
      f32x4 t;
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(laddr), "n"(16 * (i & 15)));
      asm volatile("s_waitcnt lgkmcnt(8)");
      asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(a[i & 15]) : "v"(x), "v"(v0));
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
  } else if (ROLE == LR) {  // stage C's reads alone
#pragma unroll
    for (int i = 0; i < 64; i++) {
      float2 t;
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(t) : "v"(laddr), "n"(8 * (i & 31)));
      if ((i & 7) == 7) asm volatile("s_waitcnt lgkmcnt(8)");
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
  } else if (ROLE == PK) {   // packed f32 FMA, all operands in registers
    float2 *a2 = reinterpret_cast<float2 *>(a);
    const float2 xx = {x, v0};
#pragma unroll
    for (int i = 0; i < 128; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a2[i & 7]) : "v"(xx), "v"(xx));
  } else if (ROLE == PKS) {  // the stage-C form on the vector pipe: weights = scalar pair, pixel value broadcast by op_sel
    float2 *a2 = reinterpret_cast<float2 *>(a);
    const float2 xx = {x, v0};
    const float2 ws = {__builtin_amdgcn_readfirstlane(x), __builtin_amdgcn_readfirstlane(v0)};
#pragma unroll
    for (int i = 0; i < 128; i++) {
      if (i & 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(a2[i & 7]) : "s"(ws), "v"(xx));
      else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(a2[i & 7]) : "s"(ws), "v"(xx));
    }
  } else if (ROLE == PKC) {  // a stage-C pass on the vector pipe in miniature: 4 steps of (4 ds_read_b64, 24 v_pk_fma_f32)
    float2 *a2 = reinterpret_cast<float2 *>(a);
    const float2 ws = {__builtin_amdgcn_readfirstlane(x), __builtin_amdgcn_readfirstlane(v0)};
    float2 t[2][4];
#pragma unroll
    for (int q = 0; q < 4; q++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(t[0][q]) : "v"(laddr), "n"(8 * q));
#pragma unroll
    for (int st = 0; st < 4; st++) {
#pragma unroll
      for (int q = 0; q < 4; q++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(t[(st + 1) & 1][q]) : "v"(laddr), "n"(8 * q + 64 * (st + 1)));
      asm volatile("s_waitcnt lgkmcnt(4)");
#pragma unroll
      for (int i = 0; i < 24; i++) {
        if (i & 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(a2[i & 7]) : "s"(ws), "v"(t[st & 1][(i >> 1) & 3]));
        else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(a2[i & 7]) : "s"(ws), "v"(t[st & 1][(i >> 1) & 3]));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
  } else if (ROLE == M16V) {  // stage B in miniature: per 16x16x4 one v_med3 + half a ds_write_b32 + half a ds_read_b32
#pragma unroll
    for (int i = 0; i < 16; i++) {
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(m[i & 3]) : "v"(x), "v"(v0));
      asm volatile("v_med3_f32 %0, %1, %2, %0" : "+v"(a[i & 15]) : "v"(x), "v"(v0));
      if (i & 1) asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(laddr), "v"(a[i & 15]), "n"(4 * (i & 15)));
      else { float t; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(t) : "v"(laddr), "n"(4 * (i & 15))); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
  }
}

template <int R0, int R1, int R2, int R3>
__global__ __launch_bounds__(1024) void k(float *out, unsigned long long *span, unsigned *count, const float *w, int iters) {
  __shared__ __align__(16) float lds[16 * 1024];
  const int wave = threadIdx.x >> 6, slot = wave >> 2;
  const float s0 = w[0];
  const float x = (float)threadIdx.x * 1e-3f, v0 = s0 + x;
  for (int i = threadIdx.x; i < 16 * 1024; i += 1024) lds[i] = x + i;
  f32x4 m[4];
  float a[16];
  for (int i = 0; i < 4; i++) m[i] = (f32x4){x, x + i, x - i, x};
  for (int i = 0; i < 16; i++) a[i] = x + i;
  const unsigned laddr = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds + (threadIdx.x & 63) * 16 + wave * 1024;
  __syncthreads();
  // every wave works for the same WINDOW of cycles and counts its loop bodies: rates of streams that really run side by
  // side (r03: fixed-length streams let a short stream finish and the long one run alone - the spans then say nothing
  // about co-execution)
  const unsigned long long window = (unsigned long long)iters * 1000ull;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long t1 = t0;
  unsigned n_done = 0;
#define OFX_RUN(R) do { for (int r_ = 0; r_ < 4; r_++) body<R>(m, a, x, v0, lds, laddr); n_done += 4; t1 = __builtin_amdgcn_s_memtime(); } while (t1 - t0 < window)
  if (slot == 0) { OFX_RUN(R0); }
  else if (slot == 1) { OFX_RUN(R1); }
  else if (slot == 2) { OFX_RUN(R2); }
  else { OFX_RUN(R3); }
#undef OFX_RUN
  float sum = 0;
  for (int i = 0; i < 4; i++) sum += m[i][0] + m[i][1] + m[i][2] + m[i][3];
  for (int i = 0; i < 16; i++) sum += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if ((threadIdx.x & 63) == 0) { span[blockIdx.x * 16 + wave] = t1 - t0; count[blockIdx.x * 16 + wave] = n_done; }
}

template <int R0, int R1, int R2, int R3>
void run(const char *what) {
  const int iters = 2000, blocks = 256;   // window = iters * 1000 cycles
  float *out, *w;
  unsigned long long *span, hs[256 * 16];
  unsigned *count, hc[256 * 16];
  CHECK(hipMalloc(&count, sizeof(hc)));
  CHECK(hipMalloc(&out, sizeof(float) * blocks * 1024));
  CHECK(hipMalloc(&span, sizeof(hs)));
  CHECK(hipMalloc(&w, 64));
  CHECK(hipMemset(w, 0, 64));
  for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k<R0, R1, R2, R3>), dim3(blocks), dim3(1024), 0, 0, out, span, count, w, iters);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(hs, span, sizeof(hs), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hc, count, sizeof(hc), hipMemcpyDeviceToHost));
  const int roles[4] = {R0, R1, R2, R3};
  printf("%-44s", what);
  for (int s = 0; s < 4; s++) {
    if (roles[s] == IDLE) { printf(" | %-17s %8s", "-", ""); continue; }
    double c = 0, nb = 0;
    for (int b = 0; b < blocks; b++) for (int q = 0; q < 4; q++) { c += (double)hs[b * 16 + 4 * s + q]; nb += (double)hc[b * 16 + 4 * s + q]; }
    c /= nb * role_n[roles[s]];
    printf(" | %-17s %6.2f c", role_name[roles[s]], c);
  }
  printf("\n");
  CHECK(hipFree(out)); CHECK(hipFree(span)); CHECK(hipFree(count)); CHECK(hipFree(w));
}

int main() {
  printf("cycles (s_memtime) per instruction of each slot's own stream; slots share a SIMD\n");
  run<M16, IDLE, IDLE, IDLE>("16x16x4 alone");
  run<M4, IDLE, IDLE, IDLE>("4x4x1 alone");
  run<VF, IDLE, IDLE, IDLE>("v_fmac alone");
  run<LT, IDLE, IDLE, IDLE>("table phase alone");
  run<LR, IDLE, IDLE, IDLE>("ds_read_b64 alone");
  run<M16V, IDLE, IDLE, IDLE>("16x16x4 + relu / lds alone");
  run<M16, M16, IDLE, IDLE>("16x16x4 | 16x16x4");
  run<M4, M4, IDLE, IDLE>("4x4x1 | 4x4x1");
  run<VF, VF, IDLE, IDLE>("v_fmac | v_fmac");
  run<M16, VF, IDLE, IDLE>("16x16x4 | v_fmac      (separate waves)");
  run<M4, VF, IDLE, IDLE>("4x4x1 | v_fmac");
  run<M16, LT, IDLE, IDLE>("16x16x4 | table phase (the trunk question)");
  run<M16, LT, LT, IDLE>("16x16x4 | 2 x table phase");
  run<M16, M16, LT, LT>("2 x 16x16x4 | 2 x table phase");
  run<LT, LT, LT, LT>("4 x table phase");
  run<M16, M16, M16, M16>("4 x 16x16x4");
  run<M16, M4, IDLE, IDLE>("16x16x4 | 4x4x1");
  run<M4, LR, IDLE, IDLE>("4x4x1 | ds_read_b64");
  run<M16V, M4, IDLE, IDLE>("stage-B-like | 4x4x1");
  run<M16V, M4, M16V, M4>("2 x (stage-B-like | 4x4x1)   (k_head_stream)");
  run<M16V, M16V, M4, M4>("same, slots grouped");
  run<M16V, M16V, M16V, M16V>("4 x stage-B-like");
  run<M4, M4, M4, M4>("4 x 4x4x1");
  // r03: can the vector pipe take stage C while the matrix pipe runs stage B?
  run<PK, IDLE, IDLE, IDLE>("v_pk_fma alone");
  run<PKS, IDLE, IDLE, IDLE>("v_pk_fma sgpr / broadcast alone");
  run<PKC, IDLE, IDLE, IDLE>("pk_fma + lds alone");
  run<PKS, PKS, IDLE, IDLE>("pk | pk");
  run<PKS, PKS, PKS, PKS>("4 x pk");
  run<M16, PKS, IDLE, IDLE>("16x16x4 | pk");
  run<M4, PKS, IDLE, IDLE>("4x4x1 | pk");
  run<M16, PKS, M16, PKS>("2 x (16x16x4 | pk)");
  run<M16V, PKC, IDLE, IDLE>("stage-B-like | pk_fma + lds");
  run<M16V, PKC, M16V, PKC>("2 x (stage-B-like | pk_fma + lds)");
  run<M16V, M16V, PKC, PKC>("same, slots grouped");
  run<M16, PKC, PKC, PKC>("16x16x4 | 3 x (pk_fma + lds)");
  run<M16V, PKC, PKC, PKC>("stage-B-like | 3 x (pk_fma + lds)");
  return 0;
}
