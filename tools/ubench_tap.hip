// Micro-benchmark (diagnostic, not product): prices the TAP-PLANE form of k_head_stream's stage C against the r03 form, in
// the kernel's own setting - 512-thread workgroups of 4 producer + 4 consumer waves, two per CU, one barrier per
// sub-step - before the kernel is rewritten (r04).
//   mix 0 (r03): producers = stage-B tile pair (36 v_mfma_f32_16x16x4 + 18 LDS gathers + 16 LDS writes of uprelu3),
//                consumers = stage-C pass (144 v_mfma_f32_4x4x1 + 48 8-byte LDS reads of uprelu3)
//   mix 1 (r04): producers = the same tile pair in TRANSPOSED orientation (weights as the A operand: a lane ends up with
//                all 8 channels of ONE uprelu3 pixel) + ReLU + the 1x1 convolution 8 channels -> 9 tap planes
//                V_t = sum_c w4[t][c] uprelu3_c as 24 v_mfma_f32_4x4x1 per tile + 9 LDS writes of V;
//                consumers = the up-sampling stencil on the tap planes on the vector ALU (42 8-byte LDS reads,
//                ~100 v_pk_fma_f32 per pixel pair): heat[2y+a][2x+b] = sum_t up(V_t)[2y+a+ky-1][2x+b+kx-1]
//   hipcc -O3 --offload-arch=gfx950 -o ubench_tap tools/ubench_tap.hip && ./ubench_tap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const volatile __attribute__((address_space(3))) f32x2 lds_v2;
constexpr int P3 = 106, PL3 = 16 * P3 + 4;   // ring row pitch / plane stride (floats)
constexpr int P2 = 56, PL2 = 18 * P2;
constexpr int THREADS = 512;

__device__ __forceinline__ float max_raw(float x, float floor) {
  float pinf;
  asm("s_mov_b32 %0, 0x7f800000" : "=s"(pinf));
  return __builtin_amdgcn_fmed3f(x, floor, pinf);
}
__device__ __forceinline__ f32x2 fma2(float c, f32x2 v, f32x2 z) { return __builtin_elementwise_fma((f32x2){c, c}, v, z); }

template <int MIX>
__global__ __launch_bounds__(THREADS, 4) void k(float *out, const float *w, int iters) {
  __shared__ __align__(16) float ring[(MIX ? 9 : 8) * PL3];
  __shared__ __align__(16) float u2r[4 * PL2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int e = tid; e < (MIX ? 9 : 8) * PL3; e += THREADS) ring[e] = (float)(e & 255) * 1e-3f;
  for (int e = tid; e < 4 * PL2; e += THREADS) u2r[e] = (float)(e & 127) * 1e-3f;
  __syncthreads();
  float sum = 0.f;
  const int n16 = lane & 15, kq = lane >> 4;
  if (wv < 4) {
    // ------------------------------------------------------------------ producers
    float bw[2][9];
    for (int hf = 0; hf < 2; hf++) for (int j = 0; j < 9; j++) bw[hf][j] = w[(hf * 9 + j) * 64 + lane];
    float wq[2];
    wq[0] = w[20 * 64 + lane]; wq[1] = w[21 * 64 + lane];
    const f32x4 bi = {0.1f, 0.2f, 0.3f, 0.4f};
    const float *a_lane = &u2r[kq * PL2 + (n16 & 3)];
    for (int it = 0; it < iters; it++) {
      const int row = (it * 5 + wv) & 15;
      const float *a0p = a_lane + (row & 7) * P2 + 4 * (n16 >> 2), *a1p = a0p + 16;
      float a0[9], a1[9];
#pragma unroll
      for (int j = 0; j < 9; j++) { a0[j] = a0p[(j / 3) * P2 + j % 3]; a1[j] = a1p[(j / 3) * P2 + j % 3]; }
      f32x4 d00, d01, d10, d11;
      if (MIX == 0) {
        float *w_lane = &ring[(n16 & 3) * PL3 + (n16 >> 3) * P3 + ((n16 >> 2) & 1) + 1];
        float *wp = w_lane + row * P3 + 8 * kq;
#pragma unroll
        for (int j = 0; j < 9; j++) {
          d00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], bw[0][j], j ? d00 : bi, 0, 0, 0);
          d01 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], bw[1][j], j ? d01 : bi, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 9; j++) {
          d10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], bw[0][j], j ? d10 : bi, 0, 0, 0);
          d11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], bw[1][j], j ? d11 : bi, 0, 0, 0);
          if (j == 1) {
#pragma unroll
            for (int i = 0; i < 4; i++) { wp[2 * i] = max_raw(d00[i], 0.f); wp[4 * PL3 + 2 * i] = max_raw(d01[i], 0.f); }
          }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) { wp[32 + 2 * i] = max_raw(d10[i], 0.f); wp[4 * PL3 + 32 + 2 * i] = max_raw(d11[i], 0.f); }
      } else {
        // transposed: D[m = phase * 4 + cl][n = quad]: lane (quad n16, phase kq) holds channels 4 hf + i
        const int pa = kq >> 1, pb = kq & 1;
        float *w_lane = &ring[pa * P3 + 2 * (n16 & 3) + pb + 1];
        float *wp0 = w_lane + ((2 * row) & 15) * P3 + 8 * (n16 >> 2), *wp1 = wp0 + 32;
        auto one_by_one = [&](const f32x4 &u0, const f32x4 &u1, float *wp) {
          f32x4 v0 = {0, 0, 0, 0}, v1 = v0, v2 = v0;
#pragma unroll
          for (int co = 0; co < 8; co++) {
            const float b = co < 4 ? u0[co & 3] : u1[co & 3];
#pragma unroll
            for (int g = 0; g < 3; g++) {
              const int idx = g * 8 + co;
              f32x4 &acc = g == 0 ? v0 : g == 1 ? v1 : v2;
              switch (idx & 15) {
#define CASE(B) case B: acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wq[idx >> 4], b, acc, 4, B, 0); break;
                CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15)
#undef CASE
              }
            }
          }
#pragma unroll
          for (int i = 0; i < 4; i++) { wp[i * PL3] = v0[i]; wp[(4 + i) * PL3] = v1[i]; }
          wp[8 * PL3] = v2[0];
        };
#pragma unroll
        for (int j = 0; j < 9; j++) {
          d00 = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[0][j], a0[j], j ? d00 : bi, 0, 0, 0);
          d01 = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[1][j], a0[j], j ? d01 : bi, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 9; j++) {
          d10 = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[0][j], a1[j], j ? d10 : bi, 0, 0, 0);
          d11 = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[1][j], a1[j], j ? d11 : bi, 0, 0, 0);
        }
        f32x4 u00, u01, u10, u11;
#pragma unroll
        for (int i = 0; i < 4; i++) { u00[i] = max_raw(d00[i], 0.f); u01[i] = max_raw(d01[i], 0.f); }
        one_by_one(u00, u01, wp0);
#pragma unroll
        for (int i = 0; i < 4; i++) { u10[i] = max_raw(d10[i], 0.f); u11[i] = max_raw(d11[i], 0.f); }
        one_by_one(u10, u11, wp1);
      }
      __syncthreads();
    }
    sum = ring[tid];
  } else {
    // ------------------------------------------------------------------ consumers
    const int task = min(64 * (wv - 4) + lane, 249), r_in = task / 50, jx = task - 50 * r_in;
    constexpr unsigned PB = P3 * 4, RING = 16 * PB;
    const unsigned lcol = (unsigned)(2 * jx) * 4;
    unsigned q[3];
    for (int dy = 0; dy < 3; dy++) q[dy] = (unsigned)((r_in + dy) & 15) * PB + lcol;
    const unsigned qlim = RING + lcol;
    const char *u3b = reinterpret_cast<const char *>(ring);
    if (MIX == 0) {
      float wreg[5];
      for (int r = 0; r < 5; r++) wreg[r] = w[(30 + r) * 64 + lane];
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
        const float m8 = max_raw(max_raw(max_raw(o0[0], o0[1]), max_raw(o0[2], o0[3])), max_raw(max_raw(o1[0], o1[1]), max_raw(o1[2], o1[3])));
        if (m8 > sum) sum = m8;
        for (int dy = 0; dy < 3; dy++) { const unsigned n = q[dy] + 5 * PB; q[dy] = n >= qlim ? n - RING : n; }
        __syncthreads();
      }
    } else {
      // coefficients of the x2 bilinear (half-pixel): CY[a][ky][ty]; the non-zero pattern is the same for both conventions
      float c75, c25;
      asm("s_mov_b32 %0, 0x3f400000" : "=s"(c75));
      asm("s_mov_b32 %0, 0x3e800000" : "=s"(c25));
      const float CY[2][3][3] = {{{c75, c25, 0.f}, {c25, c75, 0.f}, {0.f, c75, c25}}, {{c25, c75, 0.f}, {0.f, c75, c25}, {0.f, c25, c75}}};
      const bool NZ[2][3][3] = {{{1, 1, 0}, {1, 1, 0}, {0, 1, 1}}, {{1, 1, 0}, {0, 1, 1}, {0, 1, 1}}};
      for (int it = 0; it < iters; it++) {
        f32x2 Z[2][3][2];
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
          f32x2 V[3][3][2];   // [ky][ty][pair]
#pragma unroll
          for (int ky = 0; ky < 3; ky++)
#pragma unroll
            for (int ty = 0; ty < 3; ty++) {
              if (!(NZ[0][ky][ty] || NZ[1][ky][ty])) continue;
              const char *qq = u3b + q[ty] + (ky * 3 + kx) * (PL3 * 4);
              V[ky][ty][0] = *(lds_v2 *)(qq);
              V[ky][ty][1] = *(lds_v2 *)(qq + 8);
            }
#pragma unroll
          for (int a = 0; a < 2; a++)
#pragma unroll
            for (int p = 0; p < 2; p++) {
              f32x2 z = {0.f, 0.f};
              bool first = true;
#pragma unroll
              for (int ky = 0; ky < 3; ky++)
#pragma unroll
                for (int ty = 0; ty < 3; ty++) {
                  if (!NZ[a][ky][ty]) continue;
                  if (first) { z = (f32x2){CY[a][ky][ty], CY[a][ky][ty]} * V[ky][ty][p]; first = false; }
                  else z = fma2(CY[a][ky][ty], V[ky][ty][p], z);
                }
              Z[a][kx][p] = z;
            }
        }
        // horizontal: out[a][b] (pair x0, x0 + 1) = sum_kx sum_tx CY[b][kx][tx] Z[a][kx](x + tx - 1)
        f32x4 o0, o1;
#pragma unroll
        for (int a = 0; a < 2; a++) {
          f32x2 M[3];
#pragma unroll
          for (int kx = 0; kx < 3; kx++) M[kx] = (f32x2){Z[a][kx][0].y, Z[a][kx][1].x};
#pragma unroll
          for (int b = 0; b < 2; b++) {
            f32x2 o = {0.5f, 0.5f};
#pragma unroll
            for (int kx = 0; kx < 3; kx++)
#pragma unroll
              for (int tx = 0; tx < 3; tx++) {
                if (!NZ[b][kx][tx]) continue;
                o = fma2(CY[b][kx][tx], tx == 0 ? Z[a][kx][0] : tx == 1 ? M[kx] : Z[a][kx][1], o);
              }
            o0[2 * a + b] = o.x;
            o1[2 * a + b] = o.y;
          }
        }
        const float m8 = max_raw(max_raw(max_raw(o0[0], o0[1]), max_raw(o0[2], o0[3])), max_raw(max_raw(o1[0], o1[1]), max_raw(o1[2], o1[3])));
        if (m8 > sum) sum = m8;
        for (int dy = 0; dy < 3; dy++) { const unsigned n = q[dy] + 5 * PB; q[dy] = n >= qlim ? n - RING : n; }
        __syncthreads();
      }
    }
  }
  out[blockIdx.x * THREADS + tid] = sum;
}

template <int MIX>
float run(const char *name) {
  const int iters = 2000, blocks = 256 * 2 * 4;   // 4 workgroups per resident slot
  float *out, *w;
  CHECK(hipMalloc(&out, sizeof(float) * blocks * THREADS));
  CHECK(hipMalloc(&w, 64 * 64 * 4));
  CHECK(hipMemset(w, 0, 64 * 64 * 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MIX>), dim3(blocks), dim3(THREADS), 0, 0, out, w, iters);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MIX>), dim3(blocks), dim3(THREADS), 0, 0, out, w, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  // cycles per sub-step of a resident workgroup pair: 4 rounds of 2000 sub-steps
  const double cyc = ms * 1e-3 * 2.4e9 / (4.0 * iters);
  printf("%-60s %8.3f ms  = %6.0f cycles per sub-step\n", name, ms, cyc);
  CHECK(hipFree(out)); CHECK(hipFree(w));
  return ms;
}

int main() {
  const float a = run<0>("r03 mix: B tile pair | C pass on 4x4x1 MFMAs");
  const float b = run<1>("tap-plane mix: B tile pair + 1x1 (4x4x1) | C stencil on VALU");
  printf("tap-plane / r03 = %.3f\n", b / a);
  return 0;
}
