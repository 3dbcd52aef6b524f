// Micro-probe (round 3, for the asm-owned attention body of DESIGN section 10): how much VALU work fits in the shadow of
// a wave's OWN v_mfma_f32_16x16x32_bf16 stream, with one and with two waves per SIMD, and what CLUSTERING costs.
// Every wave runs `iters` times a body of PER MFMAs (8 independent accumulators, round-robin) followed by a VALU group of
// NE v_exp_f32 + NC v_cvt_pk_bf16_f32 + NM v_mul_f32 (all on independent registers). PER = 1 spreads the VALU work evenly
// between the MFMAs; PER = 32 with NE = 32, NC = 16 is the shape of today's attention tile body ([QK^T cluster][exp cluster]).
// The attention kernel's mix per MFMA is ~0.9 exp + 0.45 cvt + 0.35 other (64 + 32 + 25 VALU per 72 MFMAs).
// Reported: shader-clock ticks per MFMA of wave 0 of block 0 (s_memtime) and the wall time of a whole-chip launch
// (256 blocks: realistic clocks). 16 ticks per MFMA per wave = the matrix pipe of the SIMD is saturated with 1 wave,
// 32 with two waves. Build: hipcc --offload-arch=gfx950 -O2 -o mfma_valu_mix mfma_valu_mix.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int PER, int NE, int NC, int NM, int NACC = 8>
__global__ __launch_bounds__(512) void mix(float* out, long long* cyc, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  float e[8], c[8], m[8];
  for (int i = 0; i < 8; ++i) {
    e[i] = -0.001f * (threadIdx.x + i) - 0.3f;
    c[i] = 0.5f + i;
    m[i] = 1.0f;
  }
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  constexpr int G = PER < 8 ? 8 / PER : 1;            // groups per loop iteration: consecutive MFMAs always rotate over all 8 accumulators
  for (int it = 0; it < iters / G; ++it) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int k = 0; k < PER; ++k) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, %0" : "+v"(acc[(g * PER + k) & (NACC - 1)]) : "v"(a));
#pragma unroll
      for (int j = 0; j < NE; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(e[(g * NE + j) & 7]));
#pragma unroll
      for (int j = 0; j < NC; ++j) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(c[(g * NC + j) & 7]));
#pragma unroll
      for (int j = 0; j < NM; ++j) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(m[(g * NM + j) & 7]));
    }
  }
  asm volatile("s_nop 15\n s_nop 15" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + e[i] + c[i] + m[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

// The same with v_mfma_f32_32x32x16_bf16 (twice the FLOP per issued instruction, 32 matrix-pipe cycles): 4 independent accumulators of 16 registers
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int PER, int NE, int NC, int NM, int NACC = 4>
__global__ __launch_bounds__(512) void mix32(float* out, long long* cyc, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  float e[8], c[8], m[8];
  for (int i = 0; i < 8; ++i) {
    e[i] = -0.001f * (threadIdx.x + i) - 0.3f;
    c[i] = 0.5f + i;
    m[i] = 1.0f;
  }
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  constexpr int G = PER < 4 ? 4 / PER : 1;
  for (int it = 0; it < iters / G; ++it) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int k = 0; k < PER; ++k) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %1, %0" : "+v"(acc[(g * PER + k) & (NACC - 1)]) : "v"(a));
#pragma unroll
      for (int j = 0; j < NE; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(e[(g * NE + j) & 7]));
#pragma unroll
      for (int j = 0; j < NC; ++j) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(c[(g * NC + j) & 7]));
#pragma unroll
      for (int j = 0; j < NM; ++j) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(m[(g * NM + j) & 7]));
    }
  }
  asm volatile("s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  for (int i = 0; i < 8; ++i) s += e[i] + c[i] + m[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int PER, int NE, int NC, int NM, int NACC = 4>
void run32(const char* what) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  const int total_mfma = 1 << 17;                    // 32x32x16 MFMAs per wave (= 2^18 16x16x32 equivalents)
  const int iters = total_mfma / PER;
  for (int waves = 1; waves <= 2; ++waves) {
    const int threads = 256 * waves;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mix32<PER, NE, NC, NM, NACC><<<256, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mix32<PER, NE, NC, NM, NACC><<<256, threads>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[8];
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-52s %d wave/SIMD: %7.2f ticks per 32x32x16 MFMA (wave 0), %7.2f (last wave) | chip launch %.3f ms = %.1f ns per 16x16x32 EQUIVALENT per wave\n", what,
           waves, (double)h[0] / total_mfma, (double)h[threads / 64 - 1] / total_mfma, ms, ms * 1e6 / (2.0 * total_mfma));
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  hipFree(out); hipFree(cyc);
}

template <int PER, int NE, int NC, int NM, int NACC = 8>
void run(const char* what) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  const int total_mfma = 1 << 18;                    // MFMAs per wave
  const int iters = total_mfma / PER;
  for (int waves = 1; waves <= 2; ++waves) {
    const int threads = 256 * waves;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mix<PER, NE, NC, NM, NACC><<<256, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mix<PER, NE, NC, NM, NACC><<<256, threads>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[8];
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    const double per0 = (double)h[0] / total_mfma, perl = (double)h[threads / 64 - 1] / total_mfma;
    // matrix-pipe utilisation of a SIMD = waves * 16 cycles per MFMA / measured cycles per MFMA (if s_memtime ticks are shader clocks)
    printf("%-44s %d wave/SIMD: %7.2f ticks per MFMA (wave 0), %7.2f (last wave) | chip launch %.3f ms = %.1f ns per MFMA per wave\n", what, waves, per0, perl,
           ms, ms * 1e6 / total_mfma);
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  hipFree(out); hipFree(cyc);
}

int main() {
  run<1, 0, 0, 0>("MFMA only");
  run<1, 1, 0, 0>("1 MFMA : 1 exp");
  run<1, 2, 0, 0>("1 MFMA : 2 exp");
  run<1, 3, 0, 0>("1 MFMA : 3 exp");
  run<1, 0, 1, 0>("1 MFMA : 1 cvt_pk");
  run<1, 0, 0, 1>("1 MFMA : 1 mul");
  run<1, 0, 0, 2>("1 MFMA : 2 mul");
  run<1, 0, 0, 3>("1 MFMA : 3 mul");
  run<1, 1, 1, 0>("1 MFMA : 1 exp + 1 cvt");
  run<2, 2, 1, 1>("2 MFMA : 2 exp + 1 cvt + 1 mul (attention mix)");
  run<4, 4, 2, 1>("4 MFMA : 4 exp + 2 cvt + 1 mul");
  run<8, 8, 4, 3>("8 MFMA : 8 exp + 4 cvt + 3 mul");
  run<16, 16, 8, 6>("16 MFMA : 16 exp + 8 cvt + 6 mul");
  run<32, 32, 16, 11>("32 MFMA : 32 exp + 16 cvt + 11 mul (today's clusters)");
  run<72, 64, 32, 25>("72 MFMA : 64 exp + 32 cvt + 25 mul (whole tile clustered)");
  run<1, 4, 0, 0>("1 MFMA : 4 exp (VALU bound?)");
  printf("\n---- v_mfma_f32_32x32x16_bf16: the same VALU work per FLOP needs twice the VALU per MFMA ----\n");
  run32<1, 0, 0, 0>("MFMA32 only");
  run32<1, 1, 0, 0>("1 MFMA32 : 1 exp");
  run32<1, 2, 0, 0>("1 MFMA32 : 2 exp");
  run32<1, 3, 0, 0>("1 MFMA32 : 3 exp");
  run32<1, 4, 0, 0>("1 MFMA32 : 4 exp");
  run32<1, 2, 1, 0>("1 MFMA32 : 2 exp + 1 cvt");
  run32<1, 2, 1, 1>("1 MFMA32 : 2 exp + 1 cvt + 1 mul (attention mix)");
  run32<2, 4, 2, 1>("2 MFMA32 : 4 exp + 2 cvt + 1 mul");
  run32<4, 8, 4, 3>("4 MFMA32 : 8 exp + 4 cvt + 3 mul");
  run32<8, 16, 8, 6>("8 MFMA32 : 16 exp + 8 cvt + 6 mul");
  run32<16, 32, 16, 11>("16 MFMA32 : 32 exp + 16 cvt + 11 mul");
  run32<36, 64, 32, 25>("36 MFMA32 : 64 exp + 32 cvt + 25 mul (whole tile clustered)");
  printf("\n---- dependent MFMAs: the same accumulator every 1 / 2 / 4 (/ 8) MFMAs, bare and with one VALU filler per MFMA ----\n");
  run<1, 0, 0, 0, 1>("16x16x32, 1 accumulator (back-to-back dependent)");
  run<1, 0, 0, 0, 2>("16x16x32, 2 accumulators");
  run<1, 0, 0, 0, 4>("16x16x32, 4 accumulators");
  run<1, 1, 0, 0, 1>("16x16x32, 1 accumulator  + 1 exp per MFMA");
  run<1, 1, 0, 0, 2>("16x16x32, 2 accumulators + 1 exp per MFMA");
  run<1, 1, 0, 0, 4>("16x16x32, 4 accumulators + 1 exp per MFMA");
  run<1, 0, 0, 1, 1>("16x16x32, 1 accumulator  + 1 mul per MFMA");
  run<1, 0, 0, 1, 2>("16x16x32, 2 accumulators + 1 mul per MFMA");
  run32<1, 0, 0, 0, 1>("32x32x16, 1 accumulator (back-to-back dependent)");
  run32<1, 0, 0, 0, 2>("32x32x16, 2 accumulators");
  run32<1, 2, 1, 1, 1>("32x32x16, 1 accumulator  + attention mix");
  run32<1, 2, 1, 1, 2>("32x32x16, 2 accumulators + attention mix");
  run32<1, 0, 0, 1, 1>("32x32x16, 1 accumulator  + 1 mul per MFMA");
  run32<1, 0, 0, 1, 2>("32x32x16, 2 accumulators + 1 mul per MFMA");
  return 0;
}
