// Micro-probe: issue cost (cycles per wave64 instruction) of v_exp_f32, v_mul_f32, v_cvt_pk_bf16_f32,
// v_max3_f32 and v_pk_mul_f32 with one wave per SIMD, and v_exp_f32 beside a stream of MFMAs from a
// second wave on the same SIMD.  Build: hipcc --offload-arch=gfx950 -O2 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ __launch_bounds__(256) void probe(float* out, long long* cyc, int iters, int mfma_waves) {
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = 0.001f * (threadIdx.x + i) - 0.3f;
  const int wave = threadIdx.x >> 6;
  f32x4 acc = {0, 0, 0, 0};
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  long long t0 = 0, t1 = 0;
  __syncthreads();
  if (wave >= 4 && mfma_waves) {            // companion waves: MFMA stream on the same SIMDs
    for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
      for (int k = 0; k < 16; ++k)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
    }
    out[threadIdx.x] = acc[0];
    return;
  }
  if (wave >= 4) return;
  t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
        if (OP == 1) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(v[i]));
        if (OP == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(v[i]));
        if (OP == 3) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(v[i]));
        if (OP == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
        if (OP == 5) asm volatile("v_exp_f16 %0, %0" : "+v"(v[i]));
      }
    }
  }
  t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += v[i];
  out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int OP>
void run(const char* name, int mfma) {
  float* out; long long* cyc;
  hipMalloc(&out, 4096); hipMalloc(&cyc, 64 * 8);
  const int iters = 2000;
  const int threads = mfma ? 512 : 256;
  probe<OP><<<1, threads>>>(out, cyc, iters, mfma);
  hipDeviceSynchronize();
  probe<OP><<<1, threads>>>(out, cyc, iters, mfma);
  hipDeviceSynchronize();
  long long h[4];
  hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
  printf("%-22s %s: %.2f shader-clock ticks per wave64 instruction (wave0), %.2f (wave3)\n", name, mfma ? "beside MFMA wave" : "alone          ",
         (double)h[0] / (iters * 64.0), (double)h[3] / (iters * 64.0));
  hipFree(out); hipFree(cyc);
}

int main() {
  run<1>("v_mul_f32", 0);
  run<0>("v_exp_f32", 0);
  run<4>("v_rcp_f32", 0);
  run<5>("v_exp_f16", 0);
  run<2>("v_cvt_pk_bf16_f32", 0);
  run<3>("v_max3_f32", 0);
  run<1>("v_mul_f32", 1);
  run<0>("v_exp_f32", 1);
  run<2>("v_cvt_pk_bf16_f32", 1);
  return 0;
}
