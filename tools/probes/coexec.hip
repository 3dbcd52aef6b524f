// Micro-probe: does a VALU stream of one wave overlap with an MFMA stream of ANOTHER wave on the same SIMD?
// Block = 8 waves (2 per SIMD): waves 0-3 issue back-to-back independent MFMAs (16x16x32 bf16), waves 4-7 issue
// a VALU stream (v_exp_f32 / v_mul_f32 / v_cvt_pk) or nothing. Reports wall time of the MFMA waves and of the VALU
// waves alone and together. Build: hipcc --offload-arch=gfx950 -O2 -o coexec coexec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int OP>
__global__ __launch_bounds__(512) void probe(float* out, int mfma_iters, int valu_iters, int valu_first, int prio_valu, int prio_mfma) {
  const int wave = threadIdx.x >> 6;
  const bool is_mfma = valu_first ? (wave >= 4) : (wave < 4);
  if (is_mfma) {
    if (prio_mfma) __builtin_amdgcn_s_setprio(3);
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, a), acc[k], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    out[threadIdx.x] = s;
  } else {
    if (prio_valu) __builtin_amdgcn_s_setprio(3);
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = 0.001f * (threadIdx.x + i) - 0.3f;
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
        if (OP == 1) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(v[i]));
        if (OP == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(v[i]));
      }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i];
    out[threadIdx.x] = s;
  }
}

template <int OP>
float run(int mi, int vi, int valu_first = 0, int prio_valu = 0, int prio_mfma = 0) {
  float* out; hipMalloc(&out, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<OP><<<256, 512>>>(out, mi, vi, valu_first, prio_valu, prio_mfma);          // one block per CU, warm-up
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<OP><<<256, 512>>>(out, mi, vi, valu_first, prio_valu, prio_mfma);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(out);
  return ms;
}

int main() {
  const int MI = 40000;                            // 320k MFMAs per wave: 5.1 M cycles at 16 cycles each
  const char* names[3] = {"v_exp_f32", "v_mul_f32", "v_cvt_pk_bf16_f32"};
  const int VI[3] = {40000, 80000, 64000};         // 640k exp (8 cyc) / 1.28M mul (4 cyc) / 1.02M cvt: each ~5 M cycles alone
  float m = run<1>(MI, 0);
  printf("MFMA waves alone: %.3f ms\n", m);
  float r[3][2];
  r[0][0] = run<0>(0, VI[0]); r[0][1] = run<0>(MI, VI[0]);
  r[1][0] = run<1>(0, VI[1]); r[1][1] = run<1>(MI, VI[1]);
  r[2][0] = run<2>(0, VI[2]); r[2][1] = run<2>(MI, VI[2]);
  for (int i = 0; i < 3; ++i)
    printf("%-20s alone %.3f ms | together with the MFMA waves %.3f ms | sum %.3f max %.3f -> overlap fraction %.2f\n", names[i], r[i][0], r[i][1],
           r[i][0] + m, r[i][0] > m ? r[i][0] : m, (r[i][0] + m - r[i][1]) / (r[i][0] < m ? r[i][0] : m));
  // who wins the issue port? (v_exp stream, both streams ~equal length alone)
  const float e = r[0][0];
  printf("\nv_exp beside MFMA, by age / priority (alone: MFMA %.3f, v_exp %.3f, sum %.3f):\n", m, e, m + e);
  printf("  MFMA waves older, no prio      : %.3f ms\n", run<0>(MI, VI[0], 0, 0, 0));
  printf("  VALU waves older, no prio      : %.3f ms\n", run<0>(MI, VI[0], 1, 0, 0));
  printf("  MFMA older, VALU at setprio 3  : %.3f ms\n", run<0>(MI, VI[0], 0, 1, 0));
  printf("  VALU older, MFMA at setprio 3  : %.3f ms\n", run<0>(MI, VI[0], 1, 0, 1));
  printf("  MFMA older, MFMA at setprio 3  : %.3f ms\n", run<0>(MI, VI[0], 0, 0, 1));
  printf("  VALU older, VALU at setprio 3  : %.3f ms\n", run<0>(MI, VI[0], 1, 1, 0));
  return 0;
}
