// Micro-probe: operand / accumulator layout of v_mfma_f32_32x32x16_bf16 as the lab attention kernel (tools/lab/ovg_attn32_lab.h.in) assumes it:
//   A[row = lane % 32][k = 8 (lane / 32) + j], B[k = 8 (lane / 32) + j][col = lane % 32], j = 0..7 (8 bf16 = one 16-byte operand)
//   D[row = 8 (i / 4) + 4 (lane / 32) + i % 4][col = lane % 32] in accumulator register i = 0..15
// Test 1: A[m][k] = m + 1, B[k][n] = (k == n % 16)      -> D[m][n] = m + 1          (row map of D and of A)
// Test 2: A[m][k] = k + 1, B[k][n] = (k == n % 16)      -> D[m][n] = n % 16 + 1     (A and B agree on k)
// Test 3: A[m][k] = (k == m % 16), B[k][n] = k + 2 n    -> D[m][n] = m % 16 + 2 n   (column map of D and of B)
// Build: hipcc --offload-arch=gfx950 -O2 -o mfma32_layout mfma32_layout.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void probe(float* out, int test) {
  const int lane = threadIdx.x, r = lane % 32, h = lane / 32;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * h + j;
    float av, bv;
    if (test == 1) { av = r + 1; bv = (k == r % 16) ? 1.f : 0.f; }
    else if (test == 2) { av = k + 1; bv = (k == r % 16) ? 1.f : 0.f; }
    else { av = (k == r % 16) ? 1.f : 0.f; bv = k + 2 * r; }
    a[j] = (__bf16)av;
    b[j] = (__bf16)bv;
  }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int i = 0; i < 16; ++i) out[lane * 16 + i] = c[i];
}

int main() {
  float* d; hipMalloc(&d, 64 * 16 * 4);
  float hst[64 * 16];
  int bad_total = 0;
  for (int test = 1; test <= 3; ++test) {
    probe<<<1, 64>>>(d, test);
    hipMemcpy(hst, d, sizeof(hst), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
      for (int i = 0; i < 16; ++i) {
        const int m = 8 * (i / 4) + 4 * (lane / 32) + i % 4, n = lane % 32;
        const float want = test == 1 ? m + 1 : (test == 2 ? n % 16 + 1 : m % 16 + 2 * n);
        if (hst[lane * 16 + i] != want) {
          if (bad < 4) printf("  test %d lane %d reg %d: got %g, assumed layout says %g\n", test, lane, i, hst[lane * 16 + i], want);
          ++bad;
        }
      }
    printf("mfma32 layout test %d: %s (%d mismatches)\n", test, bad ? "MISMATCH" : "as assumed", bad);
    bad_total += bad;
  }
  return bad_total ? 1 : 0;
}
