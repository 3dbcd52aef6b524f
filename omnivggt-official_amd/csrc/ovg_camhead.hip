// Camera head on gfx950 (the part of it that was still a PyTorch module):
// the iterative pose regressor of omnivggt/heads/camera_head.py:84-154 -- token_norm, then `iters` rounds of
// [embed_pose -> SiLU -> poseLN_modulation -> adaLN modulate -> 4 trunk blocks (layers/block.py:81-107 with
// dim 2048, 16 heads x 128, no RoPE / q-k-norm) -> trunk_norm -> pose_branch -> pose += delta -> activation
// (heads/head_act.py:12-35: translation / quaternion linear, field of view ReLU)] -- on S camera tokens.
//
// Shape of the problem: M = S rows (8 ... 128) against 216 M weight parameters per round. Every GEMM is a weight
// STREAM (2 bytes per parameter per round, 1.7 GB over the four rounds in the 16-bit modes; twice that in the f32 parity
// mode, whose GEMM operands and activation buffers are f32 and whose products run on the exact-f32 MFMA), i.e. HBM bound;
// the MFMA only has to keep up with the loads. So:
//  * ch_gemm_partial: one wave = 16 weight rows x a K range, the weight fragments go global -> registers -> MFMA A operand
//    (each element is used once per 64 tokens: no LDS staging), the S token rows are the B operand (L2 resident);
//    split-K over enough workgroups to fill the chip (>= 512 where K allows), deterministic: every split writes its own f32 partial;
//  * the partial sums are folded by the CONSUMER: the row kernels (bias + LayerScale + residual + the next LayerNorm in one
//    pass over the 2048-wide row), the adaLN modulation, the pose update; only QKV and fc1 have a stand-alone finish
//    (bias / bias + exact GELU -> compute dtype);
//  * attention over S keys with head dim 128 is ~1 MFLOP per head: one 128-thread workgroup per (head, query), f32.
// One C call issues the ~41 launches of a round back to back (the host never looks at intermediate results).
// Always f32: residual stream, LayerNorm statistics, softmax, biases, LayerScale, the 9-wide pose tensors.
#include "ovg_common.h"
#include <math.h>

namespace {

constexpr int CH = 2048, CH_HEADS = 16, CH_HD = 128, CH_HID = 8192, CH_MOD = 6144, CH_PB = 1024, CH_T = 9;
constexpr int CH_MAX_S = 4096;            // attention scores of one query live in LDS
constexpr int CH_PART_COLS = 32768;       // ksplit * N of any GEMM here (partial buffer = S x this many floats)

// one 16-byte chunk = TT<T>::kPerChunk elements (8 in the 16-bit modes, 4 in f32)
template <typename T> OVG_DEV void unpack_chunk(const u32x4& raw, float (&f)[TT<T>::kPerChunk]) {
  T v[TT<T>::kPerChunk];
  __builtin_memcpy(v, &raw, 16);
#pragma unroll
  for (int i = 0; i < TT<T>::kPerChunk; ++i) f[i] = TT<T>::to_f32(v[i]);
}

// sum over the 256 threads of a workgroup (red: 4 floats of LDS, reusable after the call returns on all threads)
OVG_DEV float block_sum256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// A 2048-wide f32 row as 2 x float4 per thread of a 256-thread workgroup: elements 4t..4t+3 and 1024+4t..1024+4t+3.
struct Row8 {
  f32x4 a, b;
  OVG_DEV void load(const float* p) {
    a = *reinterpret_cast<const f32x4*>(p + 4 * threadIdx.x);
    b = *reinterpret_cast<const f32x4*>(p + 1024 + 4 * threadIdx.x);
  }
  OVG_DEV void store(float* p) const {
    *reinterpret_cast<f32x4*>(p + 4 * threadIdx.x) = a;
    *reinterpret_cast<f32x4*>(p + 1024 + 4 * threadIdx.x) = b;
  }
  template <typename T> OVG_DEV void store_t(T* p) const {
    store4<T>(p + 4 * threadIdx.x, a[0], a[1], a[2], a[3]);
    store4<T>(p + 1024 + 4 * threadIdx.x, b[0], b[1], b[2], b[3]);
  }
};

// LayerNorm of the row held by the workgroup (two-pass statistics like ATen): returns the normalised row,
// with affine parameters if w != nullptr.
OVG_DEV Row8 row_layernorm(const Row8& x, const float* w, const float* b, float eps, float* red) {
  float s = x.a[0] + x.a[1] + x.a[2] + x.a[3] + x.b[0] + x.b[1] + x.b[2] + x.b[3];
  const float mean = block_sum256(s, red) * (1.0f / CH);
  Row8 d;
  d.a = x.a - mean;
  d.b = x.b - mean;
  float q = d.a[0] * d.a[0] + d.a[1] * d.a[1] + d.a[2] * d.a[2] + d.a[3] * d.a[3] + d.b[0] * d.b[0] + d.b[1] * d.b[1] + d.b[2] * d.b[2] + d.b[3] * d.b[3];
  const float var = block_sum256(q, red) * (1.0f / CH);
  const float rstd = 1.0f / sqrtf(var + eps);
  d.a *= rstd;
  d.b *= rstd;
  if (w != nullptr) {
    Row8 ww, bb;
    ww.load(w);
    bb.load(b);
    d.a = d.a * ww.a + bb.a;
    d.b = d.b * ww.b + bb.b;
  }
  return d;
}

// sum of the split-K partials of columns [col0 + 4t, +4) and [col0 + 1024 + 4t, +4) of row m (+ bias)
OVG_DEV Row8 row_partials(const float* part, int ksplit, int S, int N, int m, int col0, const float* bias) {
  Row8 r;
  r.load(bias + col0);
  for (int ks = 0; ks < ksplit; ++ks) {
    Row8 p;
    p.load(part + ((int64_t)ks * S + m) * N + col0);
    r.a += p.a;
    r.b += p.b;
  }
  return r;
}

// ---- token_norm: tok = LayerNorm(tokens[:, 0]) (camera_head.py:100-101) -------------------------------------------------
__global__ __launch_bounds__(256) void ch_token_norm_kernel(const float* tokens, int64_t ld, const float* w, const float* b, float eps, float* tok) {
  __shared__ float red[4];
  const int m = blockIdx.x;
  Row8 x;
  x.load(tokens + (int64_t)m * ld);
  row_layernorm(x, w, b, eps, red).store(tok + (int64_t)m * CH);
}

// ---- e = SiLU(embed_pose(pose)) (camera_head.py:124-132; the SiLU opens poseLN_modulation) -------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ch_embed_kernel(const float* pose, int pose_stride, const float* w, const float* b, T* e) {
  const int m = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
  const float* p = pose + (int64_t)m * pose_stride;
  float v = b[c];
#pragma unroll
  for (int i = 0; i < CH_T; ++i) v += w[c * CH_T + i] * p[i];
  v = v / (1.0f + expf(-v));
  e[(int64_t)m * CH + c] = TT<T>::from_f32(v);
}

// ---- x = gate * (adaLN(tok) * (1 + scale) + shift) + tok; xn = LayerNorm_n1(x) (camera_head.py:135-139 + the first
//      trunk block's norm1); (shift, scale, gate) = the three 2048-wide chunks of the modulation GEMM, summed from its partials
template <typename T>
__global__ __launch_bounds__(256) void ch_modulate_kernel(const float* tok, const float* part, int ksplit, int S, const float* mod_b,
                                                           const float* n_w, const float* n_b, float n_eps, float* x, T* xn) {
  __shared__ float red[4];
  const int m = blockIdx.x;
  Row8 t;
  t.load(tok + (int64_t)m * CH);
  const Row8 ln = row_layernorm(t, nullptr, nullptr, 1e-6f, red);
  const Row8 shift = row_partials(part, ksplit, S, CH_MOD, m, 0, mod_b);
  const Row8 scale = row_partials(part, ksplit, S, CH_MOD, m, CH, mod_b);
  const Row8 gate = row_partials(part, ksplit, S, CH_MOD, m, 2 * CH, mod_b);
  Row8 h;
  h.a = gate.a * (ln.a * (1.0f + scale.a) + shift.a) + t.a;
  h.b = gate.b * (ln.b * (1.0f + scale.b) + shift.b) + t.b;
  h.store(x + (int64_t)m * CH);
  row_layernorm(h, n_w, n_b, n_eps, red).store_t<T>(xn + (int64_t)m * CH);
}

// ---- x += ls * (sum of partials + bias); xn = LayerNorm(x) with the NEXT norm (block.py:105-106 + the following norm) ----
template <typename T>
__global__ __launch_bounds__(256) void ch_residual_kernel(const float* part, int ksplit, int S, const float* bias, const float* ls,
                                                           const float* n_w, const float* n_b, float n_eps, float* x, T* xn) {
  __shared__ float red[4];
  const int m = blockIdx.x;
  Row8 r, g;
  r.load(x + (int64_t)m * CH);
  g.load(ls);
  const Row8 y = row_partials(part, ksplit, S, CH, m, 0, bias);
  r.a += g.a * y.a;
  r.b += g.b * y.b;
  r.store(x + (int64_t)m * CH);
  row_layernorm(r, n_w, n_b, n_eps, red).store_t<T>(xn + (int64_t)m * CH);
}

// ---- stand-alone finish: out = sum of partials + bias (EPI 0) / exact-erf GELU of it (EPI 1), in the compute dtype ---------
template <typename T, int EPI>
__global__ __launch_bounds__(256) void ch_finish_kernel(const float* part, int ksplit, int S, int N, const float* bias, T* out) {
  const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;      // one thread = 4 columns of one row
  const int64_t total4 = (int64_t)S * N / 4;
  if (i4 >= total4) return;
  const int m = (int)(i4 / (N / 4)), n = (int)(i4 % (N / 4)) * 4;
  f32x4 v = *reinterpret_cast<const f32x4*>(bias + n);
  for (int ks = 0; ks < ksplit; ++ks) v += *reinterpret_cast<const f32x4*>(part + ((int64_t)ks * S + m) * N + n);
  if constexpr (EPI == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = 0.5f * v[i] * (1.0f + erff(v[i] * 0.70710678118654752f));
  }
  store4<T>(out + (int64_t)m * N + n, v[0], v[1], v[2], v[3]);
}

// ---- split-K weight-streaming GEMM: part[ks][m][n] = sum_{k in chunk ks} x[m][k] * w[n][k] ---------------------------------
// grid (N / 64, ksplit, ceil(S / 64)), 4 waves: wave v owns weight rows n0 + 16 v .. + 15 (the MFMA A operand, straight from
// global memory: lane (r = lane & 15, g = lane >> 4) supplies w[n0 + r][k0 + E g .. + E - 1], E = elements per 16 bytes: 8 in the
// 16-bit modes (one 16x16x32 MFMA per fragment), 4 in f32 (four exact-f32 16x16x4 MFMAs, TT<float>::mma)); the <= 64 token rows
// of the z slice are the B operand. Result lane (c, g) holds out[m0 + 16 mb + c][n0 + 4 g .. + 3] -> one 16-byte store per
// 16-token block.
template <typename T>
__global__ __launch_bounds__(256) void ch_gemm_partial_kernel(const T* x, int64_t ldx, const T* w, int64_t ldw, float* part, int S, int N, int kchunk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 64 + wave * 16;
  const int k_begin = blockIdx.y * kchunk;
  const int m0 = blockIdx.z * 64;
  const int nmb = (S - m0 + 15) / 16 < 4 ? (S - m0 + 15) / 16 : 4;
  constexpr int E = TT<T>::kPerChunk;               // a fragment (one 16-byte load per lane) covers 4 E consecutive k
  const T* wp = w + (int64_t)(n0 + r) * ldw + k_begin + E * g;
  const T* xp[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    int m = m0 + 16 * mb + r;
    m = m < S ? m : S - 1;
    xp[mb] = x + (int64_t)m * ldx + k_begin + E * g;
  }
  f32x4 acc[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < kchunk; k += 32 * E) {       // kchunk % 256 == 0 (host); 8 weight fragments (8 KB per wave) in flight
    u32x4 wf[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) wf[u] = *reinterpret_cast<const u32x4*>(wp + k + 4 * E * u);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      if (mb < nmb) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const u32x4 xf = *reinterpret_cast<const u32x4*>(xp[mb] + k + 4 * E * u);
          TT<T>::mma(acc[mb], wf[u], xf);
        }
      }
    }
  }
  float* pp = part + ((int64_t)blockIdx.y * S) * N + n0 + 4 * g;
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const int m = m0 + 16 * mb + r;
    if (mb < nmb && m < S) *reinterpret_cast<f32x4*>(pp + (int64_t)m * N) = acc[mb];
  }
}

// ---- attention of the trunk: 16 heads x 128, S keys, no mask (attention.py:50-77 without q/k-norm and RoPE) ----------------
// one 128-thread workgroup per (head, query): scores and probabilities in LDS, f32 throughout
template <typename T>
__global__ __launch_bounds__(128) void ch_attn_kernel(const T* qkv, T* out, int S, float scale) {
  __shared__ float qs[CH_HD];
  __shared__ float ps[CH_MAX_S];
  __shared__ float red[2];
  const int h = blockIdx.x, qi = blockIdx.y, t = threadIdx.x;
  qs[t] = TT<T>::to_f32(qkv[(int64_t)qi * 3 * CH + h * CH_HD + t]) * scale;
  __syncthreads();
  float lmax = -INFINITY;
  for (int j = t; j < S; j += 128) {
    const T* kr = qkv + (int64_t)j * 3 * CH + CH + h * CH_HD;
    float s = 0.f;
    constexpr int E = TT<T>::kPerChunk;
#pragma unroll
    for (int c = 0; c < CH_HD / E; ++c) {
      float kf[E];
      unpack_chunk<T>(*reinterpret_cast<const u32x4*>(kr + E * c), kf);
#pragma unroll
      for (int i = 0; i < E; ++i) s += qs[E * c + i] * kf[i];
    }
    ps[j] = s;
    lmax = fmaxf(lmax, s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o, 64));
  if ((t & 63) == 0) red[t >> 6] = lmax;
  __syncthreads();
  const float mx = fmaxf(red[0], red[1]);
  float lsum = 0.f;
  for (int j = t; j < S; j += 128) {
    const float p = expf(ps[j] - mx);
    ps[j] = p;
    lsum += p;
  }
  lsum = wave_sum(lsum);
  __syncthreads();                                  // every thread has read red (the max) and written its ps entries
  if ((t & 63) == 0) red[t >> 6] = lsum;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1]);
  const T* vc = qkv + 2 * CH + h * CH_HD + t;
  float o = 0.f;
  for (int j = 0; j < S; ++j) o += ps[j] * TT<T>::to_f32(vc[(int64_t)j * 3 * CH]);
  out[(int64_t)qi * CH + h * CH_HD + t] = TT<T>::from_f32(o * inv);
}

// ---- pose update: delta = fc2(GELU(fc1 partials + b1)); pose = (first ? 0 : pose) + delta; out = activate(pose) -------------
__global__ __launch_bounds__(256) void ch_pose_kernel(const float* part, int ksplit, int S, const float* b1, const float* w2, const float* b2,
                                                       float* pose, int first, float* out) {
  __shared__ float red[4][CH_T];
  const int m = blockIdx.x, t = threadIdx.x;
  float acc[CH_T];
#pragma unroll
  for (int o = 0; o < CH_T; ++o) acc[o] = 0.f;
#pragma unroll
  for (int j = 0; j < CH_PB / 256; ++j) {
    const int k = t + 256 * j;
    float v = b1[k];
    for (int ks = 0; ks < ksplit; ++ks) v += part[((int64_t)ks * S + m) * CH_PB + k];
    v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
#pragma unroll
    for (int o = 0; o < CH_T; ++o) acc[o] += v * w2[o * CH_PB + k];
  }
#pragma unroll
  for (int o = 0; o < CH_T; ++o) {
    const float s = wave_sum(acc[o]);
    if ((t & 63) == 0) red[t >> 6][o] = s;
  }
  __syncthreads();
  if (t < CH_T) {
    const float delta = red[0][t] + red[1][t] + red[2][t] + red[3][t] + b2[t];
    const float p = (first ? 0.f : pose[(int64_t)m * CH_T + t]) + delta;
    pose[(int64_t)m * CH_T + t] = p;
    out[(int64_t)m * CH_T + t] = t < 7 ? p : fmaxf(p, 0.f);      // head_act.py:12-35: T, quaternion linear; FoV relu
  }
}

// ---- host side ----------------------------------------------------------------------------------------------------------------
struct Ws {
  float *tok, *x, *pose, *part;
  void *xn, *e, *qkv, *attn, *hid;
  int64_t total;
};

int64_t align256(int64_t v) { return (v + 255) & ~(int64_t)255; }

Ws carve(void* base, int S, int esz) {
  Ws w{};
  int64_t off = 0;
  auto take = [&](int64_t bytes) { void* p = base ? static_cast<char*>(base) + off : nullptr; off += align256(bytes); return p; };
  w.tok = static_cast<float*>(take((int64_t)S * CH * 4));
  w.x = static_cast<float*>(take((int64_t)S * CH * 4));
  w.pose = static_cast<float*>(take((int64_t)S * CH_T * 4));
  w.part = static_cast<float*>(take((int64_t)S * CH_PART_COLS * 4));
  w.xn = take((int64_t)S * CH * esz);
  w.e = take((int64_t)S * CH * esz);
  w.qkv = take((int64_t)S * 3 * CH * esz);
  w.attn = take((int64_t)S * CH * esz);
  w.hid = take((int64_t)S * CH_HID * esz);
  w.total = off;
  return w;
}

// split-K factor: aim at >= 512 workgroups (two per CU: the stream is latency bound per wave), keep >= 256 k per split and
// the partial buffer within CH_PART_COLS columns
int pick_ksplit(int S, int N, int K) {
  const int zs = (S + 63) / 64;
  int ks = 1;
  while ((N / 64) * zs * ks < 512 && ks < 16 && K / (ks * 2) >= 256 && (int64_t)(ks * 2) * N <= CH_PART_COLS) ks *= 2;
  return ks;
}

template <typename T>
int gemm(const void* x, const void* w, float* part, int S, int N, int K, int* ksplit, hipStream_t st) {
  const int ks = pick_ksplit(S, N, K);
  *ksplit = ks;
  const dim3 grid((unsigned)(N / 64), (unsigned)ks, (unsigned)((S + 63) / 64));
  OVG_LAUNCH((ch_gemm_partial_kernel<T>), grid, dim3(256), 0, st, static_cast<const T*>(x), (int64_t)K, static_cast<const T*>(w), (int64_t)K, part, S, N, K / ks);
  OVG_CHECK_LAUNCH();
  return OVG_OK;
}

template <typename T, int EPI>
int finish(const float* part, int ksplit, int S, int N, const float* bias, void* out, hipStream_t st) {
  const int64_t total4 = (int64_t)S * N / 4;
  OVG_LAUNCH((ch_finish_kernel<T, EPI>), dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, part, ksplit, S, N, bias, static_cast<T*>(out));
  OVG_CHECK_LAUNCH();
  return OVG_OK;
}

#define CH_TRY(expr) do { const int rc_ = (expr); if (rc_ != OVG_OK) return rc_; } while (0)

template <typename T>
int run(const ovg_camera_head_params& p, hipStream_t st) {
  const int S = p.S;
  const Ws ws = carve(p.ws, S, (int)sizeof(T));
  T* xn = static_cast<T*>(ws.xn);
  const float ln_eps = 1e-5f;                                   // nn.LayerNorm default (block.py:50,67; camera_head.py:63-64)
  OVG_LAUNCH(ch_token_norm_kernel, dim3(S), dim3(256), 0, st, p.tokens, p.ld_tokens, p.token_norm_w, p.token_norm_b, ln_eps, ws.tok);
  OVG_CHECK_LAUNCH();
  int ks = 1;
  for (int it = 0; it < p.iters; ++it) {
    OVG_LAUNCH((ch_embed_kernel<T>), dim3(S, CH / 256), dim3(256), 0, st, it == 0 ? p.empty_pose : ws.pose, it == 0 ? 0 : CH_T, p.embed_w, p.embed_b,
               static_cast<T*>(ws.e));
    OVG_CHECK_LAUNCH();
    CH_TRY(gemm<T>(ws.e, p.mod_w, ws.part, S, CH_MOD, CH, &ks, st));
    OVG_LAUNCH((ch_modulate_kernel<T>), dim3(S), dim3(256), 0, st, ws.tok, ws.part, ks, S, p.mod_b, p.blk[0].n1_w, p.blk[0].n1_b, ln_eps, ws.x, xn);
    OVG_CHECK_LAUNCH();
    for (int b = 0; b < p.trunk_depth; ++b) {
      const ovg_camera_block_weights& w = p.blk[b];
      CH_TRY(gemm<T>(xn, w.qkv_w, ws.part, S, 3 * CH, CH, &ks, st));
      CH_TRY((finish<T, 0>(ws.part, ks, S, 3 * CH, w.qkv_b, ws.qkv, st)));
      OVG_LAUNCH((ch_attn_kernel<T>), dim3(CH_HEADS, S), dim3(128), 0, st, static_cast<const T*>(ws.qkv), static_cast<T*>(ws.attn), S, 0.08838834764831845f);
      OVG_CHECK_LAUNCH();
      CH_TRY(gemm<T>(ws.attn, w.proj_w, ws.part, S, CH, CH, &ks, st));
      OVG_LAUNCH((ch_residual_kernel<T>), dim3(S), dim3(256), 0, st, ws.part, ks, S, w.proj_b, w.ls1, w.n2_w, w.n2_b, ln_eps, ws.x, xn);
      OVG_CHECK_LAUNCH();
      CH_TRY(gemm<T>(xn, w.fc1_w, ws.part, S, CH_HID, CH, &ks, st));
      CH_TRY((finish<T, 1>(ws.part, ks, S, CH_HID, w.fc1_b, ws.hid, st)));
      CH_TRY(gemm<T>(ws.hid, w.fc2_w, ws.part, S, CH, CH_HID, &ks, st));
      const bool last = b + 1 == p.trunk_depth;
      OVG_LAUNCH((ch_residual_kernel<T>), dim3(S), dim3(256), 0, st, ws.part, ks, S, w.fc2_b, w.ls2, last ? p.trunk_norm_w : p.blk[b + 1].n1_w,
                 last ? p.trunk_norm_b : p.blk[b + 1].n1_b, ln_eps, ws.x, xn);
      OVG_CHECK_LAUNCH();
    }
    CH_TRY(gemm<T>(xn, p.pb1_w, ws.part, S, CH_PB, CH, &ks, st));
    OVG_LAUNCH(ch_pose_kernel, dim3(S), dim3(256), 0, st, ws.part, ks, S, p.pb1_b, p.pb2_w, p.pb2_b, ws.pose, it == 0 ? 1 : 0, p.out + (int64_t)it * S * CH_T);
    OVG_CHECK_LAUNCH();
  }
  return OVG_OK;
}

}  // namespace

extern "C" int64_t ovg_camera_head_workspace_bytes(int32_t S, int32_t dtype) {
  if (S <= 0 || S > CH_MAX_S || (dtype != OVG_BF16 && dtype != OVG_F16 && dtype != OVG_F32)) return -1;
  return carve(nullptr, S, dtype == OVG_F32 ? 4 : 2).total;
}

extern "C" int ovg_camera_head(const ovg_camera_head_params* p, void* stream) {
  if (!p || !p->tokens || !p->out || !p->ws) return OVG_E_ARG;
  if (p->S <= 0 || p->iters <= 0 || p->ld_tokens < CH || (p->ld_tokens & 3)) return OVG_E_ARG;
  if (p->S > CH_MAX_S || p->trunk_depth < 1 || p->trunk_depth > OVG_CAMERA_MAX_TRUNK) return OVG_E_UNSUPPORTED;
  if (p->dim != CH || p->heads != CH_HEADS) return OVG_E_UNSUPPORTED;
  if (p->dtype != OVG_BF16 && p->dtype != OVG_F16 && p->dtype != OVG_F32) return OVG_E_DTYPE;
  if (p->ws_bytes < carve(nullptr, p->S, p->dtype == OVG_F32 ? 4 : 2).total) return OVG_E_ARG;
  const void* req[] = {p->token_norm_w, p->token_norm_b, p->trunk_norm_w, p->trunk_norm_b, p->empty_pose, p->embed_w, p->embed_b,
                       p->mod_w, p->mod_b, p->pb1_w, p->pb1_b, p->pb2_w, p->pb2_b};
  for (const void* q : req)
    if (!q) return OVG_E_ARG;
  for (int b = 0; b < p->trunk_depth; ++b) {
    const ovg_camera_block_weights& w = p->blk[b];
    const void* rq[] = {w.n1_w, w.n1_b, w.n2_w, w.n2_b, w.ls1, w.ls2, w.qkv_w, w.qkv_b, w.proj_w, w.proj_b, w.fc1_w, w.fc1_b, w.fc2_w, w.fc2_b};
    for (const void* q : rq)
      if (!q) return OVG_E_ARG;
    if ((reinterpret_cast<uintptr_t>(w.qkv_w) | reinterpret_cast<uintptr_t>(w.proj_w) | reinterpret_cast<uintptr_t>(w.fc1_w) | reinterpret_cast<uintptr_t>(w.fc2_w)) & 15)
      return OVG_E_ARG;
  }
  if ((reinterpret_cast<uintptr_t>(p->tokens) | reinterpret_cast<uintptr_t>(p->ws) | reinterpret_cast<uintptr_t>(p->mod_w) | reinterpret_cast<uintptr_t>(p->pb1_w)) & 15) return OVG_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  return p->dtype == OVG_BF16 ? run<bf16_t>(*p, st) : p->dtype == OVG_F16 ? run<f16_t>(*p, st) : run<float>(*p, st);
}
