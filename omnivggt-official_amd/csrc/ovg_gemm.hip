// Linear layers of the aggregator as MFMA GEMMs with fused epilogues (gfx950).
//
//   Y[m,n] = epilogue( sum_k X[m,k] * W[n,k] + bias[n] )
//
// Replaces nn.Linear/addmm at attention.py:52 (qkv), attention.py:75 (proj),
// mlp.py:35-38 (fc1/GELU/fc2) and the Conv2d-as-GEMM of patch_embed.py:65, plus the
// elementwise tails the reference runs as separate ATen ops (q/k LayerNorm
// attention.py:54, RoPE rope.py:154-188, LayerScale layer_scale.py:27, residual
// block.py:105-106, camera-token injection omnivggt_aggregator.py:284-301).
//
// Tiling: 128(m) x 128(n) x 128 B(k) per 256-thread workgroup, 4 waves as 2(n) x 2(m),
// each wave 64x64 = 4x4 MFMA 16x16 tiles.  W rows are the MFMA A operand, X rows the
// B operand, so a lane ends up with 4 consecutive n for one token m: bias / gamma /
// residual / output accesses are 16-byte vectors.  LDS tiles are XOR-swizzled
// (ovg_common.h: swz_off) so the ds_read_b128 fragment reads are conflict free.
#include "ovg_common.h"

namespace {

constexpr int BM = 128, BN = 128;

// exact-erf GELU (nn.GELU default, mlp.py:22).  f32 parity mode calls libm's erff; the 16-bit modes
// use Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below bf16/f16 resolution): PMC showed the
// erff expansion (~40 VALU per element) cost as many VALU instructions as the whole fc1 main loop.
OVG_DEV float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(ax * ax * -1.4426950408889634f);
  return copysignf(fmaf(-p * t, e, 1.0f), x);
}
template <typename T> OVG_DEV float gelu_erf(float x) {
  if constexpr (sizeof(T) == 4) return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
  else return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f));
}

// Logical block id -> (m tile, n tile), "grouped" order: GM m-tiles x all n-tiles at a time, m fastest.
// Each XCD works on a contiguous run of ids (xcd_remap), so the ~96 blocks resident on one XCD cover
// GM X-tiles x ~12 W-tiles (~5 MB): both operands stay in that XCD's 4 MB L2 instead of re-streaming
// the whole weight matrix per m-tile (measured with plain n-fastest order: 61 % L2 hit rate, 590 MB of
// fabric reads for a 30 MB problem on fc1).
OVG_DEV void tile_coords(int lid, int mtiles, int ntiles_gm, int& tm, int& tn) {
  const int ntiles = ntiles_gm & 0xffff, GM = (ntiles_gm >> 16) & 0xff;   // GM == 0: plain n-fastest order
  if (GM == 0) { tm = lid / ntiles; tn = lid - tm * ntiles; return; }
  const int per_group = GM * ntiles;
  const int grp = lid / per_group, rem = lid - grp * per_group;
  const int m_first = grp * GM;
  const int gsz = (mtiles - m_first) < GM ? (mtiles - m_first) : GM;
  tm = m_first + rem % gsz;
  tn = rem / gsz;
}

// ---------------------------------------------------------------------------
// Main loop: leaves acc[nt][mt] = C[n = n0w + 16nt + 4g + r][m = m0w + 16mt + (lane&15)]
// ---------------------------------------------------------------------------
template <typename T>
OVG_DEV void gemm_mainloop(const T* __restrict__ X, int64_t ldx, const T* __restrict__ W, int64_t ldw,
                           int M, int N, int K, int m0, int n0, unsigned char* lds, f32x4 (&acc)[4][4]) {
  constexpr int BKB = 128;                       // bytes of k per step
  unsigned char* Ws = lds;
  unsigned char* Xs = lds + BN * BKB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wm = wave & 1;
  const int g = lane >> 4, lr = lane & 15;

  const unsigned char* xg[4];
  const unsigned char* wg[4];
  int soff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (tid >> 3) + 32 * i, ch = tid & 7;
    int xr = m0 + row; xr = xr < M ? xr : M - 1;
    int wr = n0 + row; wr = wr < N ? wr : N - 1;
    xg[i] = reinterpret_cast<const unsigned char*>(X + (int64_t)xr * ldx) + ch * 16;
    wg[i] = reinterpret_cast<const unsigned char*>(W + (int64_t)wr * ldw) + ch * 16;
    soff[i] = swz_off<128>(row, ch);
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (K * (int)sizeof(T)) / BKB;
  u32x4 rx[4], rw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    rx[i] = *reinterpret_cast<const u32x4*>(xg[i]);
    rw[i] = *reinterpret_cast<const u32x4*>(wg[i]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    *reinterpret_cast<u32x4*>(Xs + soff[i]) = rx[i];
    *reinterpret_cast<u32x4*>(Ws + soff[i]) = rw[i];
  }
  __syncthreads();

  const int sx = lr >> 1;                         // swizzle term of this lane's rows
  const int wrow = (wn * 64 + lr) * 128, xrow = (wm * 64 + lr) * 128;

  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + 1) < nk;
    if (more) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        rx[i] = *reinterpret_cast<const u32x4*>(xg[i] + (int64_t)(kt + 1) * BKB);
        rw[i] = *reinterpret_cast<const u32x4*>(wg[i] + (int64_t)(kt + 1) * BKB);
      }
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int coff = ((kk * 4 + g) ^ sx) << 4;
      u32x4 a[4], b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[t] = *reinterpret_cast<const u32x4*>(Ws + wrow + t * 16 * 128 + coff);
        b[t] = *reinterpret_cast<const u32x4*>(Xs + xrow + t * 16 * 128 + coff);
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) TT<T>::mma(acc[nt][mt], a[nt], b[mt]);
    }
    __syncthreads();
    if (more) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<u32x4*>(Xs + soff[i]) = rx[i];
        *reinterpret_cast<u32x4*>(Ws + soff[i]) = rw[i];
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------
// Linear epilogues (STORE / GELU / RES / PATCH) on a wave's 64(n) x 16*MT(m) accumulator block:
// acc[nt][mt] = C[n = n_w0 + 16nt + 4g + r][m = m_w0 + 16mt + (lane&15)]
// ---------------------------------------------------------------------------
template <typename T, int EPI, bool OUT_F32, int MT>
OVG_DEV void linear_epilogue(const ovg_linear_params& p, const f32x4 (&acc)[4][MT], const int m_w0, const int n_w0) {
  const int M = (int)p.M, N = (int)p.N;
  const int lane = threadIdx.x & 63, g = lane >> 4, lr = lane & 15;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m_w0 + mt * 16 + lr;
    if (m >= M) continue;
    int64_t orow = m;
    int trow = 0;
    bool inj = false;
    if constexpr (EPI == OVG_EPI_PATCH) {
      const int v = m / (int)p.p0, t = m % (int)p.p0;
      orow = (int64_t)v * p.p1 + p.row_off + t;
      trow = t + 1;
    }
    if constexpr (EPI == OVG_EPI_RES) { inj = (p.inject != nullptr) && (m % (int)p.inj_period == 0); }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n_w0 + nt * 16 + 4 * g;
      f32x4 v = acc[nt][mt];
      if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
      if constexpr (EPI == OVG_EPI_GELU) {
        v[0] = gelu_erf<T>(v[0]); v[1] = gelu_erf<T>(v[1]); v[2] = gelu_erf<T>(v[2]); v[3] = gelu_erf<T>(v[3]);
      }
      if constexpr (EPI == OVG_EPI_RES) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(p.res + (int64_t)m * p.ldres + n);
        const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + n);
        v = r + gm * v;
        if (inj) v += *reinterpret_cast<const f32x4*>(p.inject + (int64_t)(m / (int)p.inj_period) * N + n);
      }
      if constexpr (EPI == OVG_EPI_PATCH) { v += *reinterpret_cast<const f32x4*>(p.table + (int64_t)trow * N + n); }
      if constexpr (OUT_F32 || EPI == OVG_EPI_RES || EPI == OVG_EPI_PATCH) {
        *reinterpret_cast<f32x4*>(static_cast<float*>(p.y) + orow * p.ldy + n) = v;
      } else {
        store4<T>(static_cast<T*>(p.y) + orow * p.ldy + n, v[0], v[1], v[2], v[3]);
      }
    }
  }
}

template <typename T, int EPI, bool OUT_F32>
__global__ __launch_bounds__(256, 2) void linear_kernel(ovg_linear_params p, int ntiles_n) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 128 * 128];
  const int M = (int)p.M, N = (int)p.N, K = (int)p.K;
  int tm, tn;
  tile_coords(xcd_remap(blockIdx.x, gridDim.x), (M + BM - 1) / BM, ntiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  f32x4 acc[4][4];
  gemm_mainloop<T>(static_cast<const T*>(p.x), p.ldx, static_cast<const T*>(p.w), p.ldw, M, N, K, m0, n0, lds, acc);
  const int wave = threadIdx.x >> 6;
  linear_epilogue<T, EPI, OUT_F32, 4>(p, acc, m0 + (wave & 1) * 64, n0 + (wave >> 1) * 64);
}

// ---------------------------------------------------------------------------
// QKV epilogue on a wave's 64(n) x 16*MT(m) block (64 columns = one head of q, k or v):
// bias + per-head LayerNorm(64) + 2-D RoPE + q scale, head-major stores, V transposed
// ---------------------------------------------------------------------------
template <typename T, int MT>
OVG_DEV void qkv_epilogue(const ovg_qkv_params& p, const f32x4 (&acc)[4][MT], const int m_w0, const int ncol0_) {
  const int ncol0 = __builtin_amdgcn_readfirstlane(ncol0_);   // wave-uniform: keep the q/k/v dispatch scalar
  const int M = (int)p.M;
  const int lane = threadIdx.x & 63, g = lane >> 4, lr = lane & 15;
  const int seq = (int)p.seq;
  const int which = ncol0 / OVG_C;                      // 0 q, 1 k, 2 v (uniform per wave)
  const int h = (ncol0 % OVG_C) / OVG_D;                // head of this wave's 64 columns

  float bias[16];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + ncol0 + nt * 16 + 4 * g);
    bias[nt * 4 + 0] = b[0]; bias[nt * 4 + 1] = b[1]; bias[nt * 4 + 2] = b[2]; bias[nt * 4 + 3] = b[3];
  }
  float nw[16], nb[16];
  const bool do_norm = p.qk_norm && which < 2;
  if (do_norm) {
    const float* w_ = which == 0 ? p.qn_w : p.kn_w;
    const float* b_ = which == 0 ? p.qn_b : p.kn_b;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(w_ + nt * 16 + 4 * g);
      const f32x4 b = *reinterpret_cast<const f32x4*>(b_ + nt * 16 + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) { nw[nt * 4 + r] = a[r]; nb[nt * 4 + r] = b[r]; }
    }
  }

#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m_w0 + mt * 16 + lr;
    const bool valid = m < M;
    float v[16];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[nt * 4 + r] = acc[nt][mt][r] + bias[nt * 4 + r];

    const int mm = valid ? m : M - 1;
    const int bidx = mm / seq, n = mm % seq;
    if (which < 2) {
      if (do_norm) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += v[i];
        s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
        const float mean = s * (1.0f / 64.0f);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const float d = v[i] - mean; q += d * d; }
        q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
        const float rstd = 1.0f / sqrtf(q * (1.0f / 64.0f) + p.qk_eps);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = (v[i] - mean) * rstd * nw[i] + nb[i];
      }
      if (p.rope) {
        const int t = mm % (int)p.tokens_per_view;
        int py = 0, px = 0;
        if (t >= p.n_special) { const int pp = t - p.n_special; py = pp / p.grid_w + 1; px = pp % p.grid_w + 1; }
        const f32x4 cy = *reinterpret_cast<const f32x4*>(p.rope_cos + py * 16 + 4 * g);
        const f32x4 sy = *reinterpret_cast<const f32x4*>(p.rope_sin + py * 16 + 4 * g);
        const f32x4 cx = *reinterpret_cast<const f32x4*>(p.rope_cos + px * 16 + 4 * g);
        const f32x4 sxn = *reinterpret_cast<const f32x4*>(p.rope_sin + px * 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a0 = v[r], a1 = v[4 + r];          // features j, j+16 of the y half
          v[r] = a0 * cy[r] - a1 * sy[r];
          v[4 + r] = a1 * cy[r] + a0 * sy[r];
          const float b0 = v[8 + r], b1 = v[12 + r];     // features j, j+16 of the x half
          v[8 + r] = b0 * cx[r] - b1 * sxn[r];
          v[12 + r] = b1 * cx[r] + b0 * sxn[r];
        }
      }
      if (which == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] *= p.q_scale;
      }
      if (valid) {
        const int64_t npad = which == 0 ? p.nq_pad : p.nk_pad;
        T* dst = static_cast<T*>(which == 0 ? p.q : p.k) + (((int64_t)bidx * OVG_H + h) * npad + n) * OVG_D + 4 * g;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) store4<T>(dst + nt * 16, v[nt * 4], v[nt * 4 + 1], v[nt * 4 + 2], v[nt * 4 + 3]);
      }
    } else if (valid) {
      T* dst = static_cast<T*>(p.vt) + ((int64_t)bidx * OVG_H + h) * OVG_D * p.nk_pad + n;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(int64_t)(nt * 16 + 4 * g + r) * p.nk_pad] = TT<T>::from_f32(v[nt * 4 + r]);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256, 2) void qkv_kernel(ovg_qkv_params p, int nt_begin, int nt_count) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 128 * 128];
  constexpr int N = 3 * OVG_C, K = OVG_C;
  const int M = (int)p.M;
  int tm, tn;
  tile_coords(xcd_remap(blockIdx.x, gridDim.x), (M + BM - 1) / BM, nt_count, tm, tn);   // nt_count carries GM in its high half
  const int m0 = tm * BM, n0 = (nt_begin + tn) * BN;
  f32x4 acc[4][4];
  gemm_mainloop<T>(static_cast<const T*>(p.x), p.ldx, static_cast<const T*>(p.w), (int64_t)K, M, N, K, m0, n0, lds, acc);
  const int wave = threadIdx.x >> 6;
  qkv_epilogue<T, 4>(p, acc, m0 + (wave & 1) * 64, n0 + (wave >> 1) * 64);
}

#include "ovg_gemm256.h"

// 256 x 256 ping-pong variants (16-bit modes): same epilogues on acc[4][8]
template <typename T, int EPI, bool OUT_F32>
__global__ __launch_bounds__(512) void linear256_kernel(ovg_linear_params p, int ntiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds256[];
  const int M = (int)p.M, N = (int)p.N, K = (int)p.K;
  int tm, tn;
  tile_coords(xcd_remap(blockIdx.x, gridDim.x), (M + g256::BM2 - 1) / g256::BM2, ntiles_n, tm, tn);
  const int m0 = tm * g256::BM2, n0 = tn * g256::BN2;
  f32x4 acc[4][8];
  g256::mainloop<T>(static_cast<const T*>(p.x), p.ldx, static_cast<const T*>(p.w), p.ldw, M, N, K, m0, n0, lds256, acc);
  const int wave = threadIdx.x >> 6;
  linear_epilogue<T, EPI, OUT_F32, 8>(p, acc, m0 + (wave >> 2) * 128, n0 + (wave & 3) * 64);
}

template <typename T>
__global__ __launch_bounds__(512) void qkv256_kernel(ovg_qkv_params p, int nt_begin, int nt_count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds256[];
  constexpr int N = 3 * OVG_C, K = OVG_C;
  const int M = (int)p.M;
  int tm, tn;
  tile_coords(xcd_remap(blockIdx.x, gridDim.x), (M + g256::BM2 - 1) / g256::BM2, nt_count, tm, tn);
  const int m0 = tm * g256::BM2, n0 = (nt_begin + tn) * g256::BN2;
  f32x4 acc[4][8];
  g256::mainloop<T>(static_cast<const T*>(p.x), p.ldx, static_cast<const T*>(p.w), (int64_t)K, M, N, K, m0, n0, lds256, acc);
  const int wave = threadIdx.x >> 6;
  qkv_epilogue<T, 8>(p, acc, m0 + (wave >> 2) * 128, n0 + (wave & 3) * 64);
}

template <typename KernelT>
int allow_big_lds(KernelT kernel) {     // once per kernel: opt in to > 64 KB of dynamic LDS
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, g256::LDS_BYTES) == hipSuccess ? OVG_OK : OVG_E_LAUNCH;
}

// tile-order group sizes (m-tiles per group, tile_coords): measured in profiles/r01_gemm_tile_order_ab.txt / r01_gemm256_ab.txt
constexpr int TILE_GROUP = 8, TILE_GROUP256 = 4;

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename T>
int launch_linear128(const ovg_linear_params& p, hipStream_t st) {
  const int mt = (int)((p.M + BM - 1) / BM), nt = (int)(p.N / BN);
  const dim3 grid(mt * nt), block(256);
  const int ntg = nt | (TILE_GROUP << 16);
  switch (p.epilogue) {
    case OVG_EPI_STORE:
      if (p.out_f32) OVG_LAUNCH((linear_kernel<T, OVG_EPI_STORE, true>), grid, block, 0, st, p, ntg);
      else OVG_LAUNCH((linear_kernel<T, OVG_EPI_STORE, false>), grid, block, 0, st, p, ntg);
      break;
    case OVG_EPI_GELU:
      OVG_LAUNCH((linear_kernel<T, OVG_EPI_GELU, false>), grid, block, 0, st, p, ntg);
      break;
    case OVG_EPI_RES:
      OVG_LAUNCH((linear_kernel<T, OVG_EPI_RES, true>), grid, block, 0, st, p, ntg);
      break;
    case OVG_EPI_PATCH:
      OVG_LAUNCH((linear_kernel<T, OVG_EPI_PATCH, true>), grid, block, 0, st, p, ntg);
      break;
    default: return OVG_E_ARG;
  }
  OVG_CHECK_LAUNCH();
  return OVG_OK;
}
template <typename T, int EPI, bool OUT_F32>
int launch_linear256_one(const ovg_linear_params& p, hipStream_t st) {
  static const int ok = allow_big_lds(linear256_kernel<T, EPI, OUT_F32>);
  if (ok != OVG_OK) return ok;
  const int mt = (int)((p.M + g256::BM2 - 1) / g256::BM2), nt = (int)(p.N / g256::BN2);
  const int ntg = nt | (TILE_GROUP256 << 16);
  OVG_LAUNCH((linear256_kernel<T, EPI, OUT_F32>), dim3(mt * nt), dim3(512), g256::LDS_BYTES, st, p, ntg);
  OVG_CHECK_LAUNCH();
  return OVG_OK;
}
template <typename T>
int launch_linear256(const ovg_linear_params& p, hipStream_t st) {
  switch (p.epilogue) {
    case OVG_EPI_STORE: return p.out_f32 ? launch_linear256_one<T, OVG_EPI_STORE, true>(p, st) : launch_linear256_one<T, OVG_EPI_STORE, false>(p, st);
    case OVG_EPI_GELU: return launch_linear256_one<T, OVG_EPI_GELU, false>(p, st);
    case OVG_EPI_RES: return launch_linear256_one<T, OVG_EPI_RES, true>(p, st);
    case OVG_EPI_PATCH: return launch_linear256_one<T, OVG_EPI_PATCH, true>(p, st);
    default: return OVG_E_ARG;
  }
}
// Tile choice for the 16-bit modes (measured, tests/bench_kernels.py gemm, profiles/r01_gemm256_ab.txt): the 256 x 256
// ping-pong loop wins by 9-14 % on QKV / fc1 / fc2 once its tiles fill the 256 CUs evenly, and loses on the proj GEMM
// (K = 1024 with the f32 residual epilogue: one workgroup per CU cannot overlap that epilogue with another workgroup's
// main loop) and on badly quantised grids (QKV at M = 10 992: 516 tiles = 2.02 rounds).
// Returns 1 = use 256^2, 0 = use 128^2, -1 = the caller forced a tile this shape / dtype cannot run.
int choose_256(int tile, bool sixteen_bit, int64_t M, int64_t N, int64_t K, bool light_epilogue_or_long_k) {
  const bool legal = sixteen_bit && N % g256::BN2 == 0 && K % 32 == 0;
  if (tile == OVG_TILE_128) return 0;
  if (tile == OVG_TILE_256) return legal ? 1 : -1;
  if (tile != OVG_TILE_AUTO) return -1;
  if (!legal) return 0;
  // in situ (bench.py, whole forward) the 256^2 kernels only pay off for long token slices: at M = 10 992 the
  // forward is 2 % faster with 128^2 everywhere, at M = 87 936 it is 2 % faster with this choice
  if (!light_epilogue_or_long_k || M < 32768) return 0;
  const int64_t tiles = ((M + g256::BM2 - 1) / g256::BM2) * (N / g256::BN2);
  const int64_t rounds = (tiles + 255) / 256;
  return (tiles * 100 >= rounds * 256 * 80 || K >= 2048) ? 1 : 0;   // >= 80 % of the last round's CUs busy
}

template <typename T>
int launch_linear(const ovg_linear_params& p, hipStream_t st) {
  const int big = choose_256(p.tile, sizeof(T) == 2, p.M, p.N, p.K, p.epilogue != OVG_EPI_RES || p.K >= 2048);
  if (big < 0) return OVG_E_ARG;
  if constexpr (sizeof(T) == 2) {
    if (big) return launch_linear256<T>(p, st);
  }
  return launch_linear128<T>(p, st);
}

}  // namespace

extern "C" int ovg_linear(const ovg_linear_params* p, void* stream) {
  if (!p || !p->x || !p->w || !p->y) return OVG_E_ARG;
  if (p->M <= 0 || p->N <= 0 || p->K <= 0 || p->M > (1 << 30)) return OVG_E_ARG;
  if (p->N % BN != 0 || p->K % 64 != 0) return OVG_E_ARG;
  const int64_t esz = p->dtype == OVG_F32 ? 4 : 2;
  if ((p->ldx * esz) % 16 || (p->ldw * esz) % 16 || !aligned16(p->x) || !aligned16(p->w) || !aligned16(p->y)) return OVG_E_ARG;
  if (p->bias && !aligned16(p->bias)) return OVG_E_ARG;
  if (p->epilogue == OVG_EPI_RES) {
    if (!p->res || !p->gamma || (p->ldres % 4) || (p->ldy % 4) || !aligned16(p->res) || !aligned16(p->gamma)) return OVG_E_ARG;
    if (p->inject && (p->inj_period <= 0 || !aligned16(p->inject))) return OVG_E_ARG;
  } else if (p->epilogue == OVG_EPI_PATCH) {
    if (!p->table || p->p0 <= 0 || p->p1 <= 0 || (p->ldy % 4) || !aligned16(p->table)) return OVG_E_ARG;
  } else {
    const int64_t osz = p->out_f32 ? 4 : esz;
    if ((p->ldy * osz) % (4 * osz)) return OVG_E_ARG;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (p->dtype) {
    case OVG_BF16: return launch_linear<bf16_t>(*p, st);
    case OVG_F16: return launch_linear<f16_t>(*p, st);
    case OVG_F32: return launch_linear<float>(*p, st);
    default: return OVG_E_DTYPE;
  }
}

extern "C" int ovg_qkv(const ovg_qkv_params* p, void* stream) {
  if (!p || !p->x || !p->w || !p->bias || !p->q || !p->k || !p->vt) return OVG_E_ARG;
  if (p->M <= 0 || p->M > (1 << 30) || p->seq <= 0 || p->M % p->seq != 0) return OVG_E_ARG;
  if (p->nq_pad < p->seq || p->nk_pad < p->seq || p->nk_pad % OVG_KV_TILE != 0) return OVG_E_ARG;
  if (p->dtype != OVG_BF16 && p->dtype != OVG_F16 && p->dtype != OVG_F32) return OVG_E_DTYPE;
  const int64_t esz = p->dtype == OVG_F32 ? 4 : 2;
  if ((p->ldx * esz) % 16 || !aligned16(p->x) || !aligned16(p->w) || !aligned16(p->bias) || !aligned16(p->q) || !aligned16(p->k) || !aligned16(p->vt)) return OVG_E_ARG;
  if (p->qk_norm && (!p->qn_w || !p->qn_b || !p->kn_w || !p->kn_b)) return OVG_E_ARG;
  if (p->rope) {
    if (!p->rope_cos || !p->rope_sin || p->tokens_per_view <= 0 || p->grid_w <= 0) return OVG_E_ARG;
    const int64_t np = p->tokens_per_view - p->n_special;
    if (np <= 0 || (np - 1) / p->grid_w + 1 >= p->max_pos || p->grid_w >= p->max_pos) return OVG_E_ARG;
  }
  if (p->part < 0 || p->part > 2) return OVG_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int big = choose_256(p->tile, p->dtype != OVG_F32, p->M, (p->part == 0 ? 3 : (p->part == 1 ? 2 : 1)) * OVG_C, OVG_C, true);
  if (big < 0) return OVG_E_ARG;
  if (big) {
    const int q_t = OVG_C / g256::BN2, all_t = 3 * OVG_C / g256::BN2;
    const int ntb = p->part == 1 ? q_t : 0;
    const int ntc = p->part == 0 ? all_t : (p->part == 1 ? all_t - q_t : q_t);
    const dim3 grid2((unsigned)(((p->M + g256::BM2 - 1) / g256::BM2) * ntc));
    const int ntg2 = ntc | (TILE_GROUP256 << 16);
    if (p->dtype == OVG_BF16) {
      static const int ok = allow_big_lds(qkv256_kernel<bf16_t>);
      if (ok != OVG_OK) return ok;
      OVG_LAUNCH((qkv256_kernel<bf16_t>), grid2, dim3(512), g256::LDS_BYTES, st, *p, ntb, ntg2);
    } else {
      static const int ok = allow_big_lds(qkv256_kernel<f16_t>);
      if (ok != OVG_OK) return ok;
      OVG_LAUNCH((qkv256_kernel<f16_t>), grid2, dim3(512), g256::LDS_BYTES, st, *p, ntb, ntg2);
    }
    OVG_CHECK_LAUNCH();
    return OVG_OK;
  }
  const int q_tiles = OVG_C / BN, all_tiles = 3 * OVG_C / BN;
  const int nt_begin = p->part == 1 ? q_tiles : 0;
  const int nt_count = p->part == 0 ? all_tiles : (p->part == 1 ? all_tiles - q_tiles : q_tiles);
  const dim3 grid((unsigned)(((p->M + BM - 1) / BM) * nt_count)), block(256);
  const int ntg = nt_count | (TILE_GROUP << 16);
  switch (p->dtype) {
    case OVG_BF16: OVG_LAUNCH((qkv_kernel<bf16_t>), grid, block, 0, st, *p, nt_begin, ntg); break;
    case OVG_F16: OVG_LAUNCH((qkv_kernel<f16_t>), grid, block, 0, st, *p, nt_begin, ntg); break;
    default: OVG_LAUNCH((qkv_kernel<float>), grid, block, 0, st, *p, nt_begin, ntg); break;
  }
  OVG_CHECK_LAUNCH();
  return OVG_OK;
}
