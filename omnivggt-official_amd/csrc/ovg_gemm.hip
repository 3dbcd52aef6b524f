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
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

namespace {

constexpr int BM = 128, BN = 128;

// Lab instrumentation (tools/probes/gemm_timeline.py builds the library with -DOVG_GEMM_TIMELINE; the product build compiles none of it):
// lane 0 of the first and of the last wave of every workgroup stamps the 100 MHz wall clock at kernel entry (0), when the first k-stage is
// visible (1), behind the main loop (2) and behind the epilogue (3); g_tl_stagger > 0 delays the workgroups of the first round by
// ((blockIdx.x >> 3) & 7) steps so that the CUs stop running their tiles in lockstep.
#ifdef OVG_GEMM_TIMELINE
__device__ unsigned long long* g_tl_buf = nullptr;
__device__ int g_tl_stagger = 0;
OVG_DEV void tl_mark(int slot) {
  if ((threadIdx.x & 63) != 0 || g_tl_buf == nullptr) return;
  const int wave = threadIdx.x >> 6, last = (blockDim.x >> 6) - 1;
  if (wave != 0 && wave != last) return;
  unsigned long long* row = g_tl_buf + (size_t)blockIdx.x * 16 + (wave == 0 ? 0 : 8);
  row[slot] = wall_clock64();
  if (slot == 0) { row[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4); row[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20); row[6] = clock64(); }
  if (slot == 3) row[7] = clock64();
}
OVG_DEV void tl_stagger() {
  if (g_tl_stagger <= 0 || blockIdx.x >= 256) return;
  const unsigned long long until = wall_clock64() + (unsigned long long)(((blockIdx.x >> 3) & 7) * g_tl_stagger);
  while (wall_clock64() < until) __builtin_amdgcn_s_sleep(16);
}
#else
OVG_DEV void tl_mark(int) {}
OVG_DEV void tl_stagger() {}
#endif

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
// Transcendental-free form for the 16-bit modes (r03): GELU(x) = x * Phi(x), Phi(x) = 1/2 + xc * R(xc^2), xc = clamp(x, -5, 5),
// R a degree-12 minimax polynomial in u = 2 xc^2 / 25 - 1 (fitted against erf in double; evaluated in f32 Horner form the
// error of GELU is <= 2.2e-6 absolute for |x| <= 12 and <= 3e-7 |x| beyond: two decimal orders below bf16 / f16 resolution,
// the same class as erf_as). 17 plain VALU operations, 16 of them FMA / MUL that hipcc pairs into v_pk_fma_f32 / v_pk_mul_f32
// across neighbouring elements, no v_rcp / v_exp and none of the ~2 s_nop per element their result hazards cost
// (ISA of the fc1 epilogue: 13.5 -> 9 issue slots per element).
// One row block of a lane (16 values) at a time, Horner step by Horner step ACROSS the 8 register pairs: a v_pk_fma_f32 that
// consumes the previous packed result back-to-back costs an s_nop each (hipcc, evaluating chain after chain, emitted 652 s_nop for
// 896 v_pk_fma_f32); eight independent chains in flight need none.
OVG_DEV void gelu_poly16(f32x4 (&v)[4]) {
  constexpr float c[13] = {1.413637876e-01f, -7.029628064e-02f, 5.152052248e-02f, -4.044260701e-02f, 3.144217031e-02f, -2.326865498e-02f,
                           1.640218381e-02f, -1.116433345e-02f, 6.410491046e-03f, -2.685581490e-03f, 1.722817794e-03f, -1.613262584e-03f,
                           6.087207876e-04f};
  f32x4 xc[4], u[4], r[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) xc[nt][i] = __builtin_amdgcn_fmed3f(v[nt][i], -5.0f, 5.0f);
    u[nt] = xc[nt] * xc[nt] * 0.08f - 1.0f;
    r[nt] = u[nt] * c[12] + c[11];
  }
#pragma unroll
  for (int k = 10; k >= 0; --k) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) r[nt] = r[nt] * u[nt] + c[k];
  }
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) v[nt] = v[nt] * (xc[nt] * r[nt] + 0.5f);
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
// X3 (OVG_F16X2): the operands are (hi, lo) plane pairs; the k loop runs three passes over K into the same accumulators --
// x_lo * w_hi, x_hi * w_lo, x_hi * w_hi (small terms first) -- as ONE loop of 3 nk virtual k-stages whose source planes are picked
// per stage (wave-uniform selects), so the staging pipeline never drains between the passes.
template <typename T, bool SWAP = false, bool X3 = false>   // SWAP: operands trade places, every 16 x 16 block transposed (see ovg_gemm256.h)
OVG_DEV void gemm_mainloop(const T* __restrict__ X, int64_t ldx, const T* __restrict__ W, int64_t ldw,
                           int M, int N, int K, int m0, int n0, unsigned char* lds, f32x4 (&acc)[4][4],
                           const T* __restrict__ Xlo = nullptr, const T* __restrict__ Wlo = nullptr) {
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

  const int nk1 = (K * (int)sizeof(T)) / BKB;
  const int nk = X3 ? 3 * nk1 : nk1;
  const int64_t dxl = X3 ? reinterpret_cast<const unsigned char*>(Xlo) - reinterpret_cast<const unsigned char*>(X) : 0;
  const int64_t dwl = X3 ? reinterpret_cast<const unsigned char*>(Wlo) - reinterpret_cast<const unsigned char*>(W) : 0;
  auto src_off = [&](int kt, int64_t& xo, int64_t& wo) {     // byte offsets of virtual k-stage kt from the hi-plane row pointers
    if constexpr (X3) {
      const int pass = kt >= 2 * nk1 ? 2 : (kt >= nk1 ? 1 : 0);
      const int64_t kb = (int64_t)(kt - pass * nk1) * BKB;
      xo = kb + (pass == 0 ? dxl : 0);
      wo = kb + (pass == 1 ? dwl : 0);
    } else {
      xo = wo = (int64_t)kt * BKB;
    }
  };
  u32x4 rx[4], rw[4];
  {
    int64_t xo, wo;
    src_off(0, xo, wo);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      rx[i] = *reinterpret_cast<const u32x4*>(xg[i] + xo);
      rw[i] = *reinterpret_cast<const u32x4*>(wg[i] + wo);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    *reinterpret_cast<u32x4*>(Xs + soff[i]) = rx[i];
    *reinterpret_cast<u32x4*>(Ws + soff[i]) = rw[i];
  }
  __syncthreads();
  tl_mark(1);

  const int sx = lr >> 1;                         // swizzle term of this lane's rows
  const int wrow = (wn * 64 + lr) * 128, xrow = (wm * 64 + lr) * 128;

  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + 1) < nk;
    if (more) {
      int64_t xo, wo;
      src_off(kt + 1, xo, wo);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        rx[i] = *reinterpret_cast<const u32x4*>(xg[i] + xo);
        rw[i] = *reinterpret_cast<const u32x4*>(wg[i] + wo);
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
        for (int mt = 0; mt < 4; ++mt) {
          if constexpr (SWAP) TT<T>::mma(acc[nt][mt], b[mt], a[nt]);
          else TT<T>::mma(acc[nt][mt], a[nt], b[mt]);
        }
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
// Written for memory-level parallelism (r02 finding, profiles/r02_gemm_epilogue_mlp.txt): the first version tested
// `if (p.bias)` / `if (inj)` / `if (m >= M) continue` inside the unrolled tile loops, hipcc turned every (mt, nt) step
// into load -> s_waitcnt vmcnt(0) -> use, and a 256 x 256 tile spent 10 us (bias), 13 us (GELU) or 36 us (residual)
// in its epilogue -- one exposed memory round trip per 16 x 16 block -- against 26 us of K = 1024 main loop.
// Now: the per-column vectors (bias, gamma) are loaded ONCE per wave, the row loop bodies are branch-free (row
// indices clamped for the loads, only the store is predicated), so the four residual / table loads of a row block
// -- and, registers permitting, the next row block's -- are in flight together.
template <typename T, int EPI, bool OUT_F32, int MT, bool INJECT, int XP = 0, bool X3 = false>   // XP = 1 (OVG_TILE_R02_EPILOGUE, A/B flag): the r02 erf_as GELU instead of the polynomial one; X3: split-f16 outputs (hi / lo planes), libm erff
OVG_DEV void linear_epilogue_impl(const ovg_linear_params& p, const f32x4 (&acc)[4][MT], const int m_w0, const int n_w0, const int row_lim = -1) {
  const int M = row_lim >= 0 ? row_lim : (int)p.M, N = (int)p.N;   // row_lim: rows >= it belong to somebody else (unused by the shipped kernels: -1)
  const int lane = threadIdx.x & 63, g = lane >> 4, lr = lane & 15;
  const int ncol = n_w0 + 4 * g;                       // this lane's first column of n-block 0; block nt adds 16 nt
  f32x4 bias[4], gam[4];
  if (p.bias != nullptr) {                             // wave-uniform, outside every loop
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bias[nt] = *reinterpret_cast<const f32x4*>(p.bias + ncol + nt * 16);
  } else {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bias[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if constexpr (EPI == OVG_EPI_RES) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) gam[nt] = *reinterpret_cast<const f32x4*>(p.gamma + ncol + nt * 16);
  }
  // row-block operands (residual / position-table rows, injection rows) are fetched AH ROW BLOCKS AHEAD by hand: the
  // output may alias the residual (fc2 runs in place), so the compiler may not move a later block's loads above an
  // earlier block's stores on its own; different row blocks never touch the same rows, so doing it by hand is safe.
  // AH: 0 in the 128 x 128 kernels (3 workgroups per CU hide the latency and must stay <= 168 VGPRs), 1 in the 256 x 256
  // kernels (1 workgroup per CU: nobody else hides it). r03 A/B (profiles/r03_gemm_epilogue_forms_ab.txt): a deeper register
  // ring does not fit beside the 128 accumulators (AH = 3 spilled 380 bytes), and fetching the residual tile by LDS-DMA
  // through the idle ring (16 KB per wave in flight, no registers) gained 0.4 % on proj and 0.8 % on fc2 -- the residual
  // epilogue is not bound by load latency but by the 0.9 GB it moves (3 TB/s averaged over the launch); removed again.
  constexpr bool kRowLoads = (EPI == OVG_EPI_RES || EPI == OVG_EPI_PATCH);
  constexpr int AH = !kRowLoads ? 0 : (MT == 8 ? 1 : 0);
  constexpr int NS = AH + 1;                            // ring slots; every index below is a compile-time constant after unrolling
  const FastDiv div_p0(EPI == OVG_EPI_PATCH ? (int)p.p0 : 1), div_per(INJECT ? (int)p.inj_period : 1);
  f32x4 ex[NS][4], inj[NS][4];
  float on[NS];
  int64_t orow_s[NS];
  auto fetch = [&](int mt, f32x4 (&exs)[4], f32x4 (&injs)[4], float& ons, int64_t& orow) {
    const int m = m_w0 + mt * 16 + lr;
    const int mc = m < M ? m : M - 1;                  // loads of dead rows read a live row instead of branching
    orow = mc;
    ons = 0.f;
    if constexpr (EPI == OVG_EPI_PATCH) {
      int vw, t;
      div_p0.divmod(mc, vw, t);
      orow = (int64_t)vw * p.p1 + p.row_off + t;
      const float* trow = p.table + (int64_t)(t + 1) * N + ncol;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) exs[nt] = *reinterpret_cast<const f32x4*>(trow + nt * 16);
    }
    if constexpr (EPI == OVG_EPI_RES) {
      const float* rrow = p.res + (int64_t)mc * p.ldres + ncol;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) exs[nt] = *reinterpret_cast<const f32x4*>(rrow + nt * 16);
      if constexpr (INJECT) {
        // camera-token injection (omnivggt_aggregator.py:284-301) on rows m % period == 0: every lane reads ITS view's
        // row (an L1 / L2 hit: 1 row per 1374) and scales it by 0 or 1 -- no divergent branch in the row loop
        int vw, rem;
        div_per.divmod(mc, vw, rem);
        ons = rem == 0 ? 1.0f : 0.0f;
        const float* irow = p.inject + (int64_t)vw * N + ncol;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) injs[nt] = *reinterpret_cast<const f32x4*>(irow + nt * 16);
      }
    }
  };
  if constexpr (kRowLoads) {
#pragma unroll
    for (int a = 0; a < AH; ++a) fetch(a, ex[a], inj[a], on[a], orow_s[a]);
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if constexpr (kRowLoads) {
      if (mt + AH < MT) fetch(mt + AH, ex[(mt + AH) % NS], inj[(mt + AH) % NS], on[(mt + AH) % NS], orow_s[(mt + AH) % NS]);
    }
    const int cs = mt % NS;                             // slot of this row block
    const bool ok = (m_w0 + mt * 16 + lr) < M;
    f32x4 v[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      v[nt] = acc[nt][mt] + bias[nt];
      if constexpr (EPI == OVG_EPI_GELU) {
        if constexpr (XP || sizeof(T) == 4 || X3) {  // f32 / split-f16 parity modes: libm erff; XP (A/B flag OVG_TILE_R02_EPILOGUE): the r02 erf_as form; else the polynomial form below
          using GT = typename std::conditional<X3, float, T>::type;
          v[nt][0] = gelu_erf<GT>(v[nt][0]); v[nt][1] = gelu_erf<GT>(v[nt][1]); v[nt][2] = gelu_erf<GT>(v[nt][2]); v[nt][3] = gelu_erf<GT>(v[nt][3]);
        }
        if constexpr (std::is_same<T, f16_t>::value && XP) {
          // f16 range guard: the hidden activation is the one 16-bit tensor fed by an unnormalised f32 sum (DINOv2-style
          // massive activations reach 1e3..1e4 after fc1); saturate at the largest finite f16 instead of storing +inf
          // (inf * 0-weight = NaN in fc2). GELU is bounded below by -0.17, so only the upper side needs it.
#pragma unroll
          for (int r = 0; r < 4; ++r) v[nt][r] = fminf(v[nt][r], 65504.0f);
        }
      }
      if constexpr (EPI == OVG_EPI_RES) v[nt] = ex[cs][nt] + gam[nt] * v[nt];
      if constexpr (EPI == OVG_EPI_RES && INJECT) v[nt] += on[cs] * inj[cs][nt];
      if constexpr (EPI == OVG_EPI_PATCH) v[nt] += ex[cs][nt];
    }
    if constexpr (EPI == OVG_EPI_GELU && !XP && !X3 && sizeof(T) == 2) {
      gelu_poly16(v);
      if constexpr (std::is_same<T, f16_t>::value) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[nt][r] = fminf(v[nt][r], 65504.0f);
      }
    }
    const int64_t orow = kRowLoads ? orow_s[cs] : (int64_t)(m_w0 + mt * 16 + lr);
    if (ok) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        if constexpr (OUT_F32 || EPI == OVG_EPI_RES || EPI == OVG_EPI_PATCH) {
          *reinterpret_cast<f32x4*>(static_cast<float*>(p.y) + orow * p.ldy + ncol + nt * 16) = v[nt];
        } else if constexpr (X3) {
          const int64_t off = orow * p.ldy + ncol + nt * 16;
          store4_hilo(static_cast<f16_t*>(p.y) + off, static_cast<f16_t*>(p.y_lo) + off, v[nt][0], v[nt][1], v[nt][2], v[nt][3]);
        } else {
          store4<T>(static_cast<T*>(p.y) + orow * p.ldy + ncol + nt * 16, v[nt][0], v[nt][1], v[nt][2], v[nt][3]);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Full-line stores through the idle LDS ring (256 x 256 kernels, r03).
// In the MFMA accumulator layout a lane owns 4 consecutive columns of ONE row, so a wave-wide store instruction of the register
// epilogues touches 16 rows x 32 B (16-bit outputs: 32 global_store_dwordx2 per lane, quarter lines) or 16 rows x 64 B (f32). The
// two-point fit over K (profiles/r02_gemm_persistent_ab.txt: 111.5 us for 128 k-stages, 211.4 us for 256) puts ~11.6 us of FIXED cost
// under every 256 x 256 tile, a third of a K = 1024 tile, and the persistent-stream experiment of this round
// (profiles/r03_gemm_persistent_ab.txt: dispatch, set-up and pipeline fill removed, +-2 %) showed that it is not the launch side:
// it is the store tail -- the guide prices a row-per-lane 8-byte store epilogue at ~7 B/cycle/CU, issue-bound, and halves it with
// 16-byte stores (T21). After the main loop the 128 KB ring is idle and every wave owns 16 KB of it = exactly its 64 x 128 block in
// 16 bits (two passes of 64 rows for f32): the lanes write their values into a row-major image (16-byte chunks XOR-swizzled with the
// row so the writes spread over the banks), and read it back ROW-wise -- 8 (16-bit) or 16 (f32) lanes per row, 16 bytes each -- so one
// store instruction writes 8 whole 128-byte lines (16-bit) or 4 x 256 B (f32), and the residual is LOADED in the same shape.
// Same wave writes and reads (LDS operations of a wave execute in order): no barrier.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T> OVG_DEV void stage16_put4(unsigned char* img, int row, int col, float a, float b, float c, float d) {   // image rows of 64 T (128 B)
  const int chunk = col >> 3, half = (col >> 2) & 1;
  store4<T>(reinterpret_cast<T*>(img + row * 128 + ((chunk ^ (row & 7)) << 4) + half * 8), a, b, c, d);
}
OVG_DEV u32x4 stage16_get(const unsigned char* img, int row, int chunk) {
  return *reinterpret_cast<const u32x4*>(img + row * 128 + ((chunk ^ (row & 7)) << 4));
}
OVG_DEV void stage32_put4(unsigned char* img, int row, int col, const f32x4 v) {                                            // image rows of 64 f32 (256 B)
  *reinterpret_cast<f32x4*>(img + row * 256 + (((col >> 2) ^ (row & 15)) << 4)) = v;
}
OVG_DEV f32x4 stage32_get(const unsigned char* img, int row, int chunk) {
  return *reinterpret_cast<const f32x4*>(img + row * 256 + ((chunk ^ (row & 15)) << 4));
}

// STORE / GELU with 16-bit output and RES (f32 output, no injection row among the wave's rows -- the caller checks) on a wave's
// 64 (n) x 16 MT (m) block; `img` = the wave's private 2 KB x MT of idle LDS (16 KB of the ring in the 256 x 256 kernels, 8 KB of the
// two stage buffers in the 128 x 128 kernels): the whole block in 16 bits, half of it per pass in f32.
template <typename T, int EPI, int MT>
OVG_DEV void linear_epilogue_staged(const ovg_linear_params& p, const f32x4 (&acc)[4][MT], const int m_w0, const int n_w0, unsigned char* img) {
  const int M = (int)p.M;
  const int lane = threadIdx.x & 63, g = lane >> 4, lr = lane & 15;
  const int ncol = n_w0 + 4 * g;
  f32x4 bias[4];
  if (p.bias != nullptr) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bias[nt] = *reinterpret_cast<const f32x4*>(p.bias + ncol + nt * 16);
  } else {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bias[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if constexpr (EPI == OVG_EPI_RES) {
    constexpr int HR = 8 * MT, HB = MT / 2, NI = HR / 4;        // rows per pass, 16-row blocks per pass, read-back instructions per pass
    f32x4 gam[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      gam[nt] = *reinterpret_cast<const f32x4*>(p.gamma + ncol + nt * 16);
      bias[nt] = gam[nt] * bias[nt];
    }
    const int prow = lane >> 4, pch = lane & 15;               // read-back shape: 4 rows x 16 chunks of 16 B (4 x 256 B) per instruction
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      // residual rows of this pass, in the read-back shape, requested before the image is written (independent of it)
      f32x4 res[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        int m = m_w0 + half * HR + i * 4 + prow;
        m = m < M ? m : M - 1;
        res[i] = *reinterpret_cast<const f32x4*>(p.res + (int64_t)m * p.ldres + n_w0 + pch * 4);
      }
#pragma unroll
      for (int ml = 0; ml < HB; ++ml)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) stage32_put4(img, ml * 16 + lr, nt * 16 + 4 * g, gam[nt] * acc[nt][half * HB + ml] + bias[nt]);
#pragma unroll
      for (int i0 = 0; i0 < NI; i0 += 4) {                      // read-backs in batches of four: four LDS latencies overlap instead of NI in a row
        f32x4 t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = stage32_get(img, (i0 + j) * 4 + prow, pch);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int m = m_w0 + half * HR + (i0 + j) * 4 + prow;
          if (m < M) *reinterpret_cast<f32x4*>(static_cast<float*>(p.y) + (int64_t)m * p.ldy + n_w0 + pch * 4) = res[i0 + j] + t[j];
        }
      }
    }
  } else {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      f32x4 v[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) v[nt] = acc[nt][mt] + bias[nt];
      if constexpr (EPI == OVG_EPI_GELU) {
        gelu_poly16(v);
        if constexpr (std::is_same<T, f16_t>::value) {          // f16 range guard (see linear_epilogue_impl)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[nt][r] = fminf(v[nt][r], 65504.0f);
        }
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) stage16_put4<T>(img, mt * 16 + lr, nt * 16 + 4 * g, v[nt][0], v[nt][1], v[nt][2], v[nt][3]);
    }
    const int prow = lane >> 3, pch = lane & 7;                 // read-back shape: 8 rows x 8 chunks of 16 B = 8 whole lines per instruction
#pragma unroll
    for (int i0 = 0; i0 < 2 * MT; i0 += 4) {
      u32x4 t[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = stage16_get(img, (i0 + j) * 8 + prow, pch);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = m_w0 + (i0 + j) * 8 + prow;
        if (m < M) *reinterpret_cast<u32x4*>(static_cast<T*>(p.y) + (int64_t)m * p.ldy + n_w0 + pch * 8) = t[j];
      }
    }
  }
}

template <typename T, int EPI, bool OUT_F32, int MT, int XP = 0, bool X3 = false>
OVG_DEV void linear_epilogue(const ovg_linear_params& p, const f32x4 (&acc)[4][MT], const int m_w0, const int n_w0, const int row_lim = -1) {
  if constexpr (EPI == OVG_EPI_RES) {
    if (p.inject != nullptr) { linear_epilogue_impl<T, EPI, OUT_F32, MT, true, XP, X3>(p, acc, m_w0, n_w0, row_lim); return; }
  }
  linear_epilogue_impl<T, EPI, OUT_F32, MT, false, XP, X3>(p, acc, m_w0, n_w0, row_lim);
}

// Which epilogue a wave of a 16-bit kernel takes (wave-uniform): the staged one unless the caller pinned the r02 register form (XP), the output
// cannot take 16-byte stores, or -- residual form -- one of the wave's rows is a camera-injection row (m % inj_period == 0: 1 row in 1374).
template <typename T, int EPI, bool OUT_F32, int MT, int XP, bool X3 = false>
OVG_DEV void linear_epilogue_auto(const ovg_linear_params& p, const f32x4 (&acc)[4][MT], const int m_w0, const int n_w0, unsigned char* img) {
  if constexpr (!XP && sizeof(T) == 2) {
    if constexpr ((EPI == OVG_EPI_STORE || EPI == OVG_EPI_GELU) && !OUT_F32 && !X3) {   // split-f16 outputs (two planes) keep the register form
      if (((p.ldy * (int64_t)sizeof(T)) & 15) == 0) { linear_epilogue_staged<T, EPI, MT>(p, acc, m_w0, n_w0, img); return; }
    }
    if constexpr (EPI == OVG_EPI_RES) {
      bool inj_here = false;
      if (p.inject != nullptr) {
        const int rem = m_w0 % (int)p.inj_period;
        inj_here = rem == 0 || (int)p.inj_period - rem < 16 * MT;
      }
      if (!inj_here) { linear_epilogue_staged<T, EPI, MT>(p, acc, m_w0, n_w0, img); return; }
    }
  }
  linear_epilogue<T, EPI, OUT_F32, MT, XP, X3>(p, acc, m_w0, n_w0);
}


template <typename T, int EPI, bool OUT_F32, int XP = 0, bool X3 = false>
__global__ __launch_bounds__(256, 2) void linear_kernel(ovg_linear_params p, int ntiles_n) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 128 * 128];
  const int M = (int)p.M, N = (int)p.N, K = (int)p.K;
  int tm, tn;
  tl_stagger();
  tl_mark(0);
  tile_coords(xcd_remap(blockIdx.x, gridDim.x), (M + BM - 1) / BM, ntiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  f32x4 acc[4][4];
  gemm_mainloop<T, false, X3>(static_cast<const T*>(p.x), p.ldx, static_cast<const T*>(p.w), p.ldw, M, N, K, m0, n0, lds, acc,
                              static_cast<const T*>(p.x_lo), static_cast<const T*>(p.w_lo));
  tl_mark(2);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // gemm_mainloop ends behind a __syncthreads: both stage buffers (32 KB) are idle, wave w owns 8 KB of them
  linear_epilogue_auto<T, EPI, OUT_F32, 4, XP, X3>(p, acc, m0 + (wave & 1) * 64, n0 + (wave >> 1) * 64, lds + wave * 8192);
  tl_mark(3);
}

// ---------------------------------------------------------------------------
// QKV epilogue on a wave's 64(n) x 16*MT(m) block (64 columns = one head of q, k or v):
// bias + per-head LayerNorm(64) + 2-D RoPE + q scale, head-major stores, V transposed
// ---------------------------------------------------------------------------
// Same discipline as linear_epilogue: everything that is uniform over the wave's rows (bias, q/k-norm affine rows,
// which of q / k / v this wave holds, whether norm / RoPE apply) is decided or loaded ONCE and the row loops are
// branch-free up to the predicated store. The RoPE cos / sin table (<= 128 positions x 16 frequencies, 16 KB) is
// copied into LDS when the kernel starts and read from there (first version: four dependent global loads per row
// block, one exposed L2 round trip each time, ~10 us per 256 x 256 tile).
template <typename T, int MT, bool NORM, bool ROPE, bool X3 = false>
OVG_DEV void qk_rows(const f32x4 (&acc)[4][MT], const float* __restrict__ nw_p, const float* __restrict__ nb_p,
                     const float* __restrict__ rope_cos, const float* __restrict__ rope_sin, T* __restrict__ out, const int64_t npad,
                     const int m_w0, const int M, const int seq, const int h, const int tokens_per_view, const int n_special,
                     const int grid_w, const float qk_eps, const float scale, unsigned char* img = nullptr, T* __restrict__ out_lo = nullptr) {
  // img != nullptr (16-bit modes): the wave's 2 KB x MT of idle LDS -- the 64-wide head rows (128 B = one line per token) are written into
  // a row-major image and stored as whole lines afterwards (see linear_epilogue_staged)
  const int lane = threadIdx.x & 63, g = lane >> 4, lr = lane & 15;
  const FastDiv div_seq(seq), div_tpv(ROPE ? tokens_per_view : 1), div_gw(ROPE ? grid_w : 1);
  float nw[16], nb[16];
  if constexpr (NORM) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(nw_p + nt * 16 + 4 * g);
      const f32x4 b = *reinterpret_cast<const f32x4*>(nb_p + nt * 16 + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) { nw[nt * 4 + r] = a[r]; nb[nt * 4 + r] = b[r]; }
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m_w0 + mt * 16 + lr;
    const bool valid = m < M;
    const int mm = valid ? m : M - 1;
    int bidx, n;
    div_seq.divmod(mm, bidx, n);
    f32x4 cy, sy, cx, sxn;
    if constexpr (ROPE) {
      int vw, t, py, px;
      div_tpv.divmod(mm, vw, t);
      const int pp = t - n_special;
      div_gw.divmod(pp >= 0 ? pp : 0, py, px);
      py = pp >= 0 ? py + 1 : 0;
      px = pp >= 0 ? px + 1 : 0;
      cy = *reinterpret_cast<const f32x4*>(rope_cos + py * 16 + 4 * g);
      sy = *reinterpret_cast<const f32x4*>(rope_sin + py * 16 + 4 * g);
      cx = *reinterpret_cast<const f32x4*>(rope_cos + px * 16 + 4 * g);
      sxn = *reinterpret_cast<const f32x4*>(rope_sin + px * 16 + 4 * g);
    }
    float v[16];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[nt * 4 + r] = acc[nt][mt][r];      // bias already added (qk_epilogue)
    if constexpr (NORM) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) s += v[i];
      s = quad16_sum(s);
      const float mean = s * (1.0f / 64.0f);
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) { const float d = v[i] - mean; q += d * d; }
      q = quad16_sum(q);
      float rstd;
      if constexpr (sizeof(T) == 4 || X3) rstd = 1.0f / sqrtf(q * (1.0f / 64.0f) + qk_eps);     // parity modes: IEEE sqrt + divide
      else rstd = __builtin_amdgcn_rsqf(q * (1.0f / 64.0f) + qk_eps);                     // 1 ulp, far below bf16 / f16 resolution
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = (v[i] - mean) * rstd * nw[i] + nb[i];
    }
    if constexpr (ROPE) {
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
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] *= scale;          // 1.0 for k (exact), q_scale for q
    if constexpr (sizeof(T) == 2) {
      if (img != nullptr) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) stage16_put4<T>(img, mt * 16 + lr, nt * 16 + 4 * g, v[nt * 4], v[nt * 4 + 1], v[nt * 4 + 2], v[nt * 4 + 3]);
        continue;
      }
    }
    if (valid) {
      const int64_t off = (((int64_t)bidx * OVG_H + h) * npad + n) * OVG_D + 4 * g;
      T* dst = out + off;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        if constexpr (X3) store4_hilo(dst + nt * 16, out_lo + off + nt * 16, v[nt * 4], v[nt * 4 + 1], v[nt * 4 + 2], v[nt * 4 + 3]);
        else store4<T>(dst + nt * 16, v[nt * 4], v[nt * 4 + 1], v[nt * 4 + 2], v[nt * 4 + 3]);
      }
    }
  }
  if constexpr (sizeof(T) == 2) {
    if (img != nullptr) {
      const int prow = lane >> 3, pch = lane & 7;               // 8 tokens x 8 chunks of 16 B per instruction: 8 whole 128-byte head rows
#pragma unroll
      for (int i0 = 0; i0 < 2 * MT; i0 += 4) {
        u32x4 t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = stage16_get(img, (i0 + j) * 8 + prow, pch);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int m = m_w0 + (i0 + j) * 8 + prow;
          int bidx, n;
          div_seq.divmod(m < M ? m : M - 1, bidx, n);
          if (m < M) *reinterpret_cast<u32x4*>(out + (((int64_t)bidx * OVG_H + h) * npad + n) * OVG_D + pch * 8) = t[j];
        }
      }
    }
  }
}

template <typename T, int MT, bool X3 = false>
OVG_DEV void qk_epilogue(const ovg_qkv_params& p, f32x4 (&acc)[4][MT], const int m_w0, const int ncol0_,
                         const float* rope_c, const float* rope_s, unsigned char* img = nullptr) {
  const int ncol0 = __builtin_amdgcn_readfirstlane(ncol0_);   // wave-uniform: keep the q/k/v dispatch scalar
  const int M = (int)p.M;
  const int lane = threadIdx.x & 63, g = lane >> 4;
  const int seq = (int)p.seq;
  const int which = ncol0 / OVG_C;                      // 0 q, 1 k (uniform per wave; V^T tiles go through v_epilogue)
  const int h = (ncol0 % OVG_C) / OVG_D;                // head of this wave's 64 columns

  // the bias goes into the accumulators once, up front: 16 registers fewer live across the row loop (r04: the 128 x 128 kernel sits at its
  // 168-VGPR cap -- __launch_bounds__(256, 3) -- and spilled 17 registers into scratch inside this epilogue)
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + ncol0 + nt * 16 + 4 * g);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[nt][mt] += b;
  }
  {
    const float* nw_p = which == 0 ? p.qn_w : p.kn_w;
    const float* nb_p = which == 0 ? p.qn_b : p.kn_b;
    T* out = static_cast<T*>(which == 0 ? p.q : p.k);
    T* out_lo = static_cast<T*>(which == 0 ? p.q_lo : p.k_lo);
    const int64_t npad = which == 0 ? p.nq_pad : p.nk_pad;
    const float scale = which == 0 ? p.q_scale : 1.0f;
    const int tpv = (int)p.tokens_per_view;
#define OVG_QK_ROWS(NORM, ROPE) qk_rows<T, MT, NORM, ROPE, X3>(acc, nw_p, nb_p, rope_c, rope_s, out, npad, m_w0, M, seq, h, \
                                                                 tpv, p.n_special, p.grid_w, p.qk_eps, scale, img, out_lo)
    if (p.qk_norm) { if (p.rope) OVG_QK_ROWS(true, true); else OVG_QK_ROWS(true, false); }
    else { if (p.rope) OVG_QK_ROWS(false, true); else OVG_QK_ROWS(false, false); }
#undef OVG_QK_ROWS
  }
}

// V^T tiles: the main loop ran with SWAP, so acc[nt][mt][r] = V[token m = m_w0 + 16 mt + 4g + r][feature d = 16 nt + (lane & 15)]:
// four consecutive tokens of one feature per register group = ONE 8-byte store into V^T [B*H, 64, nk_pad] (the first
// version held 4 features of one token and issued 16 two-byte stores per row block, each block fenced by a vmcnt(0)).
template <typename T, int MT, bool X3 = false>
OVG_DEV void v_epilogue(const ovg_qkv_params& p, const f32x4 (&acc)[4][MT], const int m_w0, const int ncol0_, const int row_lim = -1) {
  const int ncol0 = __builtin_amdgcn_readfirstlane(ncol0_);
  const int M = row_lim >= 0 ? row_lim : (int)p.M, seq = (int)p.seq;
  const int lane = threadIdx.x & 63, g = lane >> 4, lr = lane & 15;
  const int h = (ncol0 % OVG_C) / OVG_D;
  const FastDiv div_seq(seq);
  float bias[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) bias[nt] = p.bias[ncol0 + nt * 16 + lr];
  T* vt = static_cast<T*>(p.vt);
  const int64_t dlo = X3 ? static_cast<T*>(p.vt_lo) - vt : 0;      // element offset of the lo plane (split-f16)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m_w0 + mt * 16 + 4 * g;               // first of this lane's 4 tokens
    const int mc = m < M ? m : M - 1;
    int bidx, n;
    div_seq.divmod(mc, bidx, n);
    // all four tokens valid and in the same sequence, and the vector store naturally aligned to its element group
    // 16-bit V^T rows hold their keys in the vt_pos16 order (ovg_common.h): the four tokens n .. n + 3 are one 8-byte group when
    // n % 4 == 0 and two 4-byte halves of neighbouring groups when n % 4 == 2 (odd views at 518^2: 1374 = 2 mod 4)
    const bool inside = (m + 3 < M) && (n + 3 < seq);
    const bool whole = inside && ((n & 3) == 0);
    const bool halves = sizeof(T) == 2 && inside && ((n & 3) == 2);
    T* rowb = vt + (((int64_t)bidx * OVG_H + h) * OVG_D + lr) * p.nk_pad;
    const int c0 = sizeof(T) == 2 ? vt_pos16(n) : n, c2 = sizeof(T) == 2 ? vt_pos16(n + 2) : n + 2;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const f32x4 v = acc[nt][mt] + bias[nt];
      T* dst = rowb + (int64_t)nt * 16 * p.nk_pad;
      if (whole) {
        if constexpr (X3) store4_hilo(dst + c0, dst + dlo + c0, v[0], v[1], v[2], v[3]);
        else store4<T>(dst + c0, v[0], v[1], v[2], v[3]);
      } else if (halves) {
        if constexpr (X3) {
          f16_t ph[4], pl[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) split_hilo(v[r], ph[r], pl[r]);
          uint32_t w0, w1, l0, l1;
          __builtin_memcpy(&w0, ph, 4); __builtin_memcpy(&w1, ph + 2, 4);
          __builtin_memcpy(&l0, pl, 4); __builtin_memcpy(&l1, pl + 2, 4);
          *reinterpret_cast<uint32_t*>(dst + c0) = w0;
          *reinterpret_cast<uint32_t*>(dst + c2) = w1;
          *reinterpret_cast<uint32_t*>(dst + dlo + c0) = l0;
          *reinterpret_cast<uint32_t*>(dst + dlo + c2) = l1;
        } else if constexpr (sizeof(T) == 2) {
          T pr[4] = {TT<T>::from_f32(v[0]), TT<T>::from_f32(v[1]), TT<T>::from_f32(v[2]), TT<T>::from_f32(v[3])};
          uint32_t w0, w1;
          __builtin_memcpy(&w0, pr, 4);
          __builtin_memcpy(&w1, pr + 2, 4);
          *reinterpret_cast<uint32_t*>(dst + c0) = w0;
          *reinterpret_cast<uint32_t*>(dst + c2) = w1;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int mr = m + r;
          if (mr < M) {
            int br, nr;
            div_seq.divmod(mr, br, nr);
            const int64_t at = (((int64_t)br * OVG_H + h) * OVG_D + nt * 16 + lr) * p.nk_pad + (sizeof(T) == 2 ? vt_pos16(nr) : nr);
            if constexpr (X3) { f16_t eh, el; split_hilo(v[r], eh, el); vt[at] = eh; vt[at + dlo] = el; }
            else vt[at] = TT<T>::from_f32(v[r]);
          }
        }
      }
    }
  }
}

// RoPE table -> LDS by LDS-DMA (no VGPR round trip, nothing waits for it: it lands long before the epilogue, and being
// issued BEFORE the main loop's stage DMAs it never disturbs their counted vmcnt waits). Layout: cos rows at [0, 8 KB),
// sin rows at [8 KB, 16 KB), 64 B per position; positions >= max_pos are filled with a clamped (valid) source address.
OVG_DEV void stage_rope_table(const ovg_qkv_params& p, unsigned char* tab, int nwaves) {
  if (!p.rope) return;
  typedef __attribute__((address_space(1))) const void* gp_t;
  typedef __attribute__((address_space(3))) void* lp_t;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bytes = p.max_pos * 64;
  for (int i = wave; i < 16; i += nwaves) {               // 16 pieces of 1 KB: 8 of cos, 8 of sin
    const int piece = i & 7;
    int off = piece * 1024 + lane * 16;
    off = off < bytes ? off : bytes - 16;
    const unsigned char* src = reinterpret_cast<const unsigned char*>(i < 8 ? p.rope_cos : p.rope_sin) + off;
    __builtin_amdgcn_global_load_lds((gp_t)src, (lp_t)(tab + i * 1024), 16, 0, 0);
  }
}
constexpr int ROPE_LDS_BYTES = 2 * 128 * 16 * 4;          // max_pos <= 128 (ovg_qkv checks)

template <typename T, bool X3 = false>
__global__ __launch_bounds__(256, 3) void qkv_kernel(ovg_qkv_params p, int nt_begin, int nt_count) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 128 * 128];
  constexpr int N = 3 * OVG_C, K = OVG_C;
  const int M = (int)p.M;
  int tm, tn;
  tl_stagger();
  tl_mark(0);
  tile_coords(xcd_remap(blockIdx.x, gridDim.x), (M + BM - 1) / BM, nt_count, tm, tn);   // nt_count carries GM in its high half
  const int m0 = tm * BM, n0 = (nt_begin + tn) * BN;
  const int wave = threadIdx.x >> 6;
  f32x4 acc[4][4];
  if (n0 >= 2 * OVG_C) {                                   // V^T tile (workgroup-uniform): transposed accumulators
    gemm_mainloop<T, true, X3>(static_cast<const T*>(p.x), p.ldx, static_cast<const T*>(p.w), (int64_t)K, M, N, K, m0, n0, lds, acc,
                               static_cast<const T*>(p.x_lo), static_cast<const T*>(p.w_lo));
    tl_mark(2);
    v_epilogue<T, 4, X3>(p, acc, m0 + (wave & 1) * 64, n0 + (wave >> 1) * 64);
  } else {
    // 3 workgroups per CU hide the RoPE-table round trips here: the table is read from global memory (L1 / L2 hits); staging it
    // in LDS cost more than it saved on these small tiles (the DMA sits in front of the register-staged loop's first loads)
    gemm_mainloop<T, false, X3>(static_cast<const T*>(p.x), p.ldx, static_cast<const T*>(p.w), (int64_t)K, M, N, K, m0, n0, lds, acc,
                                static_cast<const T*>(p.x_lo), static_cast<const T*>(p.w_lo));
    // behind the main loop's last __syncthreads the two stage buffers are idle: wave w stages its head rows through 8 KB of them
    // (the OVG_TILE_R02_EPILOGUE flag keeps the r02 per-lane 8-byte stores: A/B; the split-f16 mode writes two planes from registers)
    tl_mark(2);
    unsigned char* img = (sizeof(T) == 2 && !X3 && !(p.tile & OVG_TILE_R02_EPILOGUE)) ? lds + __builtin_amdgcn_readfirstlane(wave) * 8192 : nullptr;
    qk_epilogue<T, 4, X3>(p, acc, m0 + (wave & 1) * 64, n0 + (wave >> 1) * 64, p.rope_cos, p.rope_sin, img);
  }
  tl_mark(3);
}

#include "ovg_gemm256.h"

// 256 x 256 variants (16-bit modes): same epilogues on acc[4][8]
template <typename T, int EPI, bool OUT_F32, int XP = 0, bool X3 = false>
__global__ __launch_bounds__(512) void linear256_kernel(ovg_linear_params p, int ntiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds256[];
  const int M = (int)p.M, N = (int)p.N, K = (int)p.K;
  int tm, tn;
  tl_stagger();
  tl_mark(0);
  tile_coords(xcd_remap(blockIdx.x, gridDim.x), (M + g256::BM2 - 1) / g256::BM2, ntiles_n, tm, tn);
  const int m0 = tm * g256::BM2, n0 = tn * g256::BN2;
  f32x4 acc[4][8];
  g256::mainloop<T, false, X3>(static_cast<const T*>(p.x), p.ldx, static_cast<const T*>(p.w), p.ldw, M, N, K, m0, n0, lds256, acc,
                                static_cast<const T*>(p.x_lo), static_cast<const T*>(p.w_lo));
  tl_mark(2);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // the ring is idle: mainloop() returns behind its last barrier, every DMA waited for; wave w owns 16 KB of it
  linear_epilogue_auto<T, EPI, OUT_F32, 8, XP, X3>(p, acc, m0 + (wave >> 2) * 128, n0 + (wave & 3) * 64, lds256 + wave * 16384);
  tl_mark(3);
}

template <typename T, bool X3 = false>
__global__ __launch_bounds__(512) void qkv256_kernel(ovg_qkv_params p, int nt_begin, int nt_count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds256[];   // ring (g256::LDS_BYTES) + RoPE table (ROPE_LDS_BYTES)
  constexpr int N = 3 * OVG_C, K = OVG_C;
  const int M = (int)p.M;
  int tm, tn;
  tl_stagger();
  tl_mark(0);
  tile_coords(xcd_remap(blockIdx.x, gridDim.x), (M + g256::BM2 - 1) / g256::BM2, nt_count, tm, tn);
  const int m0 = tm * g256::BM2, n0 = (nt_begin + tn) * g256::BN2;
  const int wave = threadIdx.x >> 6;
  f32x4 acc[4][8];
  if (n0 >= 2 * OVG_C) {                                   // V^T tile (workgroup-uniform): transposed accumulators
    g256::mainloop<T, true, X3>(static_cast<const T*>(p.x), p.ldx, static_cast<const T*>(p.w), (int64_t)K, M, N, K, m0, n0, lds256, acc,
                                 static_cast<const T*>(p.x_lo), static_cast<const T*>(p.w_lo));
    tl_mark(2);
    v_epilogue<T, 8, X3>(p, acc, m0 + (wave >> 2) * 128, n0 + (wave & 3) * 64);
  } else {
    float* rope_tab = reinterpret_cast<float*>(lds256 + g256::LDS_BYTES);
    stage_rope_table(p, lds256 + g256::LDS_BYTES, 8);      // older than every stage DMA: retired by the loop's first counted wait, visible after its barriers
    g256::mainloop<T, false, X3>(static_cast<const T*>(p.x), p.ldx, static_cast<const T*>(p.w), (int64_t)K, M, N, K, m0, n0, lds256, acc,
                                  static_cast<const T*>(p.x_lo), static_cast<const T*>(p.w_lo));
    // the OVG_TILE_R02_EPILOGUE flag keeps the r02 per-lane 8-byte stores (A/B); otherwise whole head rows through the idle ring
    unsigned char* img = (X3 || (p.tile & OVG_TILE_R02_EPILOGUE)) ? nullptr : lds256 + __builtin_amdgcn_readfirstlane(wave) * 16384;
    tl_mark(2);
    qk_epilogue<T, 8, X3>(p, acc, m0 + (wave >> 2) * 128, n0 + (wave & 3) * 64, rope_tab, rope_tab + 128 * 16, img);
  }
  tl_mark(3);
}

// Opt a kernel in to > 64 KB of dynamic LDS. The attribute is PER DEVICE, so it is set once per (kernel, device) -- a process
// that drives a second GPU must not inherit the first one's "already done" (round-2 review: function-local statics did that).
// The only state is this idempotent memo of calls already made.
template <typename KernelT>
int allow_big_lds(KernelT kernel, int bytes = g256::LDS_BYTES) {
  static std::mutex mu;
  static std::vector<std::pair<const void*, int>> done;
  const void* fn = reinterpret_cast<const void*>(kernel);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return OVG_E_LAUNCH;
  std::lock_guard<std::mutex> lock(mu);
  for (const auto& d : done)
    if (d.first == fn && d.second == dev) return OVG_OK;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return OVG_E_LAUNCH;
  done.emplace_back(fn, dev);
  return OVG_OK;
}

// tile-order group sizes (m-tiles per group, tile_coords): measured in profiles/r01_gemm_tile_order_ab.txt / r01_gemm256_ab.txt
constexpr int TILE_GROUP = 8, TILE_GROUP256 = 4;

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename T, int XP>
int launch_linear128_xp(const ovg_linear_params& p, hipStream_t st) {
  const int mt = (int)((p.M + BM - 1) / BM), nt = (int)(p.N / BN);
  const dim3 grid(mt * nt), block(256);
  const int ntg = nt | (TILE_GROUP << 16);
  switch (p.epilogue) {
    case OVG_EPI_STORE:
      if (p.out_f32) OVG_LAUNCH((linear_kernel<T, OVG_EPI_STORE, true, XP>), grid, block, 0, st, p, ntg);
      else OVG_LAUNCH((linear_kernel<T, OVG_EPI_STORE, false, XP>), grid, block, 0, st, p, ntg);
      break;
    case OVG_EPI_GELU:
      OVG_LAUNCH((linear_kernel<T, OVG_EPI_GELU, false, XP>), grid, block, 0, st, p, ntg);
      break;
    case OVG_EPI_RES:
      OVG_LAUNCH((linear_kernel<T, OVG_EPI_RES, true, XP>), grid, block, 0, st, p, ntg);
      break;
    case OVG_EPI_PATCH:
      OVG_LAUNCH((linear_kernel<T, OVG_EPI_PATCH, true, XP>), grid, block, 0, st, p, ntg);
      break;
    default: return OVG_E_ARG;
  }
  OVG_CHECK_LAUNCH();
  return OVG_OK;
}
template <typename T>
int launch_linear128(const ovg_linear_params& p, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
#ifdef OVG_AB_VARIANTS
    if (p.tile & OVG_TILE_R02_EPILOGUE) return launch_linear128_xp<T, 1>(p, st);      // A/B flag: the r02 epilogue forms
#else
    if (p.tile & OVG_TILE_R02_EPILOGUE) return OVG_E_UNSUPPORTED;                      // A/B history: only in -DOVG_AB_VARIANTS builds
#endif
  }
  return launch_linear128_xp<T, 0>(p, st);
}
template <typename T, int EPI, bool OUT_F32, int XP = 0>
int launch_linear256_one(const ovg_linear_params& p, hipStream_t st) {
  const int ok = allow_big_lds(linear256_kernel<T, EPI, OUT_F32, XP>);
  if (ok != OVG_OK) return ok;
  const int mt = (int)((p.M + g256::BM2 - 1) / g256::BM2), nt = (int)(p.N / g256::BN2);
  const int ntg = nt | (TILE_GROUP256 << 16);
  OVG_LAUNCH((linear256_kernel<T, EPI, OUT_F32, XP>), dim3(mt * nt), dim3(512), g256::LDS_BYTES, st, p, ntg);
  OVG_CHECK_LAUNCH();
  return OVG_OK;
}
template <typename T>
int launch_linear256(const ovg_linear_params& p, hipStream_t st, bool xp) {
#ifdef OVG_AB_VARIANTS
  if (xp) {                                          // A/B flag: the r02 epilogue forms
    switch (p.epilogue) {
      case OVG_EPI_STORE: return p.out_f32 ? launch_linear256_one<T, OVG_EPI_STORE, true>(p, st) : launch_linear256_one<T, OVG_EPI_STORE, false, 1>(p, st);
      case OVG_EPI_GELU: return launch_linear256_one<T, OVG_EPI_GELU, false, 1>(p, st);
      case OVG_EPI_RES: return launch_linear256_one<T, OVG_EPI_RES, true, 1>(p, st);
      case OVG_EPI_PATCH: return launch_linear256_one<T, OVG_EPI_PATCH, true>(p, st);
      default: return OVG_E_ARG;
    }
  }
#else
  if (xp) return OVG_E_UNSUPPORTED;                  // A/B history: only in -DOVG_AB_VARIANTS builds
#endif
  switch (p.epilogue) {
    case OVG_EPI_STORE: return p.out_f32 ? launch_linear256_one<T, OVG_EPI_STORE, true>(p, st) : launch_linear256_one<T, OVG_EPI_STORE, false>(p, st);
    case OVG_EPI_GELU: return launch_linear256_one<T, OVG_EPI_GELU, false>(p, st);
    case OVG_EPI_RES: return launch_linear256_one<T, OVG_EPI_RES, true>(p, st);
    case OVG_EPI_PATCH: return launch_linear256_one<T, OVG_EPI_PATCH, true>(p, st);
    default: return OVG_E_ARG;
  }
}
// Tile choice for the 16-bit modes. Isolated A/B (tests/bench_kernels.py gemm, profiles/r02_gemm_epilogue_mlp.txt): with the
// r02 epilogues the 256 x 256 ping-pong loop wins on QKV / fc1 / fc2 at both bench sizes (M = 10 992: +7 / +13 / +19 %,
// M = 87 936: +28 / +5 / +17 %); the proj GEMM (K = 1024, f32 residual epilogue) is a tie at M = 87 936 and better on
// 128 x 128 below. IN SITU the picture differs for short token slices: at 8 views the whole forward is 3 % FASTER with
// 128 x 128 everywhere (44.4 vs 45.7 ms, profiles/r02_gemm_tile_in_situ.txt) -- the global-attention launches that follow
// the denser 256 x 256 GEMMs run 9 % slower (0.533 vs 0.488 ms: the chip is power-limited, rocprofv3 shows the GEMMs at
// 1.8-1.9 GHz and attention at 2.1-2.2 GHz, and the GEMMs' own in-situ gain shrinks with cold L2s and one workgroup per
// CU). Re-measured with the LDS-DMA attention kernels (profiles/r02_gemm_tile_in_situ.txt, second block): 8 views 190.1 (128 x 128) vs
// 189.0 frames/s, 16 views + aux 155.5 vs 156.8 (256 x 256) -- so the 256 x 256 kernels are used from M >= 20 000 rows (16 views) on,
// where they are worth +1 % (16 views) ... +2 % (64 views) on the forward. Re-checked in r03 with the staged-store epilogues
// (profiles/r03_gemm_mlp256_insitu.txt): fc1 / fc2 alone on 256 x 256 below the threshold -- isolated +8 / +17 % at M = 10 992 -- still
// LOSES in situ (4 / 8 / 12 views: -6.5 / -4 / -2.5 %; the global attention behind them 0.445 -> 0.482 ms): the threshold stays.
// Returns 1 = use 256^2, 0 = use 128^2, -1 = the caller forced a tile this shape / dtype cannot run.
// Short launches (round 6, profiles/r06_gemm_short_launch_tiles_insitu.txt). Below 20 000 rows the 128 x 128 kernels stay the default -- fc1 on
// 256 x 256 tiles at 8 views is isolated +21 % and in the forward -4.8 % (the global attention behind it clocks 0.442 -> 0.480 ms) -- with two
// exceptions where the 256 x 256 launch fits ONE round of the chip (tiles <= CUs):
//   fc2 (residual epilogue, K >= 2048) from 8 000 rows: 6 / 8 / 10 / 11 views +3.0 / +0.75 / +5.1 / +4.5 % on the forward (a K = 4096 tile amortises its
//       fixed cost, and at 9-11 views the 128 x 128 launch needs a second round of 3 x CUs slots for an eighth of one); 12 views (260 tiles: two
//       rounds) -1.5 %, so not beyond one round;
//   proj (residual epilogue, K = 1024) only where the 128 x 128 launch would spill into such a second round (tiles128 > 3 x CUs).
// kind: 0 = other, 1 = fc1 (GELU), 2 = fc2 (RES, K >= 2048), 3 = proj (RES, K < 2048)
#ifndef OVG_PROJ_256_ONE_ROUND
#define OVG_PROJ_256_ONE_ROUND 1
#endif
int gemm_device_cus() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    return v;
  }();
  return n;
}
int choose_256(int tile_arg, bool sixteen_bit, int64_t M, int64_t N, int64_t K, bool light_epilogue_or_long_k, int kind = 0) {
  const int tile = tile_arg & ~OVG_TILE_R02_EPILOGUE;      // the A/B flag does not take part in the tile choice
  const bool legal = sixteen_bit && N % g256::BN2 == 0 && K % 64 == 0;
  if (tile == OVG_TILE_128) return 0;
  if (tile == OVG_TILE_256) return legal ? 1 : -1;
  if (tile != OVG_TILE_AUTO) return -1;
  if (!legal) return 0;
  if (M >= 20000) return light_epilogue_or_long_k ? 1 : 0;
  const int64_t cus = gemm_device_cus();
  const int64_t t256 = ((M + 255) / 256) * (N / 256), t128 = ((M + 127) / 128) * (N / 128);
  if (kind == 2 && M >= 8000 && t256 <= cus) return 1;
  if (OVG_PROJ_256_ONE_ROUND && kind == 3 && t256 <= cus && t128 > 3 * cus) return 1;
  return 0;
}

template <typename T>
int launch_linear(const ovg_linear_params& p, hipStream_t st) {
  const int big = choose_256(p.tile, sizeof(T) == 2, p.M, p.N, p.K, p.epilogue != OVG_EPI_RES || p.K >= 2048,
                             p.epilogue == OVG_EPI_GELU ? 1 : (p.epilogue == OVG_EPI_RES ? (p.K >= 2048 ? 2 : 3) : 0));
  if (big < 0) return OVG_E_ARG;
  if constexpr (sizeof(T) == 2) {
    if (big) return launch_linear256<T>(p, st, (p.tile & OVG_TILE_R02_EPILOGUE) != 0);
  }
  return launch_linear128<T>(p, st);
}

// split-f16 mode (OVG_F16X2): the same kernels with X3 = true on f16 planes; 256 x 256 tiles from M >= 20 000 rows on (three times the
// main loop per epilogue: the ping-pong loop's advantage grows), register-form epilogues for the two-plane 16-bit outputs
template <int EPI, bool OUT_F32>
int launch_linear_x3_one(const ovg_linear_params& p, hipStream_t st, bool big) {
  if (big) {
    const int ok = allow_big_lds(linear256_kernel<f16_t, EPI, OUT_F32, 0, true>);
    if (ok != OVG_OK) return ok;
    const int mt = (int)((p.M + g256::BM2 - 1) / g256::BM2), nt = (int)(p.N / g256::BN2);
    OVG_LAUNCH((linear256_kernel<f16_t, EPI, OUT_F32, 0, true>), dim3(mt * nt), dim3(512), g256::LDS_BYTES, st, p, nt | (TILE_GROUP256 << 16));
  } else {
    const int mt = (int)((p.M + BM - 1) / BM), nt = (int)(p.N / BN);
    OVG_LAUNCH((linear_kernel<f16_t, EPI, OUT_F32, 0, true>), dim3(mt * nt), dim3(256), 0, st, p, nt | (TILE_GROUP << 16));
  }
  OVG_CHECK_LAUNCH();
  return OVG_OK;
}
int launch_linear_x3(const ovg_linear_params& p, hipStream_t st) {
  int big = choose_256(p.tile, true, p.M, p.N, p.K, true);
  if (big < 0) return OVG_E_ARG;
  switch (p.epilogue) {
    case OVG_EPI_STORE: return p.out_f32 ? launch_linear_x3_one<OVG_EPI_STORE, true>(p, st, big) : launch_linear_x3_one<OVG_EPI_STORE, false>(p, st, big);
    case OVG_EPI_GELU: return launch_linear_x3_one<OVG_EPI_GELU, false>(p, st, big);
    case OVG_EPI_RES: return launch_linear_x3_one<OVG_EPI_RES, true>(p, st, big);
    case OVG_EPI_PATCH: return launch_linear_x3_one<OVG_EPI_PATCH, true>(p, st, big);
    default: return OVG_E_ARG;
  }
}

#ifdef OVG_GEMM_TIMELINE
}  // namespace
extern "C" int ovg_lab_gemm_timeline(void* buf, int stagger_ticks) {
  unsigned long long* b = static_cast<unsigned long long*>(buf);
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_tl_buf), &b, sizeof(b)) != hipSuccess) return OVG_E_LAUNCH;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_tl_stagger), &stagger_ticks, sizeof(int)) != hipSuccess) return OVG_E_LAUNCH;
  return OVG_OK;
}
namespace {
#endif

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
    case OVG_F16X2: {
      if (!p->x_lo || !p->w_lo || !aligned16(p->x_lo) || !aligned16(p->w_lo)) return OVG_E_ARG;
      const bool out16 = (p->epilogue == OVG_EPI_STORE || p->epilogue == OVG_EPI_GELU) && !p->out_f32;
      if (out16 && (!p->y_lo || !aligned16(p->y_lo))) return OVG_E_ARG;
      return launch_linear_x3(*p, st);
    }
    default: return OVG_E_DTYPE;
  }
}

extern "C" int ovg_qkv(const ovg_qkv_params* p, void* stream) {
  if (!p || !p->x || !p->w || !p->bias || !p->q || !p->k || !p->vt) return OVG_E_ARG;
  if (p->M <= 0 || p->M > (1 << 30) || p->seq <= 0 || p->M % p->seq != 0) return OVG_E_ARG;
  if (p->nq_pad < p->seq || p->nk_pad < p->seq || p->nk_pad % OVG_KV_TILE != 0) return OVG_E_ARG;
  if (p->dtype != OVG_BF16 && p->dtype != OVG_F16 && p->dtype != OVG_F32 && p->dtype != OVG_F16X2) return OVG_E_DTYPE;
  const int64_t esz = p->dtype == OVG_F32 ? 4 : 2;
  if ((p->ldx * esz) % 16 || !aligned16(p->x) || !aligned16(p->w) || !aligned16(p->bias) || !aligned16(p->q) || !aligned16(p->k) || !aligned16(p->vt)) return OVG_E_ARG;
  if (p->dtype == OVG_F16X2) {
    if (!p->x_lo || !p->w_lo || !aligned16(p->x_lo) || !aligned16(p->w_lo)) return OVG_E_ARG;
    if ((p->part != 2 && (!p->k_lo || !p->vt_lo || !aligned16(p->k_lo) || !aligned16(p->vt_lo))) || (p->part != 1 && (!p->q_lo || !aligned16(p->q_lo)))) return OVG_E_ARG;
  }
  if (p->qk_norm && (!p->qn_w || !p->qn_b || !p->kn_w || !p->kn_b)) return OVG_E_ARG;
  if (p->rope) {
    if (!p->rope_cos || !p->rope_sin || p->tokens_per_view <= 0 || p->grid_w <= 0) return OVG_E_ARG;
    const int64_t np = p->tokens_per_view - p->n_special;
    if (np <= 0 || (np - 1) / p->grid_w + 1 >= p->max_pos || p->grid_w >= p->max_pos) return OVG_E_ARG;
    if (p->max_pos > 128 || !aligned16(p->rope_cos) || !aligned16(p->rope_sin)) return OVG_E_ARG;   // the table is staged in 16 KB of LDS
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
      const int ok = allow_big_lds(qkv256_kernel<bf16_t>, g256::LDS_BYTES + ROPE_LDS_BYTES);
      if (ok != OVG_OK) return ok;
      OVG_LAUNCH((qkv256_kernel<bf16_t>), grid2, dim3(512), g256::LDS_BYTES + ROPE_LDS_BYTES, st, *p, ntb, ntg2);
    } else if (p->dtype == OVG_F16X2) {
      const int ok = allow_big_lds(qkv256_kernel<f16_t, true>, g256::LDS_BYTES + ROPE_LDS_BYTES);
      if (ok != OVG_OK) return ok;
      OVG_LAUNCH((qkv256_kernel<f16_t, true>), grid2, dim3(512), g256::LDS_BYTES + ROPE_LDS_BYTES, st, *p, ntb, ntg2);
    } else {
      const int ok = allow_big_lds(qkv256_kernel<f16_t>, g256::LDS_BYTES + ROPE_LDS_BYTES);
      if (ok != OVG_OK) return ok;
      OVG_LAUNCH((qkv256_kernel<f16_t>), grid2, dim3(512), g256::LDS_BYTES + ROPE_LDS_BYTES, st, *p, ntb, ntg2);
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
    case OVG_F16X2: OVG_LAUNCH((qkv_kernel<f16_t, true>), grid, block, 0, st, *p, nt_begin, ntg); break;
    default: OVG_LAUNCH((qkv_kernel<float>), grid, block, 0, st, *p, nt_begin, ntg); break;
  }
  OVG_CHECK_LAUNCH();
  return OVG_OK;
}
