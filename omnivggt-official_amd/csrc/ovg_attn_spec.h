// attn_spec_kernel: speculative "anchored" softmax on top of the attn3 structure (included by
// ovg_attn.hip after ovg_attn_v3.h).
//
// Why: PMC on attn3 shows the VALU (2.55 instructions per MFMA) and the issue stalls behind the
// serial   QK^T -> row max -> cross-lane max -> (rare) rescale -> exp   chain as the limiter, not
// the matrix pipe. The running max exists only to keep exp2() in range. In bf16 mode P has the f32
// exponent range, the accumulators are f32, and the result O = (sum_k P v) / (sum_k P) is invariant
// to the reference m_ref, so ANY per-row m_ref works as long as nothing leaves the f32 range.
//
// Fast pass (SM = 2): m_ref = a max over the FIRST key tile only (a prologue, outside the loop);
// every tile is then  S' = K Q^T - m_ref (m_ref rides in the MFMA C operand), P = exp2(S'),
// O^T += V^T P^T, l += 1^T P^T  -- no max, no cross-lane traffic, no branches in the tile body.
// Verification: after the last tile every row checks, on raw bits (this file is compiled with
// -fno-honor-nans), that l is finite and in [2^-100, 2^100] and that O is finite. That is exactly
// the condition under which no exp2 overflowed and the row did not flush to zero (l >= the largest
// P of the row). If ANY row of the workgroup fails (workgroup-uniform via __syncthreads_or), the
// whole workgroup recomputes with the lazy-rescale online softmax (SM = 0, the attn3 body) -- so the
// result is always the exact softmax; a failed speculation only costs time. A row needs a logit
// spread of more than ~100 log2 units (e^69) against its anchor to fail.
//
// f16: P overflows at 2^16 and loses precision below 2^-14, so its anchor sits ANCHOR_MARGIN = 4 log2 units
// above the first-tile max (P <= 2^-4 there, 20 units of head-room before the f16 conversion saturates to inf;
// an inf P makes l inf and triggers the fallback like any other failed speculation).
//
// Template knobs (measured against each other with tests/bench_kernels.py):
//   ANCHOR 0 none (m_ref = 0) | 1 per row | 2 one per lane, shared by the QB rows a lane owns (4
//          registers instead of 4 QB for the C operand)
//   DMA    0 register-staged K / V^T tiles, V^T written key-permuted (b128 fragment reads)
//          1 LDS-DMA (global_load_lds_dwordx4), V^T in natural key order (two b64 reads per fragment)
//   HALF   0 one 64-key body | 1 two 32-key halves (half of S' live at a time)
#pragma once

namespace spec {

constexpr uint32_t EXP_HI = 127 + 100, EXP_LO = 127 - 100;

OVG_DEV bool bad_sum(float l) {
  const uint32_t e = (__builtin_bit_cast(uint32_t, l) >> 23) & 0xffu;
  return e > EXP_HI || e < EXP_LO;
}
OVG_DEV bool nonfinite(float x) { return ((__builtin_bit_cast(uint32_t, x) >> 23) & 0xffu) == 0xffu; }

template <typename T>
OVG_DEV u32x4 pack2(const f32x4 a, const f32x4 b) {
  T v[8];
  v[0] = TT<T>::from_f32(a[0]); v[1] = TT<T>::from_f32(a[1]); v[2] = TT<T>::from_f32(a[2]); v[3] = TT<T>::from_f32(a[3]);
  v[4] = TT<T>::from_f32(b[0]); v[5] = TT<T>::from_f32(b[1]); v[6] = TT<T>::from_f32(b[2]); v[7] = TT<T>::from_f32(b[3]);
  u32x4 r;
  __builtin_memcpy(&r, v, 16);
  return r;
}

template <bool B> struct Tag { static constexpr bool value = B; };
template <typename T> struct AnchorMargin { static constexpr float value = 0.f; };
template <> struct AnchorMargin<f16_t> { static constexpr float value = 4.f; };
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// One pass over all key tiles of all segments for the wave's QB x 16 query rows.
template <typename T, int QB, int WAVES, int SM, int ANCHOR, int DMA, int HALF>
OVG_DEV void run_tiles(const ovg_attn_params& p, unsigned char* lds, const int bh, const int q0, const int total_tiles,
                       f32x4 (&o)[QB][4], f32x4 (&lacc)[QB]) {
  constexpr int NT = 64 * WAVES;
  constexpr int RB = 128, KT_B = BC * RB, VT_B = OVG_D * RB, CPT = 512 / NT;
  constexpr int NNEG = (SM == 2 && ANCHOR != 1) ? 1 : QB;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, lr = lane & 15;
  const int nq = (int)p.nq;

  u32x4 qf[QB][2];
  {
    const unsigned char* qbase = static_cast<const unsigned char*>(p.q) + (int64_t)bh * p.nq_pad * RB;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      int q = q0 + qb * 16 + lr; q = q < nq ? q : nq - 1;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        qf[qb][kk] = *reinterpret_cast<const u32x4*>(qbase + (int64_t)q * RB + (4 * kk + g) * 16);
    }
  }
  f32x4 negm[NNEG];
#pragma unroll
  for (int i = 0; i < NNEG; ++i) negm[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    lacc[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[qb][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const u32x4 ones = OnesFrag<T>::get();
  auto qk_mma = [&](const u32x4 k0, const u32x4 k1, int qb) {
    f32x4 r;
    if constexpr (SM == 2 && ANCHOR == 0) r = mma_c<T>(k0, qf[qb][0], f32x4{0.f, 0.f, 0.f, 0.f});
    else r = mma_c<T>(k0, qf[qb][0], negm[NNEG == 1 ? 0 : qb]);
    return mma_c<T>(k1, qf[qb][1], r);
  };

  // ---- staging -----------------------------------------------------------------------------------
  int fseg = 0, ftile = 0;
  int f_ntiles = (int)((p.seg[0].nk + BC - 1) / BC);
  const unsigned char* kptr = static_cast<const unsigned char*>(p.seg[0].k) + (int64_t)bh * p.seg[0].nk_pad * RB;
  const unsigned char* vptr = static_cast<const unsigned char*>(p.seg[0].vt) + (int64_t)bh * OVG_D * p.seg[0].nk_pad * 2;
  int64_t vstride = p.seg[0].nk_pad * 2;          // bytes between V^T rows (d)
  auto next_tile_ptrs = [&]() {
    kptr += KT_B;
    vptr += BC * 2;
    if (++ftile == f_ntiles) {
      ftile = 0; ++fseg;
      if (fseg < p.nseg) {
        const ovg_kv_segment sg = p.seg[fseg];
        f_ntiles = (int)((sg.nk + BC - 1) / BC);
        kptr = static_cast<const unsigned char*>(sg.k) + (int64_t)bh * sg.nk_pad * RB;
        vptr = static_cast<const unsigned char*>(sg.vt) + (int64_t)bh * OVG_D * sg.nk_pad * 2;
        vstride = sg.nk_pad * 2;
      }
    }
  };
  // DMA = 0: registers in between (as attn3)
  u32x4 rk[DMA ? 1 : CPT], rv[DMA ? 1 : CPT];
  int k_goff[CPT], v_row[CPT], v_coff[CPT], k_loff[CPT], v_loff0[CPT], v_loff1[CPT];
  // DMA = 1: a wave-instruction deposits lane l's 16 bytes at (wave-uniform base) + 16 l = 8
  // consecutive 128-byte tile rows, so the XOR swizzle goes on the SOURCE chunk; wave w stages rows
  // [R w, R w + R) of both tiles, R = 64 / WAVES, 8 rows per instruction
  constexpr int RPW = 64 / WAVES, NI = RPW / 8;
  int d_row[NI], d_coff[NI];
  if constexpr (DMA == 0) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int c = tid + NT * i;
      const int row = c >> 3, ch = c & 7;
      k_goff[i] = c * 16;
      k_loff[i] = swz_off<128>(row, ch);
      v_row[i] = row; v_coff[i] = ch * 16;
      const int u = ch >> 2, c4 = ch & 3;            // key permutation inside each 32-key block
      v_loff0[i] = swz_off<128>(row, 4 * u + 2 * (c4 & 1) + 0) + 8 * (c4 >> 1);
      v_loff1[i] = swz_off<128>(row, 4 * u + 2 * (c4 & 1) + 1) + 8 * (c4 >> 1);
    }
  } else {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      d_row[i] = wave * RPW + 8 * i + (lane >> 3);
      d_coff[i] = ((lane & 7) ^ ((d_row[i] >> 1) & 7)) * 16;
    }
  }
  auto fetch = [&](int buf) {       // issue the global reads of the next tile (DMA: straight into LDS buffer buf)
    if constexpr (DMA == 0) {
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        rk[i] = *reinterpret_cast<const u32x4*>(kptr + k_goff[i]);
        rv[i] = *reinterpret_cast<const u32x4*>(vptr + v_row[i] * vstride + v_coff[i]);
      }
    } else {
      unsigned char* kb = lds + buf * (KT_B + VT_B) + wave * RPW * RB;   // wave-uniform destinations
      unsigned char* vb = kb + KT_B;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        __builtin_amdgcn_global_load_lds((gptr_t)(kptr + d_row[i] * RB + d_coff[i]), (lptr_t)(kb + i * 8 * RB), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(vptr + d_row[i] * vstride + d_coff[i]), (lptr_t)(vb + i * 8 * RB), 16, 0, 0);
      }
    }
    next_tile_ptrs();
  };
  auto stash = [&](int buf) {       // make the fetched tile visible in LDS buffer buf (before the barrier)
    if constexpr (DMA == 0) {
      unsigned char* kl = lds + buf * (KT_B + VT_B);
      unsigned char* vl = kl + KT_B;
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        *reinterpret_cast<u32x4*>(kl + k_loff[i]) = rk[i];
        *reinterpret_cast<u32x2*>(vl + v_loff0[i]) = u32x2{rv[i][0], rv[i][1]};
        *reinterpret_cast<u32x2*>(vl + v_loff1[i]) = u32x2{rv[i][2], rv[i][3]};
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };

  int cseg = 0, ctile = 0;
  int c_ntiles = f_ntiles;
  int c_nk = (int)p.seg[0].nk;
  const int sx = lr >> 1;
  const int frag_row = lr * 128;
  const int coff0 = ((0 + g) ^ sx) << 4, coff1 = ((4 + g) ^ sx) << 4;

  fetch(0);
  stash(0);
  __syncthreads();

  auto kfrag = [&](const unsigned char* kl, int kt, u32x4& k0, u32x4& k1) {
    k0 = *reinterpret_cast<const u32x4*>(kl + kt * 2048 + frag_row + coff0);
    k1 = *reinterpret_cast<const u32x4*>(kl + kt * 2048 + frag_row + coff1);
  };
  // V^T fragment for PV step u (keys 32u + 4g + {0..3} and 32u + 16 + 4g + {0..3}), rows d = 16 dt + lr
  auto vfrag = [&](const unsigned char* vl, int u, int dt) {
    if constexpr (DMA == 0) {
      return *reinterpret_cast<const u32x4*>(vl + dt * 2048 + frag_row + (((4 * u + g) ^ sx) << 4));
    } else {
      const int va = frag_row + (((4 * u + (g >> 1)) ^ sx) << 4) + 8 * (g & 1);
      const int vb = frag_row + (((4 * u + 2 + (g >> 1)) ^ sx) << 4) + 8 * (g & 1);
      const u32x2 lo = *reinterpret_cast<const u32x2*>(vl + dt * 2048 + va);
      const u32x2 hi = *reinterpret_cast<const u32x2*>(vl + dt * 2048 + vb);
      return u32x4{lo[0], lo[1], hi[0], hi[1]};
    }
  };
  auto mask_kt = [&](f32x4 (&sk)[QB], int kv0, int kt) {   // sk[qb] = S' block of keys 16 kt .. 16 kt + 15
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool dead = (kv0 + 16 * kt + 4 * g + r) >= c_nk;
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) sk[qb][r] = dead ? -INFINITY : sk[qb][r];
    }
  };
  auto row_max = [&](const f32x4 (&sq)[4]) {
    float mx = fmaxf(sq[0][0], sq[0][1]);
    mx = fmaxf(fmaxf(mx, sq[0][2]), sq[0][3]);
#pragma unroll
    for (int kt = 1; kt < 4; ++kt) {
      mx = fmaxf(fmaxf(mx, sq[kt][0]), sq[kt][1]);
      mx = fmaxf(fmaxf(mx, sq[kt][2]), sq[kt][3]);
    }
    return xl_max4(mx);
  };
  // s[kt][qb]
  auto qk_tile = [&](const unsigned char* kl, f32x4 (&s)[4][QB], bool tail, int kv0) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      u32x4 k0, k1;
      kfrag(kl, kt, k0, k1);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) s[kt][qb] = qk_mma(k0, k1, qb);
    }
    if (tail) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) mask_kt(s[kt], kv0, kt);
    }
  };
  auto pv_step = [&](const unsigned char* vl, int u, const f32x4 (&sa)[QB], const f32x4 (&sb)[QB]) {
    u32x4 pf[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      pf[qb] = pack2<T>(sa[qb], sb[qb]);
      lacc[qb] = mma_c<T>(ones, pf[qb], lacc[qb]);
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const u32x4 vf = vfrag(vl, u, dt);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) o[qb][dt] = mma_c<T>(vf, pf[qb], o[qb][dt]);
    }
  };
  auto exp_blk = [&](f32x4 (&sk)[QB]) {
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int r = 0; r < 4; ++r) sk[qb][r] = __builtin_amdgcn_exp2f(sk[qb][r]);
  };

  if constexpr (SM == 2 && ANCHOR != 0) {
    // anchor from the first key tile (tile 0 is in LDS buffer 0 now)
    f32x4 s[4][QB];
    qk_tile(lds, s, BC > c_nk, 0);
    float mx[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const f32x4 sq[4] = {s[0][qb], s[1][qb], s[2][qb], s[3][qb]};
      mx[qb] = row_max(sq) + AnchorMargin<T>::value;
    }
    if constexpr (ANCHOR == 1) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) negm[qb] = f32x4{-mx[qb], -mx[qb], -mx[qb], -mx[qb]};
    } else {
      float m = mx[0];
#pragma unroll
      for (int qb = 1; qb < QB; ++qb) m = fmaxf(m, mx[qb]);
      negm[0] = f32x4{-m, -m, -m, -m};
    }
  }

  int buf = 0;
  for (int j = 0; j < total_tiles; ++j) {
    const bool more = (j + 1) < total_tiles;
    if (more) fetch(buf ^ 1);
    const unsigned char* kl = lds + buf * (KT_B + VT_B);
    const unsigned char* vl = kl + KT_B;
    const int kv0 = ctile * BC;

    if constexpr (SM == 2) {
      // NOTE the (rare) tail-mask branch sits between the QK^T cluster and the exponentials on
      // purpose: as ONE basic block hipcc interleaves the whole tile body, stretches the live ranges
      // and spills Q around the loop (the reload's vmcnt(0) then serialises the K/V prefetch)
      const bool tail = kv0 + BC > c_nk;
      if constexpr (HALF) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          f32x4 s2[2][QB];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            u32x4 k0, k1;
            kfrag(kl, 2 * u + h, k0, k1);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) s2[h][qb] = qk_mma(k0, k1, qb);
          }
          if (tail) { mask_kt(s2[0], kv0, 2 * u); mask_kt(s2[1], kv0, 2 * u + 1); }
          exp_blk(s2[0]);
          exp_blk(s2[1]);
          pv_step(vl, u, s2[0], s2[1]);
        }
      } else {
        f32x4 s[4][QB];
        qk_tile(kl, s, tail, kv0);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) exp_blk(s[kt]);
        pv_step(vl, 0, s[0], s[1]);
        pv_step(vl, 1, s[2], s[3]);
      }
    } else {
      f32x4 s[4][QB];
      qk_tile(kl, s, kv0 + BC > c_nk, kv0);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        const f32x4 sq[4] = {s[0][qb], s[1][qb], s[2][qb], s[3][qb]};
        const float mx = row_max(sq);
        if (j == 0) {
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) s[kt][qb] -= mx;
          negm[qb] = f32x4{-mx, -mx, -mx, -mx};
        } else if (__any(mx > RESCALE_THR)) {
          const float delta = fmaxf(mx, 0.f);
          const float alpha = __builtin_amdgcn_exp2f(-delta);
          negm[qb] -= delta;
          lacc[qb] *= alpha;
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) s[kt][qb] -= delta;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) o[qb][dt] *= alpha;
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[kt][qb][r] = __builtin_amdgcn_exp2f(s[kt][qb][r]);
      }
      pv_step(vl, 0, s[0], s[1]);
      pv_step(vl, 1, s[2], s[3]);
    }

    if (++ctile == c_ntiles) {
      ctile = 0; ++cseg;
      if (cseg < p.nseg) { c_nk = (int)p.seg[cseg].nk; c_ntiles = (c_nk + BC - 1) / BC; }
    }
    if (more) stash(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
}

}  // namespace spec

// FORCE_FALLBACK (tests only): run the speculative pass, then take the recompute path regardless.
template <typename T, int QB, int WAVES, int ANCHOR, int DMA, int HALF, bool FORCE_FALLBACK = false>
__global__ __launch_bounds__(64 * WAVES, 2) void attn_spec_kernel(ovg_attn_params p, int nqt, int total_tiles) {
  static_assert(sizeof(T) == 2, "16-bit types only");
  constexpr int RB = 128, KT_B = BC * RB, VT_B = OVG_D * RB, BQ = 16 * QB * WAVES;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * (KT_B + VT_B)];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, lr = lane & 15;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = lid / nqt, qt = lid % nqt;
  const int nq = (int)p.nq;
  const int q0 = qt * BQ + wave * 16 * QB;

  f32x4 o[QB][4], lacc[QB];
  spec::run_tiles<T, QB, WAVES, 2, ANCHOR, DMA, HALF>(p, lds, bh, q0, total_tiles, o, lacc);

  bool bad = FORCE_FALLBACK;
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    bad = bad || spec::bad_sum(lacc[qb][0]);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) bad = bad || spec::nonfinite(o[qb][dt][r]);
  }
  if (__syncthreads_or(bad ? 1 : 0)) spec::run_tiles<T, QB, WAVES, 0, 1, DMA, 0>(p, lds, bh, q0, total_tiles, o, lacc);

  const int bq = bh / OVG_H, hh = bh % OVG_H;
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float inv = 1.0f / lacc[qb][0];        // every row of the ones-MFMA holds the full row sum
    const int q = q0 + qb * 16 + lr;
    if (q < nq) {
      T* dst = static_cast<T*>(p.out) + ((int64_t)bq * nq + q) * p.ldo + hh * OVG_D + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        store4<T>(dst + 16 * dt, o[qb][dt][0] * inv, o[qb][dt][1] * inv, o[qb][dt][2] * inv, o[qb][dt][3] * inv);
    }
  }
}
