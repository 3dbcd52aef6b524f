// attn5_kernel: software-pipelined 16-bit flash attention with an EXPLICIT MFMA/VALU interleave
// (included by ovg_attn.hip).  Same data flow as attn4 (32-key half-steps, K ring 3-deep / V^T ring
// 2-deep, S' of half-step t+1 formed while half-step t is in its softmax), but the instruction
// stream is laid out by hand because hipcc clusters the MFMAs of a basic block (ISA of attn4:
// "MMMMMMMMMMMM | vvvv...": an in-order wave cannot start its VALU work until the last MFMA of
// the cluster has issued).  Per half-step there are two straight-line blocks:
//   B1: for n = 0..4*QB-1:  MFMA n of S'_{t+1}  ;  exp2 of two probabilities of S'_t, their row-sum
//       add, one packed bf16 convert            (sched_barrier pins each chunk)
//   B2: for n = 0..4*QB-1:  MFMA n of O += V^T P_t ;  a slice of the row-max chain of S'_{t+1}
// The rescale decision of half-step t is taken BEFORE S'_{t+1} is issued, so S'_{t+1} is always
// accumulated against the current reference (no fix-up of pending scores).
#pragma once
#include <type_traits>

template <typename T> OVG_DEV unsigned pack2(float a, float b) {
  T v[2] = {TT<T>::from_f32(a), TT<T>::from_f32(b)};
  unsigned r;
  __builtin_memcpy(&r, v, 4);
  return r;
}

template <typename T, int QB>
__global__ __launch_bounds__(256, 2) void attn5_kernel(ovg_attn_params p, int nqt, int total_tiles) {
  static_assert(sizeof(T) == 2, "16-bit types only");
  constexpr int RB = 128, KT_B = BC * RB, VT_B = OVG_D * RB, CPT = 2, BQ = 64 * QB;
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * KT_B + 2 * VT_B];
  unsigned char* const kring = lds;
  unsigned char* const vring = lds + 3 * KT_B;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, lr = lane & 15;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = lid / nqt, qt = lid % nqt;
  const int nq = (int)p.nq;
  const int q0 = qt * BQ + wave * 16 * QB;

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
  f32x4 o[QB][4], negm[QB];
  float lsum[QB], mx[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    lsum[qb] = 0.f;
    negm[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[qb][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- staging (identical to attn4) --------------------------------------------------------------
  u32x4 rk[CPT], rv[CPT];
  int k_goff[CPT], v_row[CPT], v_coff[CPT], k_loff[CPT], v_loff0[CPT], v_loff1[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = tid + 256 * i;
    const int row = c >> 3, ch = c & 7;
    k_goff[i] = c * 16;
    k_loff[i] = swz_off<128>(row, ch);
    v_row[i] = row; v_coff[i] = ch * 16;
    const int u = ch >> 2, c4 = ch & 3;
    v_loff0[i] = swz_off<128>(row, 4 * u + 2 * (c4 & 1) + 0) + 8 * (c4 >> 1);
    v_loff1[i] = swz_off<128>(row, 4 * u + 2 * (c4 & 1) + 1) + 8 * (c4 >> 1);
  }
  int kseg = 0, ktile = 0, k_ntiles = (int)((p.seg[0].nk + BC - 1) / BC);
  const unsigned char* kptr = static_cast<const unsigned char*>(p.seg[0].k) + (int64_t)bh * p.seg[0].nk_pad * RB;
  int vseg = 0, vtile = 0, v_ntiles = k_ntiles;
  const unsigned char* vptr = static_cast<const unsigned char*>(p.seg[0].vt) + (int64_t)bh * OVG_D * p.seg[0].nk_pad * 2;
  int64_t vstride = p.seg[0].nk_pad * 2;
  auto fetch_k = [&]() {
#pragma unroll
    for (int i = 0; i < CPT; ++i) rk[i] = *reinterpret_cast<const u32x4*>(kptr + k_goff[i]);
    kptr += KT_B;
    if (++ktile == k_ntiles) {
      ktile = 0; ++kseg;
      if (kseg < p.nseg) {
        const ovg_kv_segment sg = p.seg[kseg];
        k_ntiles = (int)((sg.nk + BC - 1) / BC);
        kptr = static_cast<const unsigned char*>(sg.k) + (int64_t)bh * sg.nk_pad * RB;
      }
    }
  };
  auto fetch_v = [&]() {
#pragma unroll
    for (int i = 0; i < CPT; ++i) rv[i] = *reinterpret_cast<const u32x4*>(vptr + v_row[i] * vstride + v_coff[i]);
    vptr += BC * 2;
    if (++vtile == v_ntiles) {
      vtile = 0; ++vseg;
      if (vseg < p.nseg) {
        const ovg_kv_segment sg = p.seg[vseg];
        v_ntiles = (int)((sg.nk + BC - 1) / BC);
        vptr = static_cast<const unsigned char*>(sg.vt) + (int64_t)bh * OVG_D * sg.nk_pad * 2;
        vstride = sg.nk_pad * 2;
      }
    }
  };
  auto stash_k = [&](int slot) {
    unsigned char* kl = kring + slot * KT_B;
#pragma unroll
    for (int i = 0; i < CPT; ++i) *reinterpret_cast<u32x4*>(kl + k_loff[i]) = rk[i];
  };
  auto stash_v = [&](int slot) {
    unsigned char* vl = vring + slot * VT_B;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      *reinterpret_cast<u32x2*>(vl + v_loff0[i]) = u32x2{rv[i][0], rv[i][1]};
      *reinterpret_cast<u32x2*>(vl + v_loff1[i]) = u32x2{rv[i][2], rv[i][3]};
    }
  };

  int mseg = 0, mtile = 0, m_ntiles = k_ntiles, m_nk = (int)p.seg[0].nk;   // tile whose S' is being formed
  const int sx = lr >> 1;
  const int frag_row = lr * 128;
  const int coff0 = ((0 + g) ^ sx) << 4, coff1 = ((4 + g) ^ sx) << 4;

  // ragged tail of a segment: keys >= nk of the tile whose S' is being formed get -inf (unconditional
  // selects: the caller only instantiates this for tail tiles, keeping the hot half-step branch-free)
  auto mask_half = [&](int h, f32x4 (&s)[QB][2]) {
    const int kv0 = mtile * BC;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool dead = (kv0 + 16 * (2 * h + i) + 4 * g + r) >= m_nk;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) s[qb][i][r] = dead ? -INFINITY : s[qb][i][r];
      }
  };
  auto is_tail = [&]() { return mtile * BC + BC > m_nk; };
  auto rowmax_lane = [&](const f32x4 (&s)[2]) {
    float m = fmaxf(s[0][0], s[0][1]);
    m = fmaxf(fmaxf(m, s[0][2]), s[0][3]);
    m = fmaxf(fmaxf(m, s[1][0]), s[1][1]);
    return fmaxf(fmaxf(m, s[1][2]), s[1][3]);
  };
  // rare path: move the reference of every q block (everything at the old reference moves together)
  auto rescale = [&](f32x4 (&cur)[QB][2], bool first) {
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const float delta = first ? mx[qb] : fmaxf(mx[qb], 0.f);
      const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);
      negm[qb] -= delta;
      lsum[qb] *= alpha;
      cur[qb][0] -= delta; cur[qb][1] -= delta;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[qb][dt] *= alpha;
    }
  };

  // One half-step.  cur = S'_t with mx[] = its row maxima; forms nxt = S'_{t+1} from half `hn` of the
  // K tile at `kn`, accumulates P_t V into o from half `hv` of the V^T tile at `vl`.
  auto half_step = [&](auto mask_tag, f32x4 (&cur)[QB][2], f32x4 (&nxt)[QB][2], const unsigned char* kn, int hn,
                       const unsigned char* vl, int hv, bool first) {
    constexpr bool MASK = decltype(mask_tag)::value;
    bool need = first;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) need = need || (mx[qb] > RESCALE_THR);
    if (__any(need)) rescale(cur, first);

    // ---- B1: S'_{t+1} MFMAs  ||  exp2 / row-sum / convert of S'_t --------------------------------
    u32x4 kf[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      kf[i][0] = *reinterpret_cast<const u32x4*>(kn + (2 * hn + i) * 2048 + frag_row + coff0);
      kf[i][1] = *reinterpret_cast<const u32x4*>(kn + (2 * hn + i) * 2048 + frag_row + coff1);
    }
    u32x4 pf[QB];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)          // kk outermost: consecutive MFMAs hit different accumulators
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          nxt[qb][i] = mma_c<T>(kf[i][kk], qf[qb][kk], kk == 0 ? negm[qb] : nxt[qb][i]);
          // VALU slice for this MFMA slot: two probabilities of cur[qb][i]
          const float p0 = __builtin_amdgcn_exp2f(cur[qb][i][2 * kk]);
          const float p1 = __builtin_amdgcn_exp2f(cur[qb][i][2 * kk + 1]);
          lsum[qb] += p0 + p1;
          pf[qb][2 * i + kk] = pack2<T>(p0, p1);
          __builtin_amdgcn_sched_barrier(0);
        }
    if constexpr (MASK) mask_half(hn, nxt);

    // ---- B2: O += V^T P_t MFMAs  ||  row maxima of S'_{t+1} ----------------------------------------
    u32x4 vf[4];
    const int voff = ((4 * hv + g) ^ sx) << 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vf[dt] = *reinterpret_cast<const u32x4*>(vl + dt * 2048 + frag_row + voff);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float m = 0.f;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        o[qb][dt] = mma_c<T>(vf[dt], pf[qb], o[qb][dt]);
        if (dt == 0) m = rowmax_lane(nxt[qb]);
        else if (dt == 1) m = swap32_partner_max(m);
        else if (dt == 2) mx[qb] = swap16_partner_max(m);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // ---- prologue ------------------------------------------------------------------------------------
  fetch_k(); stash_k(0);
  if (total_tiles > 1) { fetch_k(); stash_k(1); }
  fetch_v(); stash_v(0);
  __syncthreads();

  f32x4 sa[QB][2], sb[QB][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const u32x4 k0 = *reinterpret_cast<const u32x4*>(kring + i * 2048 + frag_row + coff0);
    const u32x4 k1 = *reinterpret_cast<const u32x4*>(kring + i * 2048 + frag_row + coff1);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      sa[qb][i] = mma_c<T>(k0, qf[qb][0], negm[qb]);
      sa[qb][i] = mma_c<T>(k1, qf[qb][1], sa[qb][i]);
    }
  }
  if (is_tail()) mask_half(0, sa);
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) mx[qb] = xl_max4(rowmax_lane(sa[qb]));
  typedef std::integral_constant<bool, true> yes_t;
  typedef std::integral_constant<bool, false> no_t;

  int kb = 0, vb = 0;
  bool first = true;
  for (int j = 0; j < total_tiles; ++j) {
    const bool has_k2 = (j + 2) < total_tiles, has_n1 = (j + 1) < total_tiles;
    if (has_k2) fetch_k();
    if (has_n1) fetch_v();
    const unsigned char* kl = kring + kb * KT_B;
    const unsigned char* vl = vring + vb * VT_B;

    if (is_tail()) half_step(yes_t{}, sa, sb, kl, 1, vl, 0, first);      // (j,0): forms S'(j,1)
    else half_step(no_t{}, sa, sb, kl, 1, vl, 0, first);
    first = false;
    // (j,1): forms S'(j+1,0); on the last tile the K slot is re-read and the result is unused
    const unsigned char* kn = kl;
    if (has_n1) {
      if (++mtile == m_ntiles) { mtile = 0; ++mseg; m_nk = (int)p.seg[mseg].nk; m_ntiles = (m_nk + BC - 1) / BC; }
      kn = kring + (kb == 2 ? 0 : kb + 1) * KT_B;
    }
    if (is_tail()) half_step(yes_t{}, sb, sa, kn, 0, vl, 1, false);
    else half_step(no_t{}, sb, sa, kn, 0, vl, 1, false);

    if (has_k2) stash_k(kb == 0 ? 2 : kb - 1);
    if (has_n1) stash_v(vb ^ 1);
    __syncthreads();
    kb = kb == 2 ? 0 : kb + 1;
    vb ^= 1;
  }

  const int bq = bh / OVG_H, hh = bh % OVG_H;
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float inv = 1.0f / xl_sum4(lsum[qb]);
    const int q = q0 + qb * 16 + lr;
    if (q < nq) {
      T* dst = static_cast<T*>(p.out) + ((int64_t)bq * nq + q) * p.ldo + hh * OVG_D + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        store4<T>(dst + 16 * dt, o[qb][dt][0] * inv, o[qb][dt][1] * inv, o[qb][dt][2] * inv, o[qb][dt][3] * inv);
    }
  }
}
