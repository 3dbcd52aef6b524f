// attn4_kernel: software-pipelined 16-bit flash attention (included by ovg_attn.hip).
//
// PMC on attn3 (rocprofv3, S=32): the matrix pipe is busy 51 % of the time, only 24 % of that busy
// time overlaps VALU execution, and waves spend 40 % of their cycles stalled on issue -- each wave
// runs QK^T (MFMA) -> softmax (VALU) -> PV (MFMA) strictly in sequence and the two waves of a SIMD
// overlap only by chance.  attn4 makes every wave feed both pipes at once:
//   * the 64-key LDS tile is consumed as two 32-key half-steps;
//   * S' of half-step t+1 is ISSUED BEFORE the softmax of half-step t (independent instruction
//     streams in one basic block: hipcc interleaves the MFMAs with the max/exp/convert VALU work);
//   * K tiles therefore run one tile ahead of V^T tiles: a 3-deep K ring + 2-deep V^T ring in LDS
//     (40 KB), one barrier per tile, register-staged prefetch two K tiles / one V^T tile ahead;
//   * the lazy rescale is decided once per half-step for all q blocks (one rare branch), and it
//     also shifts the PENDING S' of the next half-step, which was accumulated against the old
//     reference (every quantity at the old reference moves together -- exact).
// Row sums are plain VALU adds here (they hide under the MFMAs; saves the 16 accumulator
// registers of attn3's ones-MFMA, which this kernel needs for the second S' buffer).
#pragma once

template <typename T, int QB>
__global__ __launch_bounds__(256, 2) void attn4_kernel(ovg_attn_params p, int nqt, int total_tiles) {
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
  float lsum[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    lsum[qb] = 0.f;
    negm[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[qb][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- staging ------------------------------------------------------------------------------
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
  // independent cursors: K runs one tile ahead of V^T
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

  // mask cursor: the tile whose S' is being formed
  int mseg = 0, mtile = 0, m_ntiles = k_ntiles, m_nk = (int)p.seg[0].nk;
  const int sx = lr >> 1;
  const int frag_row = lr * 128;
  const int coff0 = ((0 + g) ^ sx) << 4, coff1 = ((4 + g) ^ sx) << 4;

  // S'[qb][i] for the two 16-key tiles of half h of the K tile at `kl`
  auto qk_half = [&](const unsigned char* kl, int h, f32x4 (&s)[QB][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kt = 2 * h + i;
      const u32x4 k0 = *reinterpret_cast<const u32x4*>(kl + kt * 2048 + frag_row + coff0);
      const u32x4 k1 = *reinterpret_cast<const u32x4*>(kl + kt * 2048 + frag_row + coff1);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        s[qb][i] = mma_c<T>(k0, qf[qb][0], negm[qb]);
        s[qb][i] = mma_c<T>(k1, qf[qb][1], s[qb][i]);
      }
    }
    const int kv0 = mtile * BC;
    if (kv0 + BC > m_nk) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool dead = (kv0 + 16 * (2 * h + i) + 4 * g + r) >= m_nk;
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) s[qb][i][r] = dead ? -INFINITY : s[qb][i][r];
        }
    }
  };
  // softmax of the current half-step; `nxt` is the pending S' of the next one (same reference)
  auto softmax_half = [&](f32x4 (&cur)[QB][2], f32x4 (&nxt)[QB][2], bool first) {
    float mx[QB];
    bool need = first;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float m = fmaxf(cur[qb][0][0], cur[qb][0][1]);
      m = fmaxf(fmaxf(m, cur[qb][0][2]), cur[qb][0][3]);
      m = fmaxf(fmaxf(m, cur[qb][1][0]), cur[qb][1][1]);
      m = fmaxf(fmaxf(m, cur[qb][1][2]), cur[qb][1][3]);
      mx[qb] = xl_max4(m);
      need = need || (mx[qb] > RESCALE_THR);
    }
    if (__any(need)) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        // first half-step: anchor at the true row max (o = l = 0: nothing else to rescale)
        const float delta = first ? mx[qb] : fmaxf(mx[qb], 0.f);
        const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);
        negm[qb] -= delta;
        lsum[qb] *= alpha;
        cur[qb][0] -= delta; cur[qb][1] -= delta;
        nxt[qb][0] -= delta; nxt[qb][1] -= delta;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[qb][dt] *= alpha;
      }
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float rs = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = __builtin_amdgcn_exp2f(cur[qb][i][r]);
          cur[qb][i][r] = pv;
          rs += pv;
        }
      lsum[qb] += rs;
    }
  };
  auto pv_half = [&](const unsigned char* vl, int h, f32x4 (&cur)[QB][2]) {
    u32x4 pf[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      T v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = TT<T>::from_f32(cur[qb][0][r]); v[4 + r] = TT<T>::from_f32(cur[qb][1][r]); }
      __builtin_memcpy(&pf[qb], v, 16);
    }
    const int voff = ((4 * h + g) ^ sx) << 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const u32x4 vf = *reinterpret_cast<const u32x4*>(vl + dt * 2048 + frag_row + voff);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) o[qb][dt] = mma_c<T>(vf, pf[qb], o[qb][dt]);
    }
  };

  // ---- prologue: K(0), K(1), V(0) resident; S' of half-step 0 in flight --------------------------
  fetch_k(); stash_k(0);
  if (total_tiles > 1) { fetch_k(); stash_k(1); }
  fetch_v(); stash_v(0);
  __syncthreads();

  f32x4 sa[QB][2], sb[QB][2];      // ping-pong S' buffers (static indexing only)
  qk_half(kring, 0, sa);

  int kb = 0, vb = 0;              // ring slots of K(j), V(j)
  bool first = true;
  for (int j = 0; j < total_tiles; ++j) {
    const bool has_k2 = (j + 2) < total_tiles, has_n1 = (j + 1) < total_tiles;
    if (has_k2) fetch_k();
    if (has_n1) fetch_v();
    const unsigned char* kl = kring + kb * KT_B;
    const unsigned char* vl = vring + vb * VT_B;

    // half-step (j,0): current = sa, next = S'(j,1) -> sb
    qk_half(kl, 1, sb);
    softmax_half(sa, sb, first);
    first = false;
    pv_half(vl, 0, sa);

    // half-step (j,1): current = sb, next = S'(j+1,0) -> sa
    if (has_n1) {
      if (++mtile == m_ntiles) { mtile = 0; ++mseg; m_nk = (int)p.seg[mseg].nk; m_ntiles = (m_nk + BC - 1) / BC; }
      const int kb1 = kb == 2 ? 0 : kb + 1;
      qk_half(kring + kb1 * KT_B, 0, sa);
    } else {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) { sa[qb][0] = f32x4{0.f, 0.f, 0.f, 0.f}; sa[qb][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
    softmax_half(sb, sa, false);
    pv_half(vl, 1, sb);

    if (has_k2) stash_k(kb == 0 ? 2 : kb - 1);     // slot (kb+2)%3 held K(j-1): dead since iteration j-1
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
