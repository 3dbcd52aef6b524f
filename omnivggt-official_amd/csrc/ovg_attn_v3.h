// attn3_kernel: third iteration of the 16-bit flash-attention kernel (included by ovg_attn.hip).
// On top of attn2_kernel (lazy rescale, VGPR MFMAs, swizzled V^T, permlane reductions):
//  * the row sums come out of the MATRIX pipe: one extra MFMA per (q block, 32-key step) with an
//    all-ones A operand accumulates sum_k P[k,q] (already reduced across the 4 lanes of the row),
//    replacing 16 v_add per q block per tile on the VALU, which PMC showed is the busier pipe
//    (rocprofv3: VALU-active 52 % vs MFMA-busy 33 % on attn2);
//  * the row max is a pure v_max3 chain;
//  * K / V^T tile addresses advance incrementally (no 64-bit multiplies per tile);
//  * WAVES (2 or 4 waves per workgroup) is a template parameter: 2-wave workgroups halve the
//    work quantum (128-row q tiles at QB=4) so the 688-workgroup S=8 launch no longer runs 1.34
//    "rounds" on 512 slots.
#pragma once

template <typename T> struct OnesFrag;
template <> struct OnesFrag<bf16_t> { static OVG_DEV u32x4 get() { return u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u}; } };
template <> struct OnesFrag<f16_t> { static OVG_DEV u32x4 get() { return u32x4{0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u}; } };

template <typename T, int QB, int WAVES, bool PRIO = false, int SM = 0>
__global__ __launch_bounds__(64 * WAVES, 2) void attn3_kernel(ovg_attn_params p, int nqt, int total_tiles) {
  static_assert(sizeof(T) == 2, "16-bit types only");
  constexpr int NT = 64 * WAVES;
  constexpr int RB = 128, KT_B = BC * RB, VT_B = OVG_D * RB, CPT = 512 / NT, BQ = 16 * QB * WAVES;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * (KT_B + VT_B)];

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
  f32x4 o[QB][4], lacc[QB], negm[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    lacc[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    negm[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[qb][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const u32x4 ones = OnesFrag<T>::get();

  // ---- staging: per-thread chunk coordinates; tile pointers advance incrementally -------------
  u32x4 rk[CPT], rv[CPT];
  int k_goff[CPT], v_row[CPT], v_coff[CPT], k_loff[CPT], v_loff0[CPT], v_loff1[CPT];
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
  int fseg = 0, ftile = 0;
  int f_ntiles = (int)((p.seg[0].nk + BC - 1) / BC);
  const unsigned char* kptr = static_cast<const unsigned char*>(p.seg[0].k) + (int64_t)bh * p.seg[0].nk_pad * RB;
  const unsigned char* vptr = static_cast<const unsigned char*>(p.seg[0].vt) + (int64_t)bh * OVG_D * p.seg[0].nk_pad * 2;
  int64_t vstride = p.seg[0].nk_pad * 2;          // bytes between V^T rows (d)
  auto fetch = [&]() {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      rk[i] = *reinterpret_cast<const u32x4*>(kptr + k_goff[i]);
      rv[i] = *reinterpret_cast<const u32x4*>(vptr + v_row[i] * vstride + v_coff[i]);
    }
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
  auto stash = [&](int buf) {
    unsigned char* kl = lds + buf * (KT_B + VT_B);
    unsigned char* vl = kl + KT_B;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      *reinterpret_cast<u32x4*>(kl + k_loff[i]) = rk[i];
      *reinterpret_cast<u32x2*>(vl + v_loff0[i]) = u32x2{rv[i][0], rv[i][1]};
      *reinterpret_cast<u32x2*>(vl + v_loff1[i]) = u32x2{rv[i][2], rv[i][3]};
    }
  };

  int cseg = 0, ctile = 0;
  int c_ntiles = f_ntiles;
  int c_nk = (int)p.seg[0].nk;
  const int sx = lr >> 1;
  const int frag_row = lr * 128;
  const int coff0 = ((0 + g) ^ sx) << 4, coff1 = ((4 + g) ^ sx) << 4;

  fetch();
  stash(0);
  __syncthreads();

  int buf = 0;
  for (int j = 0; j < total_tiles; ++j) {
    const bool more = (j + 1) < total_tiles;
    if (more) fetch();
    const unsigned char* kl = lds + buf * (KT_B + VT_B);
    const unsigned char* vl = kl + KT_B;

    // ---- S' = K Q^T - m_ref ----------------------------------------------
    f32x4 s[QB][4];
    if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const u32x4 k0 = *reinterpret_cast<const u32x4*>(kl + kt * 2048 + frag_row + coff0);
      const u32x4 k1 = *reinterpret_cast<const u32x4*>(kl + kt * 2048 + frag_row + coff1);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        s[qb][kt] = mma_c<T>(k0, qf[qb][0], negm[qb]);
        s[qb][kt] = mma_c<T>(k1, qf[qb][1], s[qb][kt]);
      }
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    const int kv0 = ctile * BC;
    if (kv0 + BC > c_nk) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool dead = (kv0 + 16 * kt + 4 * g + r) >= c_nk;
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) s[qb][kt][r] = dead ? -INFINITY : s[qb][kt][r];
        }
    }
    // ---- lazy-rescale online softmax (row sums are taken by the MFMA below) --------------------
    if constexpr (SM == 2) {
      // bounded-logit path: the launcher has proved |s| <= LOGIT_BOUND for every (q, k) of this
      // launch (Cauchy-Schwarz on the row norms), so exp2(s) can neither overflow nor flush to zero
      // in f32 / bf16 and softmax needs no running max at all: P = exp2(s), O = (sum P v) / (sum P).
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[qb][kt][r] = __builtin_amdgcn_exp2f(s[qb][kt][r]);
    } else if constexpr (SM == 1) {
      // all row-max chains first (independent chains in ONE basic block, so hipcc interleaves them
      // and their permlane hazards), then a single rare branch for every q block, then the exps
      float mxv[QB];
      bool need = (j == 0);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        float mx = fmaxf(s[qb][0][0], s[qb][0][1]);
        mx = fmaxf(fmaxf(mx, s[qb][0][2]), s[qb][0][3]);
#pragma unroll
        for (int kt = 1; kt < 4; ++kt) {
          mx = fmaxf(fmaxf(mx, s[qb][kt][0]), s[qb][kt][1]);
          mx = fmaxf(fmaxf(mx, s[qb][kt][2]), s[qb][kt][3]);
        }
        mxv[qb] = xl_max4(mx);
        need = need || (mxv[qb] > RESCALE_THR);
      }
      if (__any(need)) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          const float delta = (j == 0) ? mxv[qb] : fmaxf(mxv[qb], 0.f);   // first tile: anchor at the row max
          const float alpha = (j == 0) ? 1.0f : __builtin_amdgcn_exp2f(-delta);
          negm[qb] -= delta;
          lacc[qb] *= alpha;
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) s[qb][kt] -= delta;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) o[qb][dt] *= alpha;
        }
      }
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[qb][kt][r] = __builtin_amdgcn_exp2f(s[qb][kt][r]);
    } else
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float mx = fmaxf(s[qb][0][0], s[qb][0][1]);
      mx = fmaxf(fmaxf(mx, s[qb][0][2]), s[qb][0][3]);
#pragma unroll
      for (int kt = 1; kt < 4; ++kt) {
        mx = fmaxf(fmaxf(mx, s[qb][kt][0]), s[qb][kt][1]);
        mx = fmaxf(fmaxf(mx, s[qb][kt][2]), s[qb][kt][3]);
      }
      mx = xl_max4(mx);
      if (j == 0) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) s[qb][kt] -= mx;
        negm[qb] = f32x4{-mx, -mx, -mx, -mx};
      } else if (__any(mx > RESCALE_THR)) {
        const float delta = fmaxf(mx, 0.f);
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        negm[qb] -= delta;
        lacc[qb] *= alpha;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) s[qb][kt] -= delta;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[qb][dt] *= alpha;
      }
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[qb][kt][r] = __builtin_amdgcn_exp2f(s[qb][kt][r]);
    }
    // ---- O^T += V^T P^T ;  l += 1^T P^T ---------------------------------------
    if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      u32x4 pf[QB];
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        pf[qb] = VFrag<T>::pfrag(s[qb], u);
        lacc[qb] = mma_c<T>(ones, pf[qb], lacc[qb]);
      }
      const int voff = ((4 * u + g) ^ sx) << 4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const u32x4 vf = *reinterpret_cast<const u32x4*>(vl + dt * 2048 + frag_row + voff);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) o[qb][dt] = mma_c<T>(vf, pf[qb], o[qb][dt]);
      }
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    if (++ctile == c_ntiles) {
      ctile = 0; ++cseg;
      if (cseg < p.nseg) { c_nk = (int)p.seg[cseg].nk; c_ntiles = (c_nk + BC - 1) / BC; }
    }
    if (more) stash(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

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
