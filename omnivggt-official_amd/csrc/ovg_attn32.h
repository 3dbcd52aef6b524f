// 32 x 32 x 16 MFMA formulation of the speculative flash-attention pass (included by ovg_attn.hip after ovg_attn16.h).
//
// Why: v_mfma_f32_16x16x32_bf16 tops out at 2075 TFLOP/s in the guide's micro-benchmark, v_mfma_f32_32x32x16_bf16 at
// 2382 (MI355X_MICROARCH.md / cdna_hip_programming.md section 3), and the 16 x 16 kernel spends 8 of its 72 MFMAs per key
// tile on the row sums (ones operand). PMC on the shipped 16 x 16 kernel at 64 views (profiles/r02_pmc_attention.txt):
// matrix pipe busy 60 % of the clocked cycles, i.e. the kernel is within 40 % of its own instruction's ceiling.
// Here a key tile costs 32 MFMAs of 32 cycles (1024 pipe cycles instead of 1152), half as many MFMA issues, and the
// row sums are 64 f32 adds per lane on the VALU.
//
// Same "swapped" scheme as ovg_attn16.h, on 32 x 32 blocks (D = MFMA(A, B): A rows -> D rows, B rows -> D columns;
// lane l supplies A row l & 31 / B row l & 31 and the 8 k-values 8 h .. 8 h + 7 of the 16-deep step, h = l >> 5;
// D[row = (r & 3) + 8 (r >> 2) + 4 h][col = l & 31], r = 0 .. 15):
//   S^T[key, q] = K[key, :] . Q[q, :]         A = K rows (LDS), B = Q rows (registers, whole kernel), C = -m_ref
//   O^T[d, q]  += V^T[d, key] * P^T[key, q]   A = V^T rows (LDS), B = P (registers, straight from S^T)
// A wave owns 64 query rows (2 blocks of 32, one per lane & 31) and walks 64-key tiles (2 key blocks of 32):
//  * a lane holds S^T for q = l & 31 and the 16 keys 8 j + 4 h + i (j, i < 4) of each 32-key block; the other 16 keys of the
//    block sit in lane l ^ 32: row maxima / sums need ONE v_permlane32_swap, and only outside the tile loop;
//  * P feeds the PV MFMA without cross-lane movement: the 16-key step t of a tile takes the accumulator registers
//    r = 8 (t & 1) .. 8 (t & 1) + 7 of key block t >> 1, i.e. keys 16 t + 8 (i8 >> 2) + 4 h + (i8 & 3) in k-slot (h, i8);
//    the V^T LDS tile stores its keys in exactly that order (16-byte chunk 2 t + h of a row = the 8 keys of slot h), so a
//    V^T fragment is one conflict-free ds_read_b128 -- the k-permutation trick of the 16 x 16 kernel;
//  * K / V^T tiles use the same XOR-swizzled 128-byte rows (chunk ^= (row >> 1) & 7): conflict-free for the 32-row
//    fragment reads as well (a ds_read_b128 lane group holds 16 rows with 8 distinct row pairs).
// Speculative anchored softmax only (SM = 2 of ovg_attn16.h): the anchor is the row maximum over the first key tile; after
// the pass every row checks l within 2^+-100 and O finite; if any row of the workgroup fails, the workgroup recomputes
// with the 16 x 16 lazy-rescale pass (attn16::run_tiles<SM = 0>) -- the result is always the exact softmax.
#pragma once

namespace attn32 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <typename T> OVG_DEV f32x16 mma32(const u32x4& a, const u32x4& b, const f32x16& c);
template <> OVG_DEV f32x16 mma32<bf16_t>(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <> OVG_DEV f32x16 mma32<f16_t>(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

OVG_DEV float swap32_partner(float v) {            // the value lane l ^ 32 holds
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned a = r[0], b = r[1];
  return (threadIdx.x & 32) ? __builtin_bit_cast(float, a) : __builtin_bit_cast(float, b);
}

template <typename T>
OVG_DEV u32x4 pack8(const f32x16& s, const int r0) {   // registers r0 .. r0 + 7 -> 8 T
  T v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = TT<T>::from_f32(s[r0 + i]);
  u32x4 r;
  __builtin_memcpy(&r, v, 16);
  return r;
}

// One speculative pass over key tiles [t_begin, t_begin + n_tiles) for the wave's 64 query rows q0 .. q0 + 63.
// Leaves the un-normalised O^T in o[db][qb] (d = 32 db + (r & 3) + 8 (r >> 2) + 4 h, q = q0 + 32 qb + (l & 31)), the FULL row
// sums in lsum[qb] (both lanes of a row hold the total) and the negated anchors in negm[qb].
template <typename T, int WAVES>
OVG_DEV void run_tiles32(const ovg_attn_params& p, unsigned char* lds, const int bh, const int q0, const int t_begin, const int n_tiles,
                         f32x16 (&o)[2][2], float (&lsum)[2], float (&negm)[2]) {
  // (-anchor) replicated over a 32 x 32 accumulator: the C operand of the first MFMA of every S^T block, so the exponent
  // needs no per-element subtract and no per-tile accumulator initialisation. ONE anchor per lane, shared by its two query
  // rows (q0 + lq and q0 + 32 + lq): the anchor only has to be within 2^+-100 of a row's maximum, the check after the pass
  // catches rows for which it is not (16 VGPRs for the whole kernel instead of 32).
  f32x16 negc;
  constexpr int NT = 64 * WAVES;
  constexpr int RB = 128, KT_B = BC * RB, VT_B = OVG_D * RB, CPT = 512 / NT;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int h = lane >> 5, lq = lane & 31;
  const int nq = (int)p.nq;

  // Q fragments (B operand of QK^T): chunk (2 ds + h) of row q
  u32x4 qf[2][4];
  {
    const unsigned char* qbase = static_cast<const unsigned char*>(p.q) + (int64_t)bh * p.nq_pad * RB;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      int q = q0 + qb * 32 + lq; q = q < nq ? q : nq - 1;
#pragma unroll
      for (int ds = 0; ds < 4; ++ds)
        qf[qb][ds] = *reinterpret_cast<const u32x4*>(qbase + (int64_t)q * RB + (2 * ds + h) * 16);
    }
  }
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    negm[qb] = 0.f;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[db][qb][r] = 0.f;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) negc[r] = 0.f;
  float part[2][4];                                  // four independent partial row sums per q block (no 32-deep add chain)
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int i = 0; i < 4; ++i) part[qb][i] = 0.f;

  // ---- staging: identical to the 16 x 16 kernel except for the key order inside a V^T row --------------------------
  u32x4 rk[CPT], rv[CPT];
  int k_goff[CPT], v_row[CPT], v_coff[CPT], k_loff[CPT], v_loff0[CPT], v_loff1[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = tid + NT * i;
    const int row = c >> 3, ch = c & 7;
    k_goff[i] = c * 16;
    k_loff[i] = swz_off<128>(row, ch);
    v_row[i] = row; v_coff[i] = ch * 16;
    // global chunk ch = keys 8 ch .. 8 ch + 7 of the row = step t = ch >> 1, half u = ch & 1: its first 4 keys belong to k-slot
    // h = 0 (LDS chunk 2 t), its last 4 to h = 1 (LDS chunk 2 t + 1), both at byte offset 8 u inside the chunk
    const int t = ch >> 1, u = ch & 1;
    v_loff0[i] = swz_off<128>(row, 2 * t + 0) + 8 * u;
    v_loff1[i] = swz_off<128>(row, 2 * t + 1) + 8 * u;
  }
  int fseg = 0, ftile = t_begin;
  int f_ntiles = (int)((p.seg[0].nk + BC - 1) / BC);
  while (ftile >= f_ntiles) { ftile -= f_ntiles; ++fseg; f_ntiles = (int)((p.seg[fseg].nk + BC - 1) / BC); }
  const int kvh = p.kv_heads > 0 ? bh % p.kv_heads : bh;
  const unsigned char* kptr = static_cast<const unsigned char*>(p.seg[fseg].k) + ((int64_t)kvh * p.seg[fseg].nk_pad + (int64_t)ftile * BC) * RB;
  const unsigned char* vptr = static_cast<const unsigned char*>(p.seg[fseg].vt) + ((int64_t)kvh * OVG_D * p.seg[fseg].nk_pad + (int64_t)ftile * BC) * 2;
  int64_t vstride = p.seg[fseg].nk_pad * 2;
  int cseg = fseg, ctile = ftile, c_ntiles = f_ntiles, c_nk = (int)p.seg[fseg].nk;
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
        kptr = static_cast<const unsigned char*>(sg.k) + (int64_t)kvh * sg.nk_pad * RB;
        vptr = static_cast<const unsigned char*>(sg.vt) + (int64_t)kvh * OVG_D * sg.nk_pad * 2;
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

  // fragment addresses: row 32 b + lq, chunk c ^ ((row >> 1) & 7); 32 b is a multiple of 16, so the swizzle term is lq's
  const int frag_row = lq * RB, sx = (lq >> 1) & 7;

  fetch();
  stash(0);
  __syncthreads();

  // S^T block pair s[qb] (32 keys x 2 x 32 queries) of key block kb of the tile in LDS at kl (C operand = -anchor), dead keys -inf.
  // One key block at a time: 32 accumulator registers live instead of 64, and the MFMA / exp phases are half as long.
  auto qk_block = [&](const unsigned char* kl, int kb, f32x16 (&s)[2], bool tail, int kv0) {
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
      const int coff = ((2 * ds + h) ^ sx) << 4;
      const u32x4 kf = *reinterpret_cast<const u32x4*>(kl + kb * 32 * RB + frag_row + coff);
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) s[qb] = mma32<T>(kf, qf[qb][ds], ds == 0 ? negc : s[qb]);
    }
    if (tail) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool dead = (kv0 + 32 * kb + 8 * (r >> 2) + 4 * h + (r & 3)) >= c_nk;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) s[qb][r] = dead ? -INFINITY : s[qb][r];
      }
    }
  };

  {
    // anchor: max over the first key tile (in LDS buffer 0 now) of both of the lane's rows; a row lives in lanes l and l ^ 32
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 s[2];
      qk_block(lds, kb, s, (ctile + 1) * BC > c_nk, ctile * BC);
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[qb][r]);
    }
    mx = fmaxf(mx, swap32_partner(mx)) + attn16::AnchorMargin<T>::value;
    negm[0] = negm[1] = -mx;
#pragma unroll
    for (int r = 0; r < 16; ++r) negc[r] = -mx;
  }

  int buf = 0;
  for (int j = 0; j < n_tiles; ++j) {
    const bool more = (j + 1) < n_tiles;
    if (more) fetch();
    const unsigned char* kl = lds + buf * (KT_B + VT_B);
    const unsigned char* vl = kl + KT_B;
    const int kv0 = ctile * BC;
    const bool tail = kv0 + BC > c_nk;

#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 s[2];
      qk_block(kl, kb, s, tail, kv0);               // the (rare) tail branch doubles as the scheduling fence (ovg_attn16.h header)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = __builtin_amdgcn_exp2f(s[qb][r]);
          s[qb][r] = e;
          part[qb][r & 3] += e;
        }
      // O^T += V^T P^T: 16-key step t = 2 kb + tt uses registers 8 tt .. 8 tt + 7 of this key block and V^T chunk 2 t + h
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int t = 2 * kb + tt;
        u32x4 pf[2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) pf[qb] = pack8<T>(s[qb], 8 * tt);
        const int voff = ((2 * t + h) ^ sx) << 4;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const u32x4 vf = *reinterpret_cast<const u32x4*>(vl + db * 32 * RB + frag_row + voff);
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) o[db][qb] = mma32<T>(vf, pf[qb], o[db][qb]);
        }
      }
    }

    if (++ctile == c_ntiles) {
      ctile = 0; ++cseg;
      if (cseg < p.nseg) { c_nk = (int)p.seg[cseg].nk; c_ntiles = (c_nk + BC - 1) / BC; }
    }
    if (more) stash(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const float own = (part[qb][0] + part[qb][1]) + (part[qb][2] + part[qb][3]);
    lsum[qb] = own + swap32_partner(own);            // both halves of a row: the full sum in both lanes
  }
}

// Normalise and store 64 rows per wave: final layout (token-major / head-major + optional lse) or split-KV partial
template <typename T>
OVG_DEV void write_out32(const ovg_attn_params& p, const f32x16 (&o)[2][2], const float (&lsum)[2], const float (&negm)[2],
                         const int bh, const int q0, const int sp, const int splits) {
  const int lane = threadIdx.x & 63, h = lane >> 5, lq = lane & 31;
  const int nq = (int)p.nq;
  const int bq = bh / OVG_H, hh = bh % OVG_H;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int q = q0 + qb * 32 + lq;
    if (q >= nq) continue;
    const float inv = 1.0f / lsum[qb];
    T* dst;
    if (splits > 1) {
      const int64_t row = ((int64_t)sp * p.BH + bh) * p.nq_pad + q;
      dst = static_cast<T*>(p.ws_part) + row * OVG_D;
      if (h == 0) p.ws_lse[row] = __builtin_amdgcn_logf(lsum[qb]) - negm[qb];
    } else {
      dst = p.out_bh_stride > 0 ? static_cast<T*>(p.out) + (int64_t)bh * p.out_bh_stride + (int64_t)q * p.ldo
                                : static_cast<T*>(p.out) + ((int64_t)bq * nq + q) * p.ldo + hh * OVG_D;
      if (p.lse != nullptr && h == 0) p.lse[(int64_t)bh * p.nq_pad + q] = __builtin_amdgcn_logf(lsum[qb]) - negm[qb];
    }
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {               // d = 32 db + 8 jj + 4 h + (0..3): 4 consecutive features
        const f32x16& a = o[db][qb];
        store4<T>(dst + 32 * db + 8 * jj + 4 * h, a[4 * jj] * inv, a[4 * jj + 1] * inv, a[4 * jj + 2] * inv, a[4 * jj + 3] * inv);
      }
  }
}

}  // namespace attn32

// MODE: 0 = speculative pass + verified fallback (16 x 16 lazy-rescale pass), 2 = fallback forced (tests)
template <typename T, int WAVES, int MODE>
__global__ __launch_bounds__(64 * WAVES, 2) void attn32_kernel(ovg_attn_params p, int nqt, int total_tiles, int splits, int per_split) {
  static_assert(sizeof(T) == 2, "16-bit types only");
  constexpr int RB = 128, KT_B = BC * RB, VT_B = OVG_D * RB, BQ = 64 * WAVES;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * (KT_B + VT_B)];
  const int wave = threadIdx.x >> 6;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int qt = lid % nqt, rest = lid / nqt;
  const int sp = rest % splits, bh = rest / splits;
  const int q0 = qt * BQ + wave * 64;
  const int t0 = sp * per_split;
  const int nt = (total_tiles - t0) < per_split ? (total_tiles - t0) : per_split;

  attn32::f32x16 o[2][2];
  float lsum[2], negm[2];
  attn32::run_tiles32<T, WAVES>(p, lds, bh, q0, t0, nt, o, lsum, negm);
  bool bad = MODE == 2;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    bad = bad || attn16::bad_sum(lsum[qb]);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) bad = bad || attn16::nonfinite(o[db][qb][r]);
  }
  if (__syncthreads_or(bad ? 1 : 0)) {
    f32x4 o16[4][4], lacc[4], nm[4];
    attn16::run_tiles<T, 4, WAVES, 0>(p, lds, bh, q0, t0, nt, o16, lacc, nm);
    attn16::write_out<T, 4>(p, o16, lacc, nm, bh, q0, sp, splits);
    return;
  }
  attn32::write_out32<T>(p, o, lsum, negm, bh, q0, sp, splits);
}
