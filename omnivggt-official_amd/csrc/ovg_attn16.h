// The tuned 16-bit flash-attention kernel (included by ovg_attn.hip): `attn16_kernel<T, QB, WAVES, MODE>`.
//
// Structure (kept from the measured iterations, see profiles/README.md):
//  * "swapped" MFMA formulation (ovg_attn.hip header): S'^T = K Q^T with -m_ref in the MFMA C operand, so
//    P = exp2(acc) needs no per-element subtract; P never leaves registers (the V^T LDS tile stores its keys
//    permuted inside each 32-key block so that a PV fragment is one conflict-free ds_read_b128);
//  * the row sums come out of the MATRIX pipe: one extra MFMA per (q block, 32-key step) with an all-ones A
//    operand accumulates sum_k P (PMC on the previous kernel: VALU-active 52 % vs MFMA-busy 33 %);
//  * __launch_bounds__(NT, 2): <= 256 VGPRs, VGPR-destination MFMAs (no v_accvgpr traffic); the file is
//    compiled with -fno-honor-nans (no canonicalising v_max before fmaxf of MFMA results);
//  * K / V^T tiles (64 keys) arrive by LDS-DMA (DMA = R > 0, every shipped launch): asm-issued global_load_lds_dwordx4
//    into a ring of R = 2 B + 1 LDS slots, B + 1 tiles ahead, behind counted vmcnt waits, one workgroup barrier every
//    B tiles (run_tiles below). DMA = 0 keeps the round-1 register-staged form (global loads of tile j+1 issued before
//    the MFMAs of tile j, written to the other of two buffers after them, one barrier per tile) as A/B variants.
//
// Two softmax bodies share that structure (run_tiles<..., SM>):
//  SM = 0  lazy-rescale online softmax: m_ref moves only when a row's tile max exceeds it by more than
//          RESCALE_THR = 8 log2 units; O, l and the pending S' are rescaled together before any exp (exact).
//  SM = 2  speculative ANCHORED softmax. The running max only exists to keep exp2() in range; with bf16 P
//          (f32 exponent range) and f32 accumulators, O = (sum P v) / (sum P) is invariant to m_ref. The pass
//          anchors m_ref per row at the max over the FIRST key tile (a prologue outside the loop) and then runs
//          a branch-free tile body: QK^T (anchor in C) -> v_exp_f32 -> pack -> row-sum MFMA + PV. No max, no
//          cross-lane traffic, no rescale: ~50 of the 184 VALU instructions per tile and the serial
//          QK^T -> max -> exp dependency are gone (+10...16 % measured).
//
// MODE 0 (bf16 default): run SM = 2, then every row checks on raw bits that l is within 2^+-100 and O is
//   finite -- exactly the condition that no exp2 overflowed and the row did not flush to zero (l >= max P).
//   If ANY row of the workgroup fails (__syncthreads_or), the whole workgroup recomputes with SM = 0, so the
//   result is always the exact softmax; a failed speculation only costs time (a row needs a logit spread of
//   more than ~100 log2 units = e^69 against its first tile; tests force it with spike / ramp inputs).
// MODE 1 (f16 default): SM = 0 only. f16 P saturates at 2^16: the f16 speculative build (anchor 4 log2 units
//   above the first-tile max, 20 units of head-room) pays the fallback on wide-spread logits (2x slower on the
//   A/B data), so it is available (MODE 0 on f16) but not the default.
// MODE 2 (tests): MODE 0 with the fallback forced.
//
// Measured dead ends, removed from the tree (logs under profiles/): LDS-DMA staging with natural-order V^T
// (-2 %), 32-key half bodies (-1.5 %), one shared anchor per lane (-1 %), QB = 3 (-5 %), a single merged
// rescale branch (-1 %), s_setprio around the MFMA clusters (+-0) or around the exp2/convert section (-4 %),
// 2-wave workgroups (-40 %), an 8-wave ping-pong with the two waves of a SIMD forced into MFMA / VALU antiphase
// by barriers (+-0 at S = 64, -17 % at S = 8; profiles/r01_probe_coexec.txt), and intra-wave software pipelining
// of exp against MFMA (compiler-scheduled, sched_barrier-pinned and sched_group_barrier 1:2:1 forms: -15...-90 %).
// hipcc trap: hoisting the (rare) tail-mask branch out of the tile body makes it one basic block; the
// scheduler then interleaves everything, runs out of registers and reloads Q fragments from scratch every
// tile -- and that reload's s_waitcnt vmcnt(0) drains the K/V prefetch (896 instead of 1154 TFLOP/s). The
// branch stays between the QK^T cluster and the exponentials on purpose.
#pragma once

namespace attn16 {

constexpr float RESCALE_THR = 8.0f;            // log2 units: p <= 256
constexpr uint32_t EXP_HI = 127 + 100, EXP_LO = 127 - 100;

OVG_DEV bool bad_sum(float l) {
  const uint32_t e = (__builtin_bit_cast(uint32_t, l) >> 23) & 0xffu;
  return e > EXP_HI || e < EXP_LO;
}
OVG_DEV bool nonfinite(float x) { return ((__builtin_bit_cast(uint32_t, x) >> 23) & 0xffu) == 0xffu; }

template <typename T> struct OnesFrag;
template <> struct OnesFrag<bf16_t> { static OVG_DEV u32x4 get() { return u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u}; } };
template <> struct OnesFrag<f16_t> { static OVG_DEV u32x4 get() { return u32x4{0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u}; } };
template <typename T> struct AnchorMargin { static constexpr float value = 0.f; };
template <> struct AnchorMargin<f16_t> { static constexpr float value = 4.f; };

template <typename T> OVG_DEV f32x4 mma_c(const u32x4& a, const u32x4& b, const f32x4& c);
template <> OVG_DEV f32x4 mma_c<bf16_t>(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <> OVG_DEV f32x4 mma_c<f16_t>(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// Max over the 4 lanes {l, l^16, l^32, l^48} of a q row with the gfx950 swap instructions
// (v_permlane32_swap: upper half of vdst <-> lower half of src; v_permlane16_swap: odd 16-lane rows of vdst
// <-> even rows of src). With the same value in both operands the two results hold the value and its partner's.
// NOTE: extract the two results into scalars first -- __builtin_bit_cast applied directly to `r[1]` (a
// vector-element lvalue) reads element 0 with this hipcc (ROCm 7.2).
OVG_DEV float swap32_partner_max(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned a = r[0], b = r[1];
  return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}
OVG_DEV float swap16_partner_max(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned a = r[0], b = r[1];
  return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}
OVG_DEV float xl_max4(float v) { return swap16_partner_max(swap32_partner_max(v)); }

template <typename T>
OVG_DEV u32x4 pack2(const f32x4 a, const f32x4 b) {
  T v[8];
  v[0] = TT<T>::from_f32(a[0]); v[1] = TT<T>::from_f32(a[1]); v[2] = TT<T>::from_f32(a[2]); v[3] = TT<T>::from_f32(a[3]);
  v[4] = TT<T>::from_f32(b[0]); v[5] = TT<T>::from_f32(b[1]); v[6] = TT<T>::from_f32(b[2]); v[7] = TT<T>::from_f32(b[3]);
  u32x4 r;
  __builtin_memcpy(&r, v, 16);
  return r;
}

// P tile -> (hi, lo) f16 fragments of the split-f16 mode: hi = f16(p), lo = f16(p - hi); p <= 2^8 (lazy rescale), no saturation needed.
// OVG_ATTN_X3_SPLIT 1 (r05): 3 VALU per value pair instead of ~11 -- v_cvt_pk_f16_f32 for the hi pair, then v_fma_mixlo/mixhi_f16
// compute f16(p * 1.0 - f32(hi half)) straight from the packed hi register (the f32 difference is exact, so the bits equal form 0's).
#ifndef OVG_ATTN_X3_SPLIT
#define OVG_ATTN_X3_SPLIT 1
#endif
// Products of the PV contraction of the split-f16 mode (template value X3 of the kernels; round 6, profiles/r06_f32x_pv_terms_ab.txt): 3 = P_hi V_lo
// + P_lo V_hi + P_hi V_hi (the mode: 1.07e-5 of the f32 mode at layer 23 of 64 views, 28.1 frames/s); 2 = without P_lo V_hi (2.96e-5 there and
// 32.6 frames/s, but 1.0e-4 on the camera token of the 64-view depth-1 parity case: opt-in through ovg_attn_params.variant 92, outside the
// <= 1e-4 contract); P_hi V_hi alone measured 4.6e-5 / 35.2 frames/s and is not offered.
template <bool WANT_LO = true>
OVG_DEV void pack2_hilo(const f32x4 a, const f32x4 b, u32x4& hi, u32x4& lo) {
#if OVG_ATTN_X3_SPLIT
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  uint32_t h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[i]) : "v"(v[2 * i]), "v"(v[2 * i + 1]));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l[i]) : "v"(v[2 * i]), "v"(h[i]));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l[i]) : "v"(v[2 * i + 1]), "v"(h[i]));
  }
  hi = u32x4{h[0], h[1], h[2], h[3]};
  lo = u32x4{l[0], l[1], l[2], l[3]};
  // hipcc pads nothing for inline asm (cdna_hip_programming.md 5.7 item 2): a VGPR written by the statements above must not feed an MFMA in
  // the next two issue slots. Rounds 4-5 relied on the fragment reads that happened to sit in between; this statement makes it a property of
  // the code (found in round 6: with the P_lo product removed in a lab build the convert landed directly in front of its MFMA -- garbage).
  if constexpr (WANT_LO) asm volatile("s_nop 1" : "+v"(hi), "+v"(lo));
  else asm volatile("s_nop 1" : "+v"(hi));
#else
  f16_t h[8], l[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = static_cast<f16_t>(a[i]); l[i] = static_cast<f16_t>(a[i] - static_cast<float>(h[i]));
    h[4 + i] = static_cast<f16_t>(b[i]); l[4 + i] = static_cast<f16_t>(b[i] - static_cast<float>(h[4 + i]));
  }
  __builtin_memcpy(&hi, h, 16);
  __builtin_memcpy(&lo, l, 16);
#endif
}

// lacc += sa + sb on the packed-f32 adder (r05, OVG_ATTN_X3_PKSUM 1): 4 v_pk_add_f32 per 8 values instead of 8 v_add_f32; the four registers
// of lacc are lane-partial sums that are reduced once behind the loop, so only the ORDER of the f32 additions differs from form 0
#ifndef OVG_ATTN_X3_PKSUM
#define OVG_ATTN_X3_PKSUM 1
#endif
typedef float f32x2_t __attribute__((ext_vector_type(2)));
OVG_DEV void rowsum_acc(f32x4& lacc, const f32x4 sa, const f32x4 sb) {
#if OVG_ATTN_X3_PKSUM
  f32x2_t l0 = {lacc[0], lacc[1]}, l1 = {lacc[2], lacc[3]};
  const f32x2_t a0 = {sa[0], sa[1]}, a1 = {sa[2], sa[3]}, b0 = {sb[0], sb[1]}, b1 = {sb[2], sb[3]};
  asm("v_pk_add_f32 %0, %0, %1" : "+v"(l0) : "v"(a0));
  asm("v_pk_add_f32 %0, %0, %1" : "+v"(l1) : "v"(a1));
  asm("v_pk_add_f32 %0, %0, %1" : "+v"(l0) : "v"(b0));
  asm("v_pk_add_f32 %0, %0, %1" : "+v"(l1) : "v"(b1));
  lacc = f32x4{l0[0], l0[1], l1[0], l1[1]};
#else
  lacc += sa + sb;
#endif
}

// One pass over all key tiles of all segments for the wave's QB x 16 query rows; leaves the un-normalised
// O^T in o, the row sums in lacc (every register of lacc[qb] holds the full sum of row q0 + 16 qb + lane&15) and the
// negated reference maximum in negm (P = exp2(s + negm)): log2 sum_k exp2(s) = log2(lacc) - negm.
// (lds_dma16, the asm-issued LDS-DMA transfer the staging below uses, lives in ovg_common.h)
// X3 (OVG_F16X2, the split-f16 parity mode; T = f16_t, SM = 0, VSUM, DMA): q, K and V^T are (hi, lo) plane pairs -- a ring slot holds
// [K hi | V^T hi | K lo | V^T lo] -- and both contractions run three MFMAs per product into the same f32 accumulator, small terms first:
//   S = K_lo Q_hi + K_hi Q_lo + K_hi Q_hi,   O += V_lo P_hi + V_hi P_lo + V_hi P_hi,   P = (hi, lo) split of exp2(S - m) in registers;
// the row sums are exact f32 sums of P on the VALU (the matrix pipe is the bound here: 104 MFMAs against ~100 VALU per tile at QB = 2).
// r05 (profiles/r05_attention_x3_ab.txt): the (hi, lo) split through v_cvt_pk_f16_f32 + v_fma_mixlo/mixhi_f16 and packed-f32 row sums cut the
// VALU work of a tile from ~190 to ~140 instructions for +1.9 ... 2.1 %; a 5-slot ring with a barrier every 2 tiles (all 160 KB of LDS) measured
// -0.6 ... -1.2 %, and a loop that issues QK^T of tile j + 1 inside the exp / split block of tile j (4-slot ring, 191 VGPRs, no spill,
// MFMA / VALU interleaved by hipcc as intended) -2.2 ... -2.6 % -- the kernel issues 1400 TFLOP/s of f16 MFMAs, the same rate as the bf16
// kernel with its row-sum MFMAs: neither VALU issue nor phase alignment of the two waves of a SIMD is what holds it there (section 5.1).
template <typename T, int QB, int WAVES, int SM, bool VSUM = false, int DMA = 0, int X3 = 0>   // VSUM: row sums on the VALU (experiment, variant 31) instead of the ones-MFMA; DMA: 0 = register staging, R = 2 B + 1 (3, 5, 7, 9): K / V^T tiles by LDS-DMA into a ring of R slots, B + 1 tiles ahead, one workgroup barrier every B tiles
OVG_DEV void run_tiles(const ovg_attn_params& p, unsigned char* lds, const int bh, const int q0, const int t_begin, const int total_tiles,
                       f32x4 (&o)[QB][4], f32x4 (&lacc)[QB], f32x4 (&negm)[QB]) {   // key tiles [t_begin, t_begin + total_tiles) of the flattened segment list
  constexpr int NT = 64 * WAVES;
  constexpr int RB = 128, KT_B = BC * RB, VT_B = OVG_D * RB, CPT = 512 / NT;
  static_assert(!X3 || (DMA > 0 && VSUM && SM == 0 && std::is_same<T, f16_t>::value), "split-f16: f16 planes, LDS-DMA staging, lazy rescale, VALU row sums");
  constexpr int PLANE_B = KT_B + VT_B;             // one plane of a ring slot: K tile + V^T tile
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int g = lane >> 4, lr = lane & 15;
  const int nq = (int)p.nq;

  u32x4 qf[QB][2], ql[X3 ? QB : 1][2];
  {
    const unsigned char* qbase = static_cast<const unsigned char*>(p.q) + (int64_t)bh * p.nq_pad * RB;
    const unsigned char* qbase_lo = static_cast<const unsigned char*>(p.q_lo) + (int64_t)bh * p.nq_pad * RB;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      int q = q0 + qb * 16 + lr; q = q < nq ? q : nq - 1;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        qf[qb][kk] = *reinterpret_cast<const u32x4*>(qbase + (int64_t)q * RB + (4 * kk + g) * 16);
        if constexpr (X3) ql[qb][kk] = *reinterpret_cast<const u32x4*>(qbase_lo + (int64_t)q * RB + (4 * kk + g) * 16);
      }
    }
  }
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    negm[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    lacc[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[qb][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  u32x4 ones = OnesFrag<T>::get();
  // The all-ones A operand of the row-sum MFMAs must LIVE in registers: as a known constant hipcc may rematerialise it inside the loop
  // (s_mov + v_mov_b64 directly in front of its use) -- harmless in front of a compiler-emitted MFMA, which gets its wait states, but in
  // front of an `asm volatile` MFMA of the pinned body that is a VALU-write -> MFMA-read hazard nobody pads: the MFMA reads the stale
  // register. (r04: exactly this broke the every-segment loop form -- 100 % verified-fallback re-runs.) Opaque to the optimiser from here on.
  asm volatile("" : "+v"(ones));

  // ---- staging ------------------------------------------------------------------------------------------------------------
  // Register path (DMA = false): global loads of tile j + 1 are issued before the MFMAs of tile j and written to the other LDS
  // buffer after them (two buffers of K tile + V^T tile).  DMA path: both tiles go global -> LDS by LDS-DMA into a ring of three
  // slots, TWO tiles ahead, no staging registers and no ds_write; every iteration issues exactly NDMA transfers per wave, so one
  // counted s_waitcnt vmcnt(NDMA) before the barrier means "everything but the newest tile has landed".
  //   lane l of a transfer -> LDS row l / 8, chunk position l % 8 of that row <- source chunk (l % 8) ^ swizzle(row), i.e. the LDS image
  //   is the XOR-swizzled tile the fragment reads expect (K rows = keys, V^T rows = features with the vt_pos16 key order).
  //   Barrier period B = (DMA - 1) / 2: the waves only meet every B tiles (between barriers they drift apart by up to B tiles, so
  //   their MFMA-heavy and exp-heavy phases stop coinciding). At the barrier after tile j - 1 (j = 0 mod B) the tiles j .. j + B - 1
  //   must have landed: with the transfers issued B + 1 tiles ahead that is "all but the newest tile" = vmcnt(NDMA). A slot is
  //   rewritten (tile j + B + 1 -> the slot of tile j - B) only after a barrier that follows the last read of tile j - B: R >= 2 B + 1.
  constexpr int SLOT_B = (X3 ? 2 : 1) * PLANE_B, NDMA = (X3 ? 2 : 1) * 2 * (8 / WAVES), BARP = DMA ? (DMA - 1) / 2 : 1;
  static_assert(DMA == 0 || (DMA == 2 * BARP + 1 && BARP >= 1), "ring = 2 * barrier period + 1");
  u32x4 rk[DMA ? 1 : CPT], rv[DMA ? 1 : CPT];
  int k_goff[CPT], v_row[CPT], v_coff[CPT], k_loff[CPT], v_loff[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = tid + NT * i;
    const int row = c >> 3, ch = c & 7;
    k_goff[i] = c * 16;
    k_loff[i] = swz_off<128>(row, ch);
    v_row[i] = row; v_coff[i] = ch * 16;
    v_loff[i] = swz_off<128>(row, ch);             // global V^T rows already hold the fragment order (vt_pos16)
  }
  int fseg = 0, ftile = t_begin;                   // split-KV: this pass starts t_begin tiles into the segment list
  int f_ntiles = (int)((p.seg[0].nk + BC - 1) / BC);
  while (ftile >= f_ntiles) { ftile -= f_ntiles; ++fseg; f_ntiles = (int)((p.seg[fseg].nk + BC - 1) / BC); }
  const int kvh = p.kv_heads > 0 ? bh % p.kv_heads : bh;   // head-parallel sharding: (source rank, head) pairs share K / V^T
  const unsigned char* kptr = static_cast<const unsigned char*>(p.seg[fseg].k) + ((int64_t)kvh * p.seg[fseg].nk_pad + (int64_t)ftile * BC) * RB;
  const unsigned char* vptr = static_cast<const unsigned char*>(p.seg[fseg].vt) + ((int64_t)kvh * OVG_D * p.seg[fseg].nk_pad + (int64_t)ftile * BC) * 2;
  int64_t vstride = p.seg[fseg].nk_pad * 2;       // bytes between V^T rows (d)
  // split-f16: byte distance from the hi to the lo plane of the current segment's K / V^T
  int64_t kdl = X3 ? static_cast<const unsigned char*>(p.seg[fseg].k_lo) - static_cast<const unsigned char*>(p.seg[fseg].k) : 0;
  int64_t vdl = X3 ? static_cast<const unsigned char*>(p.seg[fseg].vt_lo) - static_cast<const unsigned char*>(p.seg[fseg].vt) : 0;
  const int seg0 = fseg, tile0 = ftile;
  auto next_tile = [&]() {                          // advance the fetch cursor (kptr / vptr / vstride) by one tile, across segments
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
        if constexpr (X3) {
          kdl = static_cast<const unsigned char*>(sg.k_lo) - static_cast<const unsigned char*>(sg.k);
          vdl = static_cast<const unsigned char*>(sg.vt_lo) - static_cast<const unsigned char*>(sg.vt);
        }
      }
    }
  };
  auto fetch = [&]() {
#pragma unroll
    for (int i = 0; i < (DMA ? 1 : CPT); ++i) {
      rk[i] = *reinterpret_cast<const u32x4*>(kptr + k_goff[i]);
      rv[i] = *reinterpret_cast<const u32x4*>(vptr + v_row[i] * vstride + v_coff[i]);
    }
    next_tile();
  };
  auto stash = [&](int buf) {
    unsigned char* kl = lds + buf * (KT_B + VT_B);
    unsigned char* vl = kl + KT_B;
#pragma unroll
    for (int i = 0; i < (DMA ? 1 : CPT); ++i) {
      *reinterpret_cast<u32x4*>(kl + k_loff[i]) = rk[i];
      *reinterpret_cast<u32x4*>(vl + v_loff[i]) = rv[i];
    }
  };
  // DMA path: per-lane source offsets of the wave's 8 / WAVES transfers per tile and the wave-uniform LDS destinations
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
  int d_row[8 / WAVES], d_ch16[8 / WAVES];
#pragma unroll
  for (int i = 0; i < 8 / WAVES; ++i) {
    d_row[i] = (64 / WAVES) * wave_u + 8 * i + (lane >> 3);
    d_ch16[i] = ((lane & 7) ^ ((d_row[i] >> 1) & 7)) << 4;
  }
  int issued = 0;                                  // tiles handed to the DMA engine so far (the cursor stops on the last tile:
                                                   // the tail iterations re-load it into slots nobody reads, keeping NDMA per iteration)
  auto dma_issue = [&](int slot) {
    const uint32_t dst = lds_base + slot * SLOT_B + (64 / WAVES) * wave_u * RB;
#pragma unroll
    for (int i = 0; i < 8 / WAVES; ++i) lds_dma16(kptr + d_row[i] * RB + d_ch16[i], dst + i * 8 * RB);
#pragma unroll
    for (int i = 0; i < 8 / WAVES; ++i) lds_dma16(vptr + d_row[i] * vstride + d_ch16[i], dst + KT_B + i * 8 * RB);
    if constexpr (X3) {
#pragma unroll
      for (int i = 0; i < 8 / WAVES; ++i) lds_dma16(kptr + kdl + d_row[i] * RB + d_ch16[i], dst + PLANE_B + i * 8 * RB);
#pragma unroll
      for (int i = 0; i < 8 / WAVES; ++i) lds_dma16(vptr + vdl + d_row[i] * vstride + d_ch16[i], dst + PLANE_B + KT_B + i * 8 * RB);
    }
    if (++issued < total_tiles) next_tile();
  };

  int cseg = seg0, ctile = tile0;
  int c_ntiles = f_ntiles;
  int c_nk = (int)p.seg[seg0].nk;
  const int sx = lr >> 1;
  const int frag_row = lr * 128;
  const int coff0 = ((0 + g) ^ sx) << 4, coff1 = ((4 + g) ^ sx) << 4;

  if constexpr (DMA) {
#pragma unroll
    for (int t = 0; t < BARP + 1; ++t) dma_issue(t);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");     // tiles 0 .. B - 1 have landed (this wave's share), tile B may be in flight
    __builtin_amdgcn_s_barrier();
  } else {
    fetch();
    stash(0);
    __syncthreads();
  }

  // S'^T blocks s[kt][qb] (keys 16 kt .. 16 kt + 15) of the tile in LDS at kl, dead keys masked to -inf
  auto qk_tile = [&](const unsigned char* kl, f32x4 (&s)[4][QB], bool tail, int kv0) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const u32x4 k0 = *reinterpret_cast<const u32x4*>(kl + kt * 2048 + frag_row + coff0);
      const u32x4 k1 = *reinterpret_cast<const u32x4*>(kl + kt * 2048 + frag_row + coff1);
      if constexpr (X3) {
        const u32x4 k0l = *reinterpret_cast<const u32x4*>(kl + PLANE_B + kt * 2048 + frag_row + coff0);
        const u32x4 k1l = *reinterpret_cast<const u32x4*>(kl + PLANE_B + kt * 2048 + frag_row + coff1);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          s[kt][qb] = mma_c<T>(k0l, qf[qb][0], negm[qb]);
          s[kt][qb] = mma_c<T>(k1l, qf[qb][1], s[kt][qb]);
          s[kt][qb] = mma_c<T>(k0, ql[qb][0], s[kt][qb]);
          s[kt][qb] = mma_c<T>(k1, ql[qb][1], s[kt][qb]);
          s[kt][qb] = mma_c<T>(k0, qf[qb][0], s[kt][qb]);
          s[kt][qb] = mma_c<T>(k1, qf[qb][1], s[kt][qb]);
        }
        continue;
      }
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        s[kt][qb] = mma_c<T>(k0, qf[qb][0], negm[qb]);
        s[kt][qb] = mma_c<T>(k1, qf[qb][1], s[kt][qb]);
      }
    }
    if (tail) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool dead = (kv0 + 16 * kt + 4 * g + r) >= c_nk;
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) s[kt][qb][r] = dead ? -INFINITY : s[kt][qb][r];
        }
    }
  };
  auto row_max = [&](const f32x4 (&s)[4][QB], int qb) {
    float mx = fmaxf(s[0][qb][0], s[0][qb][1]);
    mx = fmaxf(fmaxf(mx, s[0][qb][2]), s[0][qb][3]);
#pragma unroll
    for (int kt = 1; kt < 4; ++kt) {
      mx = fmaxf(fmaxf(mx, s[kt][qb][0]), s[kt][qb][1]);
      mx = fmaxf(fmaxf(mx, s[kt][qb][2]), s[kt][qb][3]);
    }
    return xl_max4(mx);
  };
  auto exp_qb = [&](f32x4 (&s)[4][QB], int qb) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) s[kt][qb][r] = __builtin_amdgcn_exp2f(s[kt][qb][r]);
  };
  // O^T += V^T P^T ; l += 1^T P^T for the 32-key step u (keys 32u + 16 (j>>2) + 4g + (j&3) in slot j)
  auto pv_step = [&](const unsigned char* vl, int u, const f32x4 (&sa)[QB], const f32x4 (&sb)[QB]) {
    if constexpr (X3) {
      u32x4 ph[QB], pl[QB];
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        // The P values come out of v_exp_f32 (a transcendental: its result may not be read by a VALU instruction in the next issue slot) and
        // go into INLINE-ASM VALU statements (pack2_hilo, rowsum_acc), which hipcc does not pad (cdna_hip_programming.md 5.7 item 2): whether
        // an exp landed directly in front of its asm reader was up to the scheduler. Rounds 4-5 passed their 1e-5 checks with the schedule they
        // happened to get; in round 6 an unrelated edit moved one exp and the split-f16 attention dropped to f16-level errors (1e-3). One opaque
        // wait state between the exps and every asm reader makes it a property of the code.
        f32x4 pa = sa[qb], pb = sb[qb];
        asm volatile("s_nop 0" : "+v"(pa), "+v"(pb));
        pack2_hilo<(X3 >= 3)>(pa, pb, ph[qb], pl[qb]);
        rowsum_acc(lacc[qb], pa, pb);                 // exact f32 row sums, lane-partial, reduced once after the loop
      }
      const int voff = ((4 * u + g) ^ sx) << 4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const u32x4 vf = *reinterpret_cast<const u32x4*>(vl + dt * 2048 + frag_row + voff);
        const u32x4 vfl = *reinterpret_cast<const u32x4*>(vl + PLANE_B + dt * 2048 + frag_row + voff);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          // X3 = 3: P_hi V_lo + P_lo V_hi + P_hi V_hi; X3 = 2 (opt-in, variant 92): without P_lo V_hi -- P is then an 11-bit f16 value per
          // key, V stays (hi, lo), the row sums stay exact f32
          o[qb][dt] = mma_c<T>(vfl, ph[qb], o[qb][dt]);
          if constexpr (X3 >= 3) o[qb][dt] = mma_c<T>(vf, pl[qb], o[qb][dt]);
          o[qb][dt] = mma_c<T>(vf, ph[qb], o[qb][dt]);
        }
      }
      return;
    }
    u32x4 pf[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      pf[qb] = pack2<T>(sa[qb], sb[qb]);
      if constexpr (VSUM) lacc[qb] += sa[qb] + sb[qb];   // lane-partial sums (4 registers x 4 lanes per row), reduced once after the loop
      else lacc[qb] = mma_c<T>(ones, pf[qb], lacc[qb]);
    }
    const int voff = ((4 * u + g) ^ sx) << 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const u32x4 vf = *reinterpret_cast<const u32x4*>(vl + dt * 2048 + frag_row + voff);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) o[qb][dt] = mma_c<T>(vf, pf[qb], o[qb][dt]);
    }
  };

  if constexpr (SM == 2) {
    // anchor: m_ref = row max over the first key tile (tile 0 is in LDS buffer 0 now)
    f32x4 s[4][QB];
    qk_tile(lds, s, (tile0 + 1) * BC > c_nk, tile0 * BC);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const float mx = row_max(s, qb) + AnchorMargin<T>::value;
      negm[qb] = f32x4{-mx, -mx, -mx, -mx};
    }
  }

  // ---- order-pinned tile body of the bf16 speculative pass (r04; generator + rationale: tools/gen_attn_body.py) -----------------
  // Every MFMA / v_exp_f32 / v_cvt_pk_bf16_f32 / ds_read_b128 / s_waitcnt of a FULL tile is one `asm volatile` statement: hipcc keeps
  // their order and only allocates the registers. One exp per MFMA, evenly spread (schedule v2; v2q2 for QB = 2), instead of hipcc's
  // [32 MFMA][32 exp + 16 cvt][40 MFMA + 32 exp] clusters: +1.3 ... 2.3 % at 64 views, +5 % at 8 views, bit-identical results
  // (same arithmetic, same order per accumulator). The hazard distances hipcc no longer inserts are checked by the generator.
#define PB_DSR(d, a, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(d) : "v"(a))
#define PB_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")")
#define PB_NOP() asm volatile("s_nop 0")
#define PB_EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define PB_CVT(d, a, b) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#define PB_MFMA_NEW(d, a, b, c) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c))
#define PB_MFMA_ACC(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define PB_SPLIT(kt, qb) do { s[kt][qb][0] = t[kt][qb][0]; s[kt][qb][1] = t[kt][qb][1]; s[kt][qb][2] = t[kt][qb][2]; s[kt][qb][3] = t[kt][qb][3]; } while (0)
#define PB_PACK(u, qb) pf[u][qb] = u32x4{pw[u][qb][0], pw[u][qb][1], pw[u][qb][2], pw[u][qb][3]}
  constexpr bool PIPE = SM == 2 && (QB == 4 || QB == 2) && DMA > 0 && !VSUM && std::is_same<T, bf16_t>::value;
  auto pipe_tile = [&](int slot) {
    if constexpr (PIPE) {
      const uint32_t kb = lds_base + slot * SLOT_B + frag_row;
      const uint32_t ka0 = kb + coff0, ka1 = kb + coff1;
      const uint32_t va0 = kb + KT_B + (((0 + g) ^ sx) << 4), va1 = kb + KT_B + (((4 + g) ^ sx) << 4);
      u32x4 K[4][2], V[2][4], pf[2][QB];
      f32x4 t[4][QB];
      float s[4][QB][4];
      uint32_t pw[2][QB][4];
      if constexpr (QB == 4) {
#include "ovg_attn16_body_q4.inc"
      } else {
#include "ovg_attn16_body_q2.inc"
      }
    }
  };
#undef PB_DSR
#undef PB_LGKM
#undef PB_NOP
#undef PB_EXP
#undef PB_CVT
#undef PB_MFMA_NEW
#undef PB_MFMA_ACC
#undef PB_SPLIT
#undef PB_PACK

  int since_barrier = 0;
  int buf = 0;                                     // register path: LDS buffer of tile j; DMA path: ring slot of tile j
  // the loop body as a generic lambda, instantiated twice -- the order-pinned body for the leading FULL tiles of a single-segment
  // launch, the compiler-scheduled body for the rest (the masked last tile; every tile of a multi-segment launch) -- so that the
  // hot loop holds ONE body (with both in one loop hipcc joins their register assignments with ~50 copies and spills O)
  auto tile_iter = [&](int j, auto use_pipe) {
    const bool more = (j + 1) < total_tiles;
    if constexpr (DMA) {
      dma_issue(buf >= BARP ? buf - BARP : buf - BARP + DMA);   // tile j + B + 1 -> slot (j - B) % R
    } else {
      if (more) fetch();
    }
    const unsigned char* kl = lds + buf * SLOT_B;
    const unsigned char* vl = kl + KT_B;
    const int kv0 = ctile * BC;

    f32x4 s[4][QB];
    if constexpr (decltype(use_pipe)::value) {
      pipe_tile(buf);
    } else {
    qk_tile(kl, s, kv0 + BC > c_nk, kv0);          // the tail branch doubles as the scheduling fence (header)
    if constexpr (SM == 2) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) exp_qb(s, qb);
    } else {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        const float mx = row_max(s, qb);
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
        exp_qb(s, qb);
      }
    }
    pv_step(vl, 0, s[0], s[1]);
    pv_step(vl, 1, s[2], s[3]);
    }

    if (++ctile == c_ntiles) {
      ctile = 0; ++cseg;
      if (cseg < p.nseg) { c_nk = (int)p.seg[cseg].nk; c_ntiles = (c_nk + BC - 1) / BC; }
    }
    if constexpr (DMA) {
      // every B tiles: all but the newest tile's transfers (this wave's share) have landed -> a bare barrier publishes the next B
      // tiles and retires the last B. No __syncthreads (its fence drains vmcnt); the waves never write LDS themselves.
      if (BARP == 1 || ++since_barrier == BARP) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        __builtin_amdgcn_s_barrier();
        since_barrier = 0;
      }
      buf = buf == DMA - 1 ? 0 : buf + 1;
    } else {
      if (more) stash(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  };
  int j_all = 0;
  // OVG_ATTN_PIPE_LOOP (A/B builds, tools/probes/build_alt.py): 2 (shipped) = the pinned body on every FULL tile of every segment,
  // 1 = on the leading full tiles of single-segment launches only (the form of the first r04 passes), 0 = compiler-scheduled body everywhere.
  // History of form 2 (profiles/r04_attn_ab_loop_forms.txt): its first build computed a WRONG speculative pass -- with the outer loop around
  // the two bodies hipcc rematerialised the constant all-ones operand of the row-sum MFMAs inside the hot loop (s_mov + v_mov_b64 directly
  // in front of the asm MFMA that reads it: a VALU-write -> MFMA-read hazard nobody pads for inline asm), every row sum was garbage, every
  // workgroup failed its verification and re-ran (688 of 688, 2.1x slower, results exact, all parity tests green: found by the fallback
  // counter). With `ones` made opaque to the optimiser (above) the form is bit-identical to form 1 and +5.0 % on the 8-segment per-rank
  // launch of the view-sharded run (-0.3 ... -0.4 % on single-segment launches: noise level).
  // tests/test_attn_body_generator.py::test_pinned_hot_loops_hold_no_compiler_copies guards the shipped kernels against such copies.
#ifndef OVG_ATTN_PIPE_LOOP
#define OVG_ATTN_PIPE_LOOP 2
#endif
#if OVG_ATTN_PIPE_LOOP == 1
  if constexpr (PIPE) {
    // full tiles in front of the first masked one: single segment, tiles tile0 .. ; tile t is full iff (t + 1) * BC <= nk
    int n_full = p.nseg == 1 ? (int)(p.seg[0].nk / BC) - tile0 : 0;
    n_full = n_full < total_tiles ? n_full : total_tiles;
    for (; j_all < n_full; ++j_all) tile_iter(j_all, std::true_type{});
    asm volatile("s_nop 15\n\ts_nop 15");     // asm MFMA results -> VALU / builtin readers behind the loop (hipcc does not see the hazard)
  }
#elif OVG_ATTN_PIPE_LOOP == 2
  if constexpr (PIPE) {
    // Segment by segment (one segment on a single GPU; one per rank / source after a view-sharded exchange): the FULL tiles of the
    // segment in a loop of the pinned body, then its masked last tile, if any, through the compiler-scheduled body (one call per segment)
    while (j_all < total_tiles) {
      int n_full = c_nk / BC - ctile;
      n_full = n_full < total_tiles - j_all ? n_full : total_tiles - j_all;
      for (int i = 0; i < n_full; ++i, ++j_all) tile_iter(j_all, std::true_type{});
      asm volatile("s_nop 15\n\ts_nop 15");   // asm MFMA results -> VALU / builtin readers behind the loop
      if (j_all < total_tiles && (ctile + 1) * BC > c_nk) { tile_iter(j_all, std::false_type{}); ++j_all; }
    }
  }
#endif
  for (; j_all < total_tiles; ++j_all) tile_iter(j_all, std::false_type{});
  if constexpr (DMA) __syncthreads();              // drain the tail transfers before the ring is reused (fallback pass) or the workgroup ends
  if constexpr (VSUM) {
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const float t = quad16_sum(lacc[qb][0] + lacc[qb][1] + lacc[qb][2] + lacc[qb][3]);
      lacc[qb] = f32x4{t, t, t, t};
    }
  }
}

// Normalise and store a wave's QB x 16 rows: the caller's final layout (token-major or head-major, + optional log-sum-exp)
// or, in a split-KV pass, the head-major partial [split][entry][part_rows][64] (f32 since round 5: the bf16 partials of rounds 2-4 were an
// extra rounding point, 6.9e-3 between a split and an unsplit launch of the same call; now ~1e-6 before the final rounding) with its
// log-sum-exp. Partial rows are relative to q_row0, the first row of the launch (the key-split tail launch covers rows [q_row0, nq) only).
template <typename T, int QB, int X3 = 0>
OVG_DEV void write_out(const ovg_attn_params& p, const f32x4 (&o)[QB][4], const f32x4 (&lacc)[QB], const f32x4 (&negm)[QB],
                       const int bh, const int q0, const int sp, const int splits, const int q_row0 = 0, const int part_rows = 0) {
  const int lane = threadIdx.x & 63, g = lane >> 4, lr = lane & 15;
  const int nq = (int)p.nq;
  const int bq = bh / OVG_H, hh = bh % OVG_H;
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float inv = 1.0f / lacc[qb][0];        // every row of the ones-MFMA holds the full row sum
    const int q = q0 + qb * 16 + lr;
    if (q < nq) {
      if (splits > 1) {                           // partial pass: head-major f32 [split][entry][part_rows][64] + its log-sum-exp
        const int64_t row = ((int64_t)sp * p.BH + bh) * part_rows + (q - q_row0);
        float* dst = static_cast<float*>(p.ws_part) + row * OVG_D + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4*>(dst + 16 * dt) = o[qb][dt] * inv;
        if (g == 0) p.ws_lse[row] = __builtin_amdgcn_logf(lacc[qb][0]) - negm[qb][0];
        continue;
      }
      const int64_t off = p.out_bh_stride > 0 ? (int64_t)bh * p.out_bh_stride + (int64_t)q * p.ldo + 4 * g
                                              : ((int64_t)bq * nq + q) * p.ldo + hh * OVG_D + 4 * g;
      T* dst = static_cast<T*>(p.out) + off;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        if constexpr (X3) store4_hilo(dst + 16 * dt, static_cast<f16_t*>(p.out_lo) + off + 16 * dt, o[qb][dt][0] * inv, o[qb][dt][1] * inv, o[qb][dt][2] * inv, o[qb][dt][3] * inv);
        else store4<T>(dst + 16 * dt, o[qb][dt][0] * inv, o[qb][dt][1] * inv, o[qb][dt][2] * inv, o[qb][dt][3] * inv);
      }
      if (p.lse != nullptr && g == 0) p.lse[(int64_t)bh * p.nq_pad + q] = __builtin_amdgcn_logf(lacc[qb][0]) - negm[qb][0];
    }
  }
}

}  // namespace attn16

// MODE: 0 = speculative anchored softmax + verified fallback, 1 = lazy-rescale only, 2 = forced fallback (tests)
template <typename T, int QB, int WAVES, int MODE, int OCC = 2, bool VSUM = false, int DMA = 0, int X3 = 0>   // OCC: minimum waves per SIMD the register allocation must allow; X3: split-f16 planes (run_tiles)
__global__ __launch_bounds__(64 * WAVES, OCC) void attn16_kernel(ovg_attn_params p, int nqt, int total_tiles, int splits, int per_split, int q_row0, int part_rows) {   // q tiles [0, nqt) of the rows starting at q_row0; part_rows: rows per entry of the split workspace
  static_assert(sizeof(T) == 2, "16-bit types only");
  constexpr int RB = 128, KT_B = BC * RB, VT_B = OVG_D * RB, BQ = 16 * QB * WAVES;
  __shared__ __attribute__((aligned(16))) unsigned char lds[(DMA ? DMA : 2) * (X3 ? 2 : 1) * (KT_B + VT_B)];

  const int tid = threadIdx.x, wave = tid >> 6;
  // logical id -> (batch entry, key split, q tile), q tile fastest: the workgroups that run side by side on an XCD share
  // one (entry, split) = one K / V^T range, which with splits is 1 / splits of the head's keys
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int qt = lid % nqt, rest = lid / nqt;
  const int sp = rest % splits, bh = rest / splits;
  const int q0 = q_row0 + qt * BQ + wave * 16 * QB;
  const int t0 = sp * per_split;
  const int nt = (total_tiles - t0) < per_split ? (total_tiles - t0) : per_split;   // >= 1: the host sizes splits so that every pass has keys

  f32x4 o[QB][4], lacc[QB], negm[QB];
  if constexpr (MODE == 1) {
    attn16::run_tiles<T, QB, WAVES, 0, VSUM, DMA, X3>(p, lds, bh, q0, t0, nt, o, lacc, negm);
  } else {
    static_assert(!X3, "split-f16 runs the lazy-rescale body only");
    attn16::run_tiles<T, QB, WAVES, 2, VSUM, DMA>(p, lds, bh, q0, t0, nt, o, lacc, negm);
    bool bad = MODE == 2;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      bad = bad || attn16::bad_sum(lacc[qb][0]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) bad = bad || attn16::nonfinite(o[qb][dt][r]);
    }
    if (__syncthreads_or(bad ? 1 : 0)) {
      // telemetry (ovg_attn_params.fallback_count): one atomic per workgroup that pays the second pass. What the check above certifies is
      // the ARITHMETIC of an accepted speculative pass (no overflowed row sum, nothing non-finite); that the order-pinned instruction stream
      // itself is intact (no compiler copy inside an MFMA hazard window) is certified at BUILD time by build.check_pinned_attention_loops.
      if (p.fallback_count != nullptr && threadIdx.x == 0) atomicAdd(p.fallback_count, 1u);
      attn16::run_tiles<T, QB, WAVES, 0, VSUM, DMA>(p, lds, bh, q0, t0, nt, o, lacc, negm);
    }
  }

  attn16::write_out<T, QB, X3>(p, o, lacc, negm, bh, q0, sp, splits, q_row0, part_rows);
}

// Second launch of a split-KV call: out[entry, q, :] = sum_s w_s part[s][entry, q, :] / sum_s w_s, w_s = 2^(lse_s - max lse)
// (exact: softmax over disjoint key sets combines through the log-sum-exps) for the rows [row0, row0 + nrows) of every entry -- all of
// them after a whole-launch split, the tail rows after a key-split tail launch. One thread = 8 features (two f32x4) of one (entry, query);
// writes the caller's final layout (token-major or head-major) and, if asked, the total log-sum-exp.
template <typename T>
__global__ __launch_bounds__(256) void attn_split_merge_kernel(ovg_attn_params p, int splits, int row0, int nrows, int part_rows) {
  const int64_t total = (int64_t)p.BH * nrows * 8;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c = (int)(idx & 7);
    const int64_t rq = idx >> 3;                       // entry * nrows + local row
    const int bh = (int)(rq / nrows), ql = (int)(rq - (int64_t)bh * nrows), q = row0 + ql;
    const int64_t row = (int64_t)bh * part_rows + ql, stride = p.BH * (int64_t)part_rows;
    float l[OVG_MAX_SEG], m = -INFINITY;
#pragma unroll
    for (int s = 0; s < OVG_MAX_SEG; ++s)
      if (s < splits) { l[s] = p.ws_lse[s * stride + row]; m = fmaxf(m, l[s]); }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float wsum = 0.f;
#pragma unroll
    for (int s = 0; s < OVG_MAX_SEG; ++s)
      if (s < splits) {
        const float w = __builtin_amdgcn_exp2f(l[s] - m);
        wsum += w;
        const float* src = static_cast<const float*>(p.ws_part) + (s * stride + row) * OVG_D + c * 8;
        acc0 += w * *reinterpret_cast<const f32x4*>(src);
        acc1 += w * *reinterpret_cast<const f32x4*>(src + 4);
      }
    const float inv = 1.0f / wsum;
    const int bq = bh / OVG_H, hh = bh % OVG_H;
    T* dst = p.out_bh_stride > 0 ? static_cast<T*>(p.out) + (int64_t)bh * p.out_bh_stride + (int64_t)q * p.ldo + c * 8
                                 : static_cast<T*>(p.out) + ((int64_t)bq * p.nq + q) * p.ldo + hh * OVG_D + c * 8;
    store4<T>(dst, acc0[0] * inv, acc0[1] * inv, acc0[2] * inv, acc0[3] * inv);
    store4<T>(dst + 4, acc1[0] * inv, acc1[1] * inv, acc1[2] * inv, acc1[3] * inv);
    if (p.lse != nullptr && c == 0) p.lse[(int64_t)bh * p.nq_pad + q] = m + __builtin_amdgcn_logf(wsum);
  }
}
