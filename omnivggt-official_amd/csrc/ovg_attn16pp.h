// attn16pp_kernel: the speculative anchored-softmax attention (ovg_attn16.h, SM = 2) as an 8-wave PING-PONG
// (included by ovg_attn.hip after ovg_attn16.h).
//
// Why: in attn16_kernel the two waves that share a SIMD belong to different workgroups and drift freely, so the
// matrix pipe idles whenever both are in their VALU (exp2 / convert) section: MFMA busy ~52 %. Here one 512-thread
// workgroup owns all 8 waves of a CU (1 workgroup per CU, 2 waves per SIMD); waves 0-3 (group 0) and 4-7 (group 1)
// run the SAME per-tile program half a period apart, separated by workgroup barriers, so that between two
// consecutive barriers one group is in its MFMA section (PV of the previous tile + QK^T of this tile, 72 MFMAs at
// LOW priority) and the other in its VALU section (64 v_exp + 32 v_cvt_pk + staging, s_setprio 2): the VALU wave
// must win the issue arbitration for the two streams to overlap (tools/probes/coexec.hip).
//
//   half-period h:        2t                     2t+1                   2t+2
//   group 0:   M: PV(t-1), QK(t)     | V: exp(t), stage          | M: PV(t), QK(t+1)
//   group 1:   V: exp(t-1), stage    | M: PV(t-1), QK(t)         | V: exp(t), stage
//
// LDS: K tiles and V^T tiles double-buffered separately (4 x 8 KB). During half-periods 2t and 2t+1 the readers
// need K(t) and V(t-1); K(t+1) and V(t) are written in the same two half-periods into the other buffers (their old
// contents K(t-1), V(t-2) were last read in half-period 2t-1) and become visible at the barrier that ends 2t+1.
// Every wave stages 1/8 of each tile in its VALU section: group 0 (VALU section of tile t = half-period 2t+1) writes
// K(t+1), V(t); group 1 (VALU section of tile t = half-period 2t+2) writes K(t+2), V(t+1) -- one tile further ahead.
// The global loads for a stash are issued one VALU section earlier (a full period of latency hiding) and no barrier
// waits on vmcnt: raw s_barrier + s_waitcnt lgkmcnt(0) for the ds_writes only.
// Exactness is the speculative scheme of ovg_attn16.h: per-row anchor from the first key tile, end-of-kernel range
// check, workgroup-uniform recompute with the lazy-rescale body (run_tiles<..., SM = 0>) when it fails.
#pragma once

namespace attn16 {

struct TileIter {            // walks the key tiles of all segments in order
  int seg, tile, ntiles, nk;
  const unsigned char* k;    // segment base of this (batch, head)
  const unsigned char* vt;
  int64_t vstride;
};

OVG_DEV void iter_load_seg(TileIter& it, const ovg_attn_params& p, int bh) {
  if (it.seg < p.nseg) {
    const ovg_kv_segment sg = p.seg[it.seg];
    it.nk = (int)sg.nk;
    it.ntiles = (it.nk + BC - 1) / BC;
    it.k = static_cast<const unsigned char*>(sg.k) + (int64_t)bh * sg.nk_pad * 128;
    it.vt = static_cast<const unsigned char*>(sg.vt) + (int64_t)bh * OVG_D * sg.nk_pad * 2;
    it.vstride = sg.nk_pad * 2;
  }
}
OVG_DEV void iter_init(TileIter& it, const ovg_attn_params& p, int bh) { it.seg = 0; it.tile = 0; iter_load_seg(it, p, bh); }
OVG_DEV void iter_next(TileIter& it, const ovg_attn_params& p, int bh) {
  if (++it.tile == it.ntiles) { it.tile = 0; ++it.seg; iter_load_seg(it, p, bh); }
}
OVG_DEV bool iter_valid(const TileIter& it, const ovg_attn_params& p) { return it.seg < p.nseg; }

template <typename T, int QB>
OVG_DEV void run_tiles_pp(const ovg_attn_params& p, unsigned char* lds, const int bh, const int q0, const int total_tiles,
                          f32x4 (&o)[QB][4], f32x4 (&lacc)[QB]) {
  constexpr int TILE_B = BC * 128;                       // 8 KB: K tile [64 keys][128 B] and V^T tile [64 d][128 B]
  unsigned char* Kb = lds;                               // Kb + (t & 1) * TILE_B
  unsigned char* Vb = lds + 2 * TILE_B;
  const int tid = threadIdx.x, lane = tid & 63;
  const int grp = __builtin_amdgcn_readfirstlane(tid >> 8);       // 0: waves 0-3, 1: waves 4-7
  const int g = lane >> 4, lr = lane & 15;
  const int nq = (int)p.nq;

  u32x4 qf[QB][2];
  {
    const unsigned char* qbase = static_cast<const unsigned char*>(p.q) + (int64_t)bh * p.nq_pad * 128;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      int q = q0 + qb * 16 + lr; q = q < nq ? q : nq - 1;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        qf[qb][kk] = *reinterpret_cast<const u32x4*>(qbase + (int64_t)q * 128 + (4 * kk + g) * 16);
    }
  }
  f32x4 negm[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    negm[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    lacc[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[qb][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const u32x4 ones = OnesFrag<T>::get();

  // staging share of this thread: chunk (row, ch) of both tiles
  const int srow = tid >> 3, sch = tid & 7;
  const int k_goff = tid * 16, k_loff = swz_off<128>(srow, sch);
  const int su = sch >> 2, sc4 = sch & 3;                // key permutation inside each 32-key block (ovg_attn16.h)
  const int v_loff0 = swz_off<128>(srow, 4 * su + 2 * (sc4 & 1) + 0) + 8 * (sc4 >> 1);
  const int v_loff1 = swz_off<128>(srow, 4 * su + 2 * (sc4 & 1) + 1) + 8 * (sc4 >> 1);
  auto load_k = [&](const TileIter& it) { return *reinterpret_cast<const u32x4*>(it.k + (int64_t)it.tile * TILE_B + k_goff); };
  auto load_v = [&](const TileIter& it) { return *reinterpret_cast<const u32x4*>(it.vt + srow * it.vstride + it.tile * 128 + sch * 16); };
  auto stash_k = [&](int t, const u32x4 r) { *reinterpret_cast<u32x4*>(Kb + (t & 1) * TILE_B + k_loff) = r; };
  auto stash_v = [&](int t, const u32x4 r) {
    unsigned char* vl = Vb + (t & 1) * TILE_B;
    *reinterpret_cast<u32x2*>(vl + v_loff0) = u32x2{r[0], r[1]};
    *reinterpret_cast<u32x2*>(vl + v_loff1) = u32x2{r[2], r[3]};
  };

  const int sx = lr >> 1;
  const int frag_row = lr * 128;
  const int coff0 = ((0 + g) ^ sx) << 4, coff1 = ((4 + g) ^ sx) << 4;
  auto qk_tile = [&](const unsigned char* kl, f32x4 (&s)[4][QB]) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const u32x4 k0 = *reinterpret_cast<const u32x4*>(kl + kt * 2048 + frag_row + coff0);
      const u32x4 k1 = *reinterpret_cast<const u32x4*>(kl + kt * 2048 + frag_row + coff1);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        s[kt][qb] = mma_c<T>(k0, qf[qb][0], negm[qb]);
        s[kt][qb] = mma_c<T>(k1, qf[qb][1], s[kt][qb]);
      }
    }
  };
  auto mask_tail = [&](f32x4 (&s)[4][QB], int kv0, int nk) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool dead = (kv0 + 16 * kt + 4 * g + r) >= nk;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) s[kt][qb][r] = dead ? -INFINITY : s[kt][qb][r];
      }
  };
  auto pv_tile = [&](const unsigned char* vl, const u32x4 (&pf)[2][QB]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) lacc[qb] = mma_c<T>(ones, pf[u][qb], lacc[qb]);
      const int voff = ((4 * u + g) ^ sx) << 4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const u32x4 vf = *reinterpret_cast<const u32x4*>(vl + dt * 2048 + frag_row + voff);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) o[qb][dt] = mma_c<T>(vf, pf[u][qb], o[qb][dt]);
      }
    }
  };

  // ---- prologue: K(0) into LDS, per-row anchors, first staging registers ----------------------------------
  TileIter cit;                                          // tile being computed by this wave
  iter_init(cit, p, bh);
  TileIter kit = cit, vit = cit;                         // next tiles to FETCH: K(kit), V(vit)
  stash_k(0, load_k(kit));
  iter_next(kit, p, bh);                                 // kit = tile 1, vit = tile 0
  u32x4 rk = u32x4{0u, 0u, 0u, 0u}, rv = u32x4{0u, 0u, 0u, 0u};
  bool have_k = iter_valid(kit, p), have_v = true;       // registers hold K(1), V(0)
  if (have_k) rk = load_k(kit);
  rv = load_v(vit);
  int st_k = 1, st_v = 0;                                // tile indices the staged registers belong to
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                          // K(0) visible
  {
    f32x4 s[4][QB];
    qk_tile(Kb, s);
    if (BC > cit.nk) mask_tail(s, 0, cit.nk);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      float mx = fmaxf(s[0][qb][0], s[0][qb][1]);
      mx = fmaxf(fmaxf(mx, s[0][qb][2]), s[0][qb][3]);
#pragma unroll
      for (int kt = 1; kt < 4; ++kt) {
        mx = fmaxf(fmaxf(mx, s[kt][qb][0]), s[kt][qb][1]);
        mx = fmaxf(fmaxf(mx, s[kt][qb][2]), s[kt][qb][3]);
      }
      mx = xl_max4(mx) + AnchorMargin<T>::value;
      negm[qb] = f32x4{-mx, -mx, -mx, -mx};
    }
  }
  // one VALU-section worth of staging: write the staged registers, then fetch the next pair
  auto stage_step = [&]() {
    if (have_k) stash_k(st_k, rk);
    if (have_v) stash_v(st_v, rv);
    iter_next(kit, p, bh);
    iter_next(vit, p, bh);
    ++st_k; ++st_v;
    have_k = iter_valid(kit, p);
    have_v = iter_valid(vit, p);
    if (have_k) rk = load_k(kit);
    if (have_v) rv = load_v(vit);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the ds_writes above (NOT the global loads)
  };
  if (grp == 1) {
    stage_step();                                        // group 1 stays one tile further ahead: K(1), V(0) now
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                        // b0: one barrier behind group 0
  }

  // ---- main loop: [M: PV(t-1), QK(t)] barrier [V: exp(t), stage] barrier -------------------------------------
  u32x4 pf[2][QB];
  for (int t = 0; t < total_tiles; ++t) {
    f32x4 s[4][QB];
    __builtin_amdgcn_s_setprio(0);                 // MFMA section at LOW priority (see coexec.hip / ovg_attn16.h)
    if (t > 0) pv_tile(Vb + ((t - 1) & 1) * TILE_B, pf);
    qk_tile(Kb + (t & 1) * TILE_B, s);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(2);                 // VALU section wins the issue arbitration
    const int kv0 = cit.tile * BC;
    if (kv0 + BC > cit.nk) mask_tail(s, kv0, cit.nk);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[2 * u + h][qb][r] = __builtin_amdgcn_exp2f(s[2 * u + h][qb][r]);
        pf[u][qb] = pack2<T>(s[2 * u][qb], s[2 * u + 1][qb]);
      }
    stage_step();
    iter_next(cit, p, bh);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  __builtin_amdgcn_s_setprio(0);
  pv_tile(Vb + ((total_tiles - 1) & 1) * TILE_B, pf);
  if (grp == 0) __builtin_amdgcn_s_barrier();            // pairs with group 1's b0
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace attn16

// 8 waves x 64 q rows: BQ = 512 query rows per workgroup. FORCE (tests): take the recompute path regardless.
template <typename T, bool FORCE>
__global__ __launch_bounds__(512, 1) void attn16pp_kernel(ovg_attn_params p, int nqt, int total_tiles) {
  static_assert(sizeof(T) == 2, "16-bit types only");
  constexpr int QB = 4, WAVES = 8, BQ = 16 * QB * WAVES;
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * BC * 128];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, lr = lane & 15;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = lid / nqt, qt = lid % nqt;
  const int nq = (int)p.nq;
  const int q0 = qt * BQ + wave * 16 * QB;

  f32x4 o[QB][4], lacc[QB];
  attn16::run_tiles_pp<T, QB>(p, lds, bh, q0, total_tiles, o, lacc);
  bool bad = FORCE;
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    bad = bad || attn16::bad_sum(lacc[qb][0]);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) bad = bad || attn16::nonfinite(o[qb][dt][r]);
  }
  if (__syncthreads_or(bad ? 1 : 0)) attn16::run_tiles<T, QB, WAVES, 0>(p, lds, bh, q0, total_tiles, o, lacc);

  const int bq = bh / OVG_H, hh = bh % OVG_H;
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float inv = 1.0f / lacc[qb][0];
    const int q = q0 + qb * 16 + lr;
    if (q < nq) {
      T* dst = static_cast<T*>(p.out) + ((int64_t)bq * nq + q) * p.ldo + hh * OVG_D + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        store4<T>(dst + 16 * dt, o[qb][dt][0] * inv, o[qb][dt][1] * inv, o[qb][dt][2] * inv, o[qb][dt][3] * inv);
    }
  }
}
