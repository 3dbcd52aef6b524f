// 256 x 256 ping-pong GEMM main loop for the 16-bit modes (included by ovg_gemm.hip).
//
// Why: the 128 x 128 kernels are bound by the L2 -> LDS path, not by the matrix pipe (ablation
// builds: loads removed -> 1.2-1.3 PFLOP/s, MFMAs removed -> 93 % of the full time): 64 FLOP per
// staged byte with at most 64 KB in flight per CU. This loop doubles the FLOP per staged byte
// (256 x 256 tile), keeps three k-stages (96 KB per CU) of LDS-DMA in flight behind COUNTED
// vmcnt waits (never 0 in steady state), and runs the two waves of every SIMD in antiphase.
//
//   workgroup  512 threads = 8 waves as 4(n) x 2(m); wave tile 64(n) x 128(m) = acc[4][8] (128 VGPRs);
//              1 workgroup per CU (2 waves per SIMD, 256 VGPRs each)
//   k stage    32 elements = 64 B per row: W tile 256 x 64 B + X tile 256 x 64 B = 32 KB; 4-slot ring
//              = 128 KB LDS. With 64-byte rows a 16-row MFMA fragment is 1 KB contiguous in LDS and the
//              DMA image is lane-linear (lane l -> row l/4, slot l%4). ds_read_b128 is serviced in the
//              lane groups {0-3,12-15,20-27}, ... (MI355X_MICROARCH.md LDS table), which makes the plain
//              image 2-way conflicted (PMC: SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE); slot =
//              chunk ^ swz((row >> 2) & 3), swz = {0,2,3,1}, applied to the DMA source chunk and to the
//              fragment reads, is conflict-free for all four groups.
//   ping-pong  group = wave / 4 (the two waves that share a SIMD are in different groups). Per stage
//              a wave runs an L section (12 ds_read_b128 of tile t, 4 global_load_lds of tile t+3) and
//              an M section (32 MFMAs under s_setprio 1). Group 1 is one barrier behind group 0, so
//              between any two consecutive workgroup barriers one group is in M and the other in L:
//              the matrix pipe of every SIMD always has exactly one wave feeding it.
//
//     G0:  P  L(0) b0 M(0)+w(1) b1 L(1) b2 M(1)+w(2) b3 ...
//     G1:  P  b0 L(0)+w(1) b1 M(0) b2 L(1)+w(2) b3 M(1) ...          w(t) = counted vmcnt for tile t
//
//   RAW (DMA -> ds_read): every wave passes its own w(t) before barrier b(2t-1); the first reader of
//        tile t (G0's L(t)) starts after b(2t-1).
//   WAR (ds_read -> DMA into the same slot): tile t+3 reuses the slot of tile t-1 and is issued in L(t),
//        i.e. after b(2t-1); G0's reads of t-1 were consumed by M(t-1) before b(2t-1), G1's L(t-1)
//        ends with lgkmcnt(0) before b(2t-1).
#pragma once

namespace g256 {

constexpr int BM2 = 256, BN2 = 256, ROWB = 64, SLOTS = 4;
constexpr int W_TILE = BN2 * ROWB, X_TILE = BM2 * ROWB, STAGE_B = W_TILE + X_TILE;
constexpr int LDS_BYTES = SLOTS * STAGE_B;

OVG_DEV int swz64(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }   // {0,2,3,1}[(row>>2)&3]

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

OVG_DEV void wait_tiles_in_flight(int n) {     // leave at most n k-stages (4 DMA instructions each) outstanding
  if (n >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (n == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// leaves acc[nt][mt] = C[n = n0 + 64 wn + 16 nt + 4g + r][m = m0 + 128 wm + 16 mt + (lane & 15)]; with SWAP the MFMA
// operands trade places and every 16 x 16 block comes out transposed: acc[nt][mt][r] = C[n = .. + 16 nt + (lane & 15)][m = .. + 16 mt + 4g + r]
// (the V^T tiles of the QKV projection: a lane then owns 4 CONSECUTIVE tokens of one feature = one 8-byte store)
// X3 (OVG_F16X2): operands are (hi, lo) plane pairs and the ring streams 3 nk virtual k-stages -- x_lo * w_hi, x_hi * w_lo, x_hi * w_hi --
// whose source planes are chosen per stage in stage(); everything else (ring, waits, ping-pong) is unchanged.
// DMA_M (round 5): the four LDS-DMA requests of a k-stage are issued INSIDE the M section, one behind every eighth MFMA, instead of in the
// L section. Timeline + issue-cost arithmetic (profiles/r05_gemm_timeline.txt, MI355X_MICROARCH.md "LDS-DMA piece issue cost"): an L section is
// 12 ds_read_b128 + 4 DMA requests at 100-185 cycles each = ~600 cycles, LONGER than the 544 cycles of the partner's 32 MFMAs, so the
// barrier interval was paced by the loads; among MFMAs a request costs ~60 cycles of issue that the matrix pipe covers. The request for
// k-stage t + 3 moves from L(t) to M(t) (later for both groups: the WAR argument above holds a fortiori); group 1's counted wait w(t + 1),
// which sits in L(t) in front of M(t), now sees requests up to k-stage t + 2 only and leaves one stage in flight instead of two.
template <typename T, bool SWAP = false, bool X3 = false, bool DMA_M = false>
OVG_DEV void mainloop(const T* __restrict__ X, int64_t ldx, const T* __restrict__ W, int64_t ldw,
                      int M, int N, int K, int m0, int n0, unsigned char* lds, f32x4 (&acc)[4][8],
                      const T* __restrict__ Xlo = nullptr, const T* __restrict__ Wlo = nullptr) {
  static_assert(sizeof(T) == 2, "16-bit operands");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 3, wm = wave >> 2;
  const int g = lane >> 4, lr = lane & 15;

  // DMA sources: wave w stages rows [32w, 32w + 32) of both tiles, 16 rows per instruction
  const unsigned char* wg[2];
  const unsigned char* xg[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wave * 32 + i * 16 + (lane >> 2);
    int wr = n0 + row; wr = wr < N ? wr : N - 1;
    int xr = m0 + row; xr = xr < M ? xr : M - 1;
    const int ch = (lane & 3) ^ swz64(row);                     // source chunk for linear LDS slot (lane & 3)
    wg[i] = reinterpret_cast<const unsigned char*>(W + (int64_t)wr * ldw) + ch * 16;
    xg[i] = reinterpret_cast<const unsigned char*>(X + (int64_t)xr * ldx) + ch * 16;
  }
  const int nk1 = (K * 2) / ROWB;
  const int64_t dxl = X3 ? reinterpret_cast<const unsigned char*>(Xlo) - reinterpret_cast<const unsigned char*>(X) : 0;
  const int64_t dwl = X3 ? reinterpret_cast<const unsigned char*>(Wlo) - reinterpret_cast<const unsigned char*>(W) : 0;
  auto stage = [&](int kt) {
    unsigned char* wb = lds + (kt & (SLOTS - 1)) * STAGE_B + wave * 32 * ROWB;   // wave-uniform destinations
    unsigned char* xb = wb + W_TILE;
    int64_t xo = (int64_t)kt * ROWB, wo = xo;
    if constexpr (X3) {
      const int pass = kt >= 2 * nk1 ? 2 : (kt >= nk1 ? 1 : 0);
      const int64_t kb = (int64_t)(kt - pass * nk1) * ROWB;
      xo = kb + (pass == 0 ? dxl : 0);
      wo = kb + (pass == 1 ? dwl : 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_global_load_lds((gptr_t)(wg[i] + wo), (lptr_t)(wb + i * 16 * ROWB), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(xg[i] + xo), (lptr_t)(xb + i * 16 * ROWB), 16, 0, 0);
    }
  };
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = X3 ? 3 * nk1 : nk1;
  const int frag_off = lr * ROWB + (g ^ swz64(lr)) * 16;
  const int w_off = wn * 64 * ROWB + frag_off, x_off = W_TILE + wm * 128 * ROWB + frag_off;
  u32x4 a[4], b[8];
  auto read_frags = [&](int kt) {
    const unsigned char* base = lds + (kt & (SLOTS - 1)) * STAGE_B;
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = *reinterpret_cast<const u32x4*>(base + w_off + t * 16 * ROWB);
#pragma unroll
    for (int t = 0; t < 8; ++t) b[t] = *reinterpret_cast<const u32x4*>(base + x_off + t * 16 * ROWB);
  };
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
  auto stage_piece = [&](int kt, int j) {          // DMA_M: request j = 0..3 of k-stage kt (W rows 0-15, X rows 0-15, W rows 16-31, X rows 16-31 of this wave's 32)
    const uint32_t wb = lds_base + (kt & (SLOTS - 1)) * STAGE_B + wave * 32 * ROWB;
    int64_t xo = (int64_t)kt * ROWB, wo = xo;
    if constexpr (X3) {
      const int pass = kt >= 2 * nk1 ? 2 : (kt >= nk1 ? 1 : 0);
      const int64_t kb = (int64_t)(kt - pass * nk1) * ROWB;
      xo = kb + (pass == 0 ? dxl : 0);
      wo = kb + (pass == 1 ? dwl : 0);
    }
    const int i = j >> 1;
    if (j & 1) lds_dma16(xg[i] + xo, wb + W_TILE + i * 16 * ROWB);
    else lds_dma16(wg[i] + wo, wb + i * 16 * ROWB);
  };
  auto mfmas = [&](int kt_req = -1) {              // kt_req >= 0 (DMA_M): k-stage to request between the MFMAs
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        if constexpr (SWAP) TT<T>::mma(acc[nt][mt], b[mt], a[nt]);
        else TT<T>::mma(acc[nt][mt], a[nt], b[mt]);
        if constexpr (DMA_M) {
          if (mt == 3) {
            __builtin_amdgcn_sched_barrier(0);
            if (kt_req >= 0) stage_piece(kt_req, nt);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto in_flight_after = [&](int t) {   // k-stages issued beyond tile t when the wave has staged up to tile min(t + 2, nk - 1)
    const int last = (t + 2) < (nk - 1) ? (t + 2) : (nk - 1);
    return last - t;
  };

  for (int s = 0; s < 3; ++s)
    if (s < nk) stage(s);
  wait_tiles_in_flight(in_flight_after(0));
  __builtin_amdgcn_s_barrier();                      // P: tile 0 visible to every wave (NOT __syncthreads: its fence drains vmcnt to 0)
  tl_mark(1);

  if (wm == 0) {
    for (int t = 0; t < nk; ++t) {
      read_frags(t);                                 // L(t)
      if constexpr (!DMA_M) { if (t + 3 < nk) stage(t + 3); }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();                  // b(2t)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (DMA_M) mfmas(t + 3 < nk ? t + 3 : -1); else mfmas();   // M(t)
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < nk) wait_tiles_in_flight(in_flight_after(t + 1));   // w(t+1): stages issued so far reach t+3
      __builtin_amdgcn_s_barrier();                  // b(2t+1)
    }
    __builtin_amdgcn_s_barrier();                    // pairs with group 1's last barrier
  } else {
    __builtin_amdgcn_s_barrier();                    // b0: one barrier behind group 0
    for (int t = 0; t < nk; ++t) {
      read_frags(t);                                 // L(t)
      if constexpr (!DMA_M) { if (t + 3 < nk) stage(t + 3); }
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < nk) {                              // w(t+1); DMA_M: requests so far reach k-stage t + 2 (M(t - 1))
        if constexpr (DMA_M) wait_tiles_in_flight(((t + 2) < (nk - 1) ? (t + 2) : (nk - 1)) - (t + 1));
        else wait_tiles_in_flight(in_flight_after(t + 1));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                  // b(2t+1)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (DMA_M) mfmas(t + 3 < nk ? t + 3 : -1); else mfmas();   // M(t)
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();                  // b(2t+2)
    }
  }
}


}  // namespace g256
