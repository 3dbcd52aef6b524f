// Implicit-GEMM convolution main loops for the 16-bit modes on 256-pixel tiles (included by ovg_head.hip): the LDS-DMA ring / counted-vmcnt /
// free-running loop of ovg_gemm256.h (same ring geometry, swizzle and RAW / WAR argument) with GATHERED activation rows.
//
// Round 5 (round-4 review item 3): the 128 x 128 conv kernel ran ~500 TFLOP/s, register-staged with a per-row validity branch in front of
// every load; the DPT heads cost 84 ms of an 882 ms 64-view forward and 12 of 53 ms at 8 views. Here a k-stage is 32 channels of one
// filter tap: W rows come through one buffer descriptor over the weight matrix (SGPR offset = k-stage), X rows through a descriptor
// whose base is the first image the tile touches -- lane offset = the pixel's own offset + the tap's (scalar) offset, and a tap that
// falls outside the image gets an OUT-OF-RANGE lane offset, so the DMA engine deposits the zero padding itself: no zero page, no branch,
// no per-lane pointer arithmetic beyond one select.
//
// Round 6: (i) the ping-pong schedule became the free-running one of ovg_gemm256.h (every wave interleaves the fragment reads of k-stage
// t + 1 and the LDS-DMA requests of a later stage with the MFMAs of stage t; one barrier per k-stage); (ii) a second geometry for
// convolutions with 128 GEMM columns (output_conv1 of the DPT head: 3 x 3, 256 -> 128 channels at 296 x 296 pixels per frame -- on the
// 128 x 128 kernel 787 TFLOP/s and 10.6 % of a head's time at 8 views, on the 256-column tile half of every MFMA would be zero padding):
//
//   WN = 4:  256 (pixels) x 256 (columns), 8 waves as 4(n) x 2(m), 4-slot ring of 32 KB stages, one workgroup per CU
//   WN = 2:  256 (pixels) x 128 (columns), 4 waves as 2(n) x 2(m), 3-slot ring of 24 KB stages (72 KB), TWO workgroups per CU
//
// both with a 64(n) x 128(m) accumulator block per wave (acc[4][8]), i.e. the same epilogue.
#pragma once

namespace c256 {

constexpr int BM2 = 256, ROWB = 64;
constexpr int X_TILE = BM2 * ROWB;
template <int WN> struct Geo {
  static constexpr int BN = 64 * WN, WAVES = 2 * WN, SLOTS = WN == 4 ? 4 : 3;
  static constexpr int W_TILE = BN * ROWB, STAGE_B = W_TILE + X_TILE, LDS_BYTES = SLOTS * STAGE_B;
  static constexpr int NPX = BM2 / 16 / WAVES, NP = 2 + NPX;            // LDS-DMA requests per wave and k-stage: 2 x 16 rows of W, NPX x 16 rows of X
};
constexpr int BN2 = Geo<4>::BN, LDS_BYTES = Geo<4>::LDS_BYTES;

OVG_DEV int swz64(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }   // {0,2,3,1}[(row>>2)&3] (ovg_gemm256.h)

template <int NP> OVG_DEV void wait_stages_in_flight(int n) {     // leave at most n k-stages (NP requests each) of this wave outstanding
  if constexpr (NP == 4) {
    if (n >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (n == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    static_assert(NP == 6, "4 or 6 requests per stage");
    if (n >= 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

// acc[nt][mt] = C[n = n0 + 64 wn + 16 nt + 4g + r][m = m0 + 128 wm + 16 mt + (lane & 15)], m = output pixel (img, oy, ox) of the launch.
// Returns behind a workgroup barrier with every DMA waited for and every fragment read retired.
template <typename T, int WN>
OVG_DEV void mainloop(const ovg_conv_params& p, const int M, const int OH, const int OW, const int m0, const int n0,
                      unsigned char* lds, f32x4 (&acc)[4][8]) {
  static_assert(sizeof(T) == 2, "16-bit operands");
  using G = Geo<WN>;
  constexpr int S = G::SLOTS, NP = G::NP, NPX = G::NPX, SB = G::STAGE_B, W_TILE = G::W_TILE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & (WN - 1), wm = wave / WN;
  const int g = lane >> 4, lr = lane & 15;
  const int ks = p.ksize, pad = ks >> 1, taps = ks * ks;
  const int ktot_b = taps * p.Cin * 2;                               // bytes per weight row
  const int pix_b = (int)p.ldx * 2;                                  // bytes per input pixel
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);

  // descriptors: weights (whole padded matrix); activations from the first image this tile touches to the end of the tensor (< 4 GB:
  // the host sends a shape here only when two images fit 32-bit offsets -- a 256-pixel tile spans at most two)
  const i32x4 srd_w = make_srd(p.w, (uint32_t)p.w_rows * (uint32_t)ktot_b);
  const int img0 = m0 / (OH * OW);
  const int64_t img_b = (int64_t)p.H * p.W * pix_b;
  const int64_t rest = (p.n_img - img0) * img_b;
  const i32x4 srd_x = make_srd(static_cast<const unsigned char*>(p.x) + (int64_t)img0 * img_b, (uint32_t)(rest < 0xfffffff0ll ? rest : 0xfffffff0ll));

  uint32_t voff_w[2], xrel[NPX];
  unsigned vmask[NPX];
#pragma unroll
  for (int i = 0; i < 2; ++i) {                                       // wave w stages W rows [32 w, 32 w + 32)
    const int row = wave * 32 + i * 16 + (lane >> 2);
    voff_w[i] = (uint32_t)(row * ktot_b + (((lane & 3) ^ swz64(row)) * 16));   // source chunk for linear LDS slot (lane & 3)
  }
#pragma unroll
  for (int i = 0; i < NPX; ++i) {                                     // ... and X rows [16 NPX w, 16 NPX (w + 1))
    const int row = wave * 16 * NPX + i * 16 + (lane >> 2);
    const int ch = (lane & 3) ^ swz64(row);
    const int m = m0 + row;
    const int mc = m < M ? m : M - 1;
    const int img = mc / (OH * OW), rem = mc - img * (OH * OW);
    const int oy = rem / OW, ox = rem - oy * OW;
    const int iy0 = oy * p.stride - pad, ix0 = ox * p.stride - pad;
    // byte offset of tap (0, 0) of this pixel from the descriptor base; may be "negative" (wraps) only where tap (0, 0) is itself outside
    xrel[i] = (uint32_t)((int64_t)(img - img0) * img_b + ((int64_t)iy0 * p.W + ix0) * pix_b + ch * 16);
    unsigned vm = 0;
    if (m < M)
      for (int ky = 0; ky < ks; ++ky)
        for (int kx = 0; kx < ks; ++kx)
          if (iy0 + ky >= 0 && iy0 + ky < p.H && ix0 + kx >= 0 && ix0 + kx < p.W) vm |= 1u << (ky * ks + kx);
    vmask[i] = vm;
  }
  const int cpt = (p.Cin * 2) / ROWB;                                // k-stages per tap
  const int nk = taps * cpt;                                         // even: Cin % 64 == 0 (the host checks it)
  const uint32_t wsoff = (uint32_t)n0 * (uint32_t)ktot_b;
  int s_tap = 0, s_cc = 0;                                            // (tap, channel chunk) of the k-stage being requested; stages are requested in order, piece by piece
  auto stage_piece = [&](int kt, int j) {                            // request j of k-stage kt: the wave's two W pieces, then its NPX X pieces
    const uint32_t sb = lds_base + (S == 4 ? (kt & 3) : kt % 3) * SB;
    if (j < 2) {
      buffer_dma16(sb + (wave * 2 + j) * 16 * ROWB, voff_w[j], srd_w, wsoff + (uint32_t)kt * ROWB);
    } else {
      const int i = j - 2;
      const int ky = s_tap / ks, kx = s_tap - ky * ks;
      const uint32_t toff = (uint32_t)((ky * p.W + kx) * pix_b + s_cc * ROWB);
      const uint32_t xo = ((vmask[i] >> s_tap) & 1u) ? xrel[i] + toff : 0xfffffff0u;   // outside the image: out of range -> zeros
      buffer_dma16(sb + W_TILE + (wave * NPX + i) * 16 * ROWB, xo, srd_x, 0u);
      if (i == NPX - 1 && ++s_cc == cpt) { s_cc = 0; ++s_tap; }
    }
  };
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frag_off = lr * ROWB + (g ^ swz64(lr)) * 16;
  const unsigned char* wfrag = lds + wn * 64 * ROWB + frag_off;
  const unsigned char* xfrag = lds + W_TILE + wm * 128 * ROWB + frag_off;
  u32x4 a0[4], a1[4], b[8];

  for (int s = 0; s < S; ++s)
    if (s < nk)
      for (int j = 0; j < NP; ++j) stage_piece(s, j);
  wait_stages_in_flight<NP>((nk < S ? nk : S) - 1);           // stage 0 landed, the others stay in flight
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int t = 0; t < 4; ++t) a0[t] = *reinterpret_cast<const u32x4*>(wfrag + t * 16 * ROWB);
#pragma unroll
  for (int t = 0; t < 8; ++t) b[t] = *reinterpret_cast<const u32x4*>(xfrag + t * 16 * ROWB);

  // one k-stage (ovg_gemm256.h): MFMAs of stage t on (acur, b), fragment reads of stage t + 1 into (anxt, b), DMA requests of stage t + S
  auto body = [&](int t, const u32x4 (&acur)[4], u32x4 (&anxt)[4], bool dma, int infl) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (infl >= 0) wait_stages_in_flight<NP>(infl);
    __builtin_amdgcn_s_barrier();                              // B(t)
    __builtin_amdgcn_sched_barrier(0);
    const int so = (S == 4 ? ((t + 1) & 3) : (t + 1) % 3) * SB;
    const unsigned char* wn_ = wfrag + so;
    const unsigned char* xn_ = xfrag + so;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) TT<T>::mma(acc[nt][mt], acur[nt], b[mt]);
      if (mt < 4) anxt[mt] = *reinterpret_cast<const u32x4*>(wn_ + mt * 16 * ROWB);
      b[mt] = *reinterpret_cast<const u32x4*>(xn_ + mt * 16 * ROWB);
      __builtin_amdgcn_sched_barrier(0);
      constexpr int kPiece4[8] = {-1, 0, -1, 1, -1, 2, -1, 3}, kPiece6[8] = {-1, 0, 1, 2, -1, 3, 4, 5};
      const int pj = NP == 4 ? kPiece4[mt] : kPiece6[mt];
      if (dma && pj >= 0) { stage_piece(t + S, pj); __builtin_amdgcn_sched_barrier(0); }
    }
  };

  int t = 0;
  for (; t + S + 1 < nk; t += 2) {                             // steady state
    body(t, a0, a1, true, S - 2);
    body(t + 1, a1, a0, true, S - 2);
  }
  for (; t + 1 < nk; t += 2) {                                 // drain: the last requests (3-slot ring), then none
    const int i0 = nk - 2 - t, i1 = nk - 3 - t;
    body(t, a0, a1, t + S < nk, i0 < S - 2 ? i0 : S - 2);
    body(t + 1, a1, a0, t + 1 + S < nk, i1 < 0 ? -1 : (i1 < S - 2 ? i1 : S - 2));
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

}  // namespace c256
