// 256 x 256 implicit-GEMM convolution main loop for the 16-bit modes (included by ovg_head.hip): the LDS-DMA ring / counted-vmcnt /
// ping-pong loop of ovg_gemm256.h (same ring geometry, swizzle, barrier schedule and RAW / WAR argument) with GATHERED activation rows.
//
// Round 5 (round-4 review item 3): the 128 x 128 conv kernel ran ~500 TFLOP/s, register-staged with a per-row validity branch in front of
// every load; the DPT heads cost 84 ms of an 882 ms 64-view forward and 12 of 53 ms at 8 views. Here a k-stage is 32 channels of one
// filter tap: W rows come through one buffer descriptor over the weight matrix (SGPR offset = k-stage), X rows through a descriptor
// whose base is the first image the tile touches -- lane offset = the pixel's own offset + the tap's (scalar) offset, and a tap that
// falls outside the image gets an OUT-OF-RANGE lane offset, so the DMA engine deposits the zero padding itself: no zero page, no branch,
// no per-lane pointer arithmetic beyond one select.
#pragma once

namespace c256 {

constexpr int BM2 = 256, BN2 = 256, ROWB = 64, SLOTS = 4;
constexpr int W_TILE = BN2 * ROWB, X_TILE = BM2 * ROWB, STAGE_B = W_TILE + X_TILE;
constexpr int LDS_BYTES = SLOTS * STAGE_B;

OVG_DEV int swz64(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }   // {0,2,3,1}[(row>>2)&3] (ovg_gemm256.h)

OVG_DEV void wait_tiles_in_flight(int n) {     // leave at most n k-stages (4 DMA instructions each) outstanding
  if (n >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (n == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// acc[nt][mt] = C[n = n0 + 64 wn + 16 nt + 4g + r][m = m0 + 128 wm + 16 mt + (lane & 15)], m = output pixel (img, oy, ox) of the launch
template <typename T>
OVG_DEV void mainloop(const ovg_conv_params& p, const int M, const int OH, const int OW, const int m0, const int n0,
                      unsigned char* lds, f32x4 (&acc)[4][8]) {
  static_assert(sizeof(T) == 2, "16-bit operands");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 3, wm = wave >> 2;
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

  uint32_t voff_w[2], xrel[2];
  unsigned vmask[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wave * 32 + i * 16 + (lane >> 2);
    const int ch = (lane & 3) ^ swz64(row);                          // source chunk for linear LDS slot (lane & 3)
    voff_w[i] = (uint32_t)(row * ktot_b + ch * 16);
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
  const int nk = taps * cpt;
  const uint32_t wsoff = (uint32_t)n0 * (uint32_t)ktot_b;
  int s_tap = 0, s_cc = 0;                                            // (tap, channel chunk) of the NEXT k-stage to request; stages are requested in order
  auto stage = [&](int kt) {
    const uint32_t wb = lds_base + (kt & (SLOTS - 1)) * STAGE_B + wave * 32 * ROWB;   // wave-uniform destinations
    const uint32_t xb = wb + W_TILE;
    const int ky = s_tap / ks, kx = s_tap - ky * ks;
    const uint32_t toff = (uint32_t)((ky * p.W + kx) * pix_b + s_cc * ROWB);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      buffer_dma16(wb + i * 16 * ROWB, voff_w[i], srd_w, wsoff + (uint32_t)kt * ROWB);
      const uint32_t xo = ((vmask[i] >> s_tap) & 1u) ? xrel[i] + toff : 0xfffffff0u;   // outside the image: out of range -> zeros
      buffer_dma16(xb + i * 16 * ROWB, xo, srd_x, 0u);
    }
    if (++s_cc == cpt) { s_cc = 0; ++s_tap; }
  };
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

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
  auto mfmas = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) TT<T>::mma(acc[nt][mt], a[nt], b[mt]);
    __builtin_amdgcn_s_setprio(0);
  };
  auto in_flight_after = [&](int t) {
    const int last = (t + 2) < (nk - 1) ? (t + 2) : (nk - 1);
    return last - t;
  };

  for (int s = 0; s < 3; ++s)
    if (s < nk) stage(s);
  wait_tiles_in_flight(in_flight_after(0));
  __builtin_amdgcn_s_barrier();                      // P

  if (wm == 0) {
    for (int t = 0; t < nk; ++t) {
      read_frags(t);                                 // L(t)
      if (t + 3 < nk) stage(t + 3);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();                  // b(2t)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      mfmas();                                       // M(t)
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < nk) wait_tiles_in_flight(in_flight_after(t + 1));
      __builtin_amdgcn_s_barrier();                  // b(2t+1)
    }
    __builtin_amdgcn_s_barrier();                    // pairs with group 1's last barrier
  } else {
    __builtin_amdgcn_s_barrier();                    // b0: one barrier behind group 0
    for (int t = 0; t < nk; ++t) {
      read_frags(t);                                 // L(t)
      if (t + 3 < nk) stage(t + 3);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < nk) wait_tiles_in_flight(in_flight_after(t + 1));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                  // b(2t+1)
      __builtin_amdgcn_sched_barrier(0);
      mfmas();                                       // M(t)
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();                  // b(2t+2)
    }
  }
}

}  // namespace c256
