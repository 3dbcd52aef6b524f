// Output stage of the DPT head in ONE kernel (included by ovg_head.hip; entry ovg_dpt_tail):
//   bilinear upsample (align_corners) of the 128-channel map to the image resolution + UV position embedding     dpt_head.py:242-250
//   -> conv3x3(128 -> 32) + ReLU -> conv1x1(32 -> out_dim) -> activation / confidence                            dpt_head.py:252-258, head_act.py:61-125
// The three-launch form (ovg_upsample -> ovg_conv -> ovg_dpt_out) writes the upsampled map (n x 518 x 518 x 128 16-bit values: 550 MB at
// 8 views), reads it nine times through L2 in a GEMM whose 128-column tile is three quarters zero padding (Cout = 32), and round-trips a
// 275 MB f32 map: 0.99 ms + 0.07 ms per head at 8 views. Here a persistent workgroup (one per CU, 8 waves) owns
//   * the conv weights in LDS for its whole life, in MFMA-fragment order (72 KB: fragment (k-step, n-block) = 1 KB, lane l at 16 l),
//   * an 18 x 16 pixel patch of the UPSAMPLED map for a 16 x 14 output tile (72 KB; 518 = 37 x 14, so no tile is cut in x), computed from
//     the four source taps per value straight into LDS -- the upsampled map never exists in HBM. Patch pixel (r, c) keeps its 16-byte
//     chunks XOR-swizzled with (c + 6 r) & 7: the 16 lanes of an MFMA B fragment are 16 consecutive tile pixels (14-wide rows: r steps
//     once inside a fragment, and 6 = 14 mod 8 makes the swizzle class continue across the step), shifted by the tap -- conflict-free.
// Per tile: [phase 1] every thread interpolates 9 (pixel, 8-channel chunk) items whose taps were prefetched into registers while the previous
// tile's MFMA phase ran; [phase 2] wave w < 7 contracts m-blocks {2w, 2w+1} over the 36 k-steps (tap-major, as ovg_conv
// orders them) and finishes in registers: bias + ReLU, the 1x1 conv as 8 FMAs per output and lane + a 4-lane swap reduction, activation, store.
#pragma once

#ifndef OVG_DT_SKIP
#define OVG_DT_SKIP 0      // lab builds (tools/probes/dpt_tail_probe.py): 1 = no MFMA phase, 2 = no interpolation phase, 4 = no tap prefetch
#endif

namespace dtail {

constexpr int TH = 16, TW = 14, PR = TH + 2, PC = TW + 2;
constexpr int CI = 128, CO = 32, NKS = 9 * CI / 32;
constexpr int NT = 512;
constexpr int W_B = CO * 9 * CI * 2;                       // 73 728
constexpr int PATCH_B = PR * PC * CI * 2;                  // 73 728
constexpr int POS_ROWS = PC + PR;                          // 16 column rows (pos_x) then 18 row rows (pos_y), 64 f32 each
constexpr int POS_B = POS_ROWS * 64 * 4;                   // 8 704
constexpr int OUTW_B = (4 * 32 + 32 + 4) * 4;              // w2 [4][32], b1 [32], b2 [4]
constexpr int LDS_B = W_B + PATCH_B + POS_B + OUTW_B;      // 156 816 of 163 840
constexpr int IPT = PR * PC * (CI / 8) / NT;               // 9 items per thread: fixed patch column and chunk, rows r0 + 2 i
static_assert(PR * PC * (CI / 8) == IPT * NT, "items divide evenly");

typedef float f32x2v __attribute__((ext_vector_type(2)));

// 8 values of T in a 16-byte register quad -> four f32 pairs
template <typename T> OVG_DEV void unpack8(const u32x4& v, f32x2v (&o)[4]);
template <> OVG_DEV void unpack8<bf16_t>(const u32x4& v, f32x2v (&o)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = f32x2v{__builtin_bit_cast(float, v[k] << 16), __builtin_bit_cast(float, v[k] & 0xffff0000u)};
}
template <> OVG_DEV void unpack8<f16_t>(const u32x4& v, f32x2v (&o)[4]) {
  f16_t h[8];
  __builtin_memcpy(h, &v, 16);
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = f32x2v{static_cast<float>(h[2 * k]), static_cast<float>(h[2 * k + 1])};
}

template <typename T>
__global__ __launch_bounds__(NT) void dpt_tail_kernel(ovg_dpt_tail_params p, float sy, float sx, int tiles_x, int tiles_y, int ntiles) {
  extern __shared__ __attribute__((aligned(256))) unsigned char lds_dt[];
  unsigned char* w_l = lds_dt;
  float* pos_l = reinterpret_cast<float*>(lds_dt + W_B + PATCH_B);
  float* ow_l = reinterpret_cast<float*>(lds_dt + W_B + PATCH_B + POS_B);
  constexpr uint32_t patch_a = W_B;                                        // byte offset of the patch inside lds_dt (a multiple of 256)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, lr = lane & 15;
  const int half = CI / 2;

  // ---- once per workgroup: conv weights in fragment order, the output-stage constants
  for (int idx = tid; idx < NKS * 2 * 64; idx += NT) {
    const int f = idx >> 6, l = idx & 63;
    const int ks = f >> 1, nt = f & 1;
    const T* src = static_cast<const T*>(p.w1) + (int64_t)(16 * nt + (l & 15)) * p.ldw1 + ks * 32 + (l >> 4) * 8;
    *reinterpret_cast<u32x4*>(w_l + idx * 16) = *reinterpret_cast<const u32x4*>(src);
  }
  if (tid < 128) ow_l[tid] = tid < p.out_dim * 32 ? p.w2[tid] : 0.f;
  if (tid < 32) ow_l[128 + tid] = p.b1 ? p.b1[tid] : 0.f;
  if (tid < 4) ow_l[160 + tid] = tid < p.out_dim ? p.b2[tid] : 0.f;

  // ---- this thread's share of a patch: column pc, chunk ch, rows r0 + 2 i
  const int ch = tid & 15, pc = (tid >> 4) & 15, r0 = tid >> 8;
  const int ldx = (int)p.ldx, row_e = p.W * ldx;                           // elements per source pixel / row (one image < 2^31 elements: checked by the host)
  const int64_t img_e = (int64_t)p.H * row_e;

  u32x4 ta[IPT], tb[IPT], tc[IPT], td[IPT];
  f32x4 pp[2];
  auto decode = [&](int t, int& img, int& oy0, int& ox0) {
    img = t / (tiles_y * tiles_x);
    const int rem = t - img * (tiles_y * tiles_x);
    const int ty = rem / tiles_x;
    oy0 = ty * TH - 1; ox0 = (rem - ty * tiles_x) * TW - 1;                // output coordinates of patch pixel (0, 0)
  };
  auto xgeo = [&](int ox, int& x0, int& x1, float& lx) {
    const int oc = ox < 0 ? 0 : (ox > p.OW - 1 ? p.OW - 1 : ox);
    const float fx = sx * oc;
    x0 = (int)fx; x0 = x0 < p.W - 1 ? x0 : p.W - 1;
    x1 = x0 + (x0 < p.W - 1);
    lx = fx - x0;
  };
  auto ygeo = [&](int oy, int& y0, int& dy, float& ly) {
    const int oc = oy < 0 ? 0 : (oy > p.OH - 1 ? p.OH - 1 : oy);
    const float fy = sy * oc;
    y0 = (int)fy; y0 = y0 < p.H - 1 ? y0 : p.H - 1;
    dy = y0 < p.H - 1;
    ly = fy - y0;
  };
  auto prefetch = [&](int t) {
    int img, oy0, ox0;
    decode(t, img, oy0, ox0);
    int x0, x1; float lx;
    xgeo(ox0 + pc, x0, x1, lx);
    const T* base = static_cast<const T*>(p.x) + img * img_e + ch * 8;
    const int xe0 = x0 * ldx, xe1 = x1 * ldx;
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
      int y0, dy; float ly;
      ygeo(oy0 + r0 + 2 * i, y0, dy, ly);
      const int ra = y0 * row_e, rc = ra + (dy ? row_e : 0);
      ta[i] = *reinterpret_cast<const u32x4*>(base + (ra + xe0));
      tb[i] = *reinterpret_cast<const u32x4*>(base + (ra + xe1));
      tc[i] = *reinterpret_cast<const u32x4*>(base + (rc + xe0));
      td[i] = *reinterpret_cast<const u32x4*>(base + (rc + xe1));
    }
    // position-embedding rows of the tile: entry e = tid (+ 512): row e / 16 of the table (16 columns then 18 rows), 4 floats at (e % 16) * 4
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      int e = tid + NT * k;
      asm volatile("" : "+v"(e));                       // per-tile opaque: keeps the 64-bit table addresses from being hoisted out of the tile loop (and spilled)
      pp[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.pos_x && e < POS_ROWS * 16) {
        const int row = e >> 4, c4 = (e & 15) * 4;
        if (row < PC) {
          const int ox = ox0 + row;
          if (ox >= 0 && ox < p.OW) pp[k] = *reinterpret_cast<const f32x4*>(p.pos_x + (int64_t)ox * half + c4);
        } else {
          const int oy = oy0 + row - PC;
          if (oy >= 0 && oy < p.OH) pp[k] = *reinterpret_cast<const f32x4*>(p.pos_y + (int64_t)oy * half + c4);
        }
      }
    }
  };
  auto stash_pos = [&]() {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = tid + NT * k;
      if (e < POS_ROWS * 16) *reinterpret_cast<f32x4*>(pos_l + e * 4) = pp[k];
    }
  };

  int t = blockIdx.x;
  if (t < ntiles) prefetch(t);
  stash_pos();
  __syncthreads();

  // m-blocks {2 w, 2 w + 1} of waves 0 .. 6 (wave 7 only interpolates) and the per-lane patch origin of their pixels
  int p0[2], f0[2], my[2], mx[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int m = 16 * (2 * (wave < 7 ? wave : 6) + b) + lr;
    my[b] = m / TW; mx[b] = m - my[b] * TW;
    p0[b] = my[b] * PC + mx[b];
    f0[b] = (mx[b] + 6 * my[b]) & 7;
  }
  const uint32_t wa = lane * 16;

  for (; t < ntiles; t += gridDim.x) {
    int img, oy0, ox0;
    decode(t, img, oy0, ox0);
    // ---- phase 1: the upsampled patch (registers -> LDS). out = w00 a + w01 b + w10 c + w11 d + pos on the packed-f32 FMA
#if !(OVG_DT_SKIP & 2)
    {
      int x0, x1; float lx;
      xgeo(ox0 + pc, x0, x1, lx);
      const float hx = 1.f - lx;
      const bool xin = (ox0 + pc) >= 0 && (ox0 + pc) < p.OW;
#pragma unroll
      for (int i = 0; i < IPT; ++i) {
        const int pr = r0 + 2 * i, oy = oy0 + pr;
        int y0, dy; float ly;
        ygeo(oy, y0, dy, ly);
        const float hy = 1.f - ly;
        const float w00 = hy * hx, w01 = hy * lx, w10 = ly * hx, w11 = ly * lx;
        const f32x4* pe = reinterpret_cast<const f32x4*>(pos_l + ((ch >= 8 ? PC + pr : pc) * 64 + (ch & 7) * 8));
        const f32x4 e0 = pe[0], e1 = pe[1];
        f32x2v a[4], b[4], c[4], d[4];
        unpack8<T>(ta[i], a); unpack8<T>(tb[i], b); unpack8<T>(tc[i], c); unpack8<T>(td[i], d);
        const f32x2v e[4] = {f32x2v{e0[0], e0[1]}, f32x2v{e0[2], e0[3]}, f32x2v{e1[0], e1[1]}, f32x2v{e1[2], e1[3]}};
        T vo[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          f32x2v o = __builtin_elementwise_fma(f32x2v{w11, w11}, d[k], e[k]);
          o = __builtin_elementwise_fma(f32x2v{w10, w10}, c[k], o);
          o = __builtin_elementwise_fma(f32x2v{w01, w01}, b[k], o);
          o = __builtin_elementwise_fma(f32x2v{w00, w00}, a[k], o);
          vo[2 * k] = TT<T>::from_f32(o[0]); vo[2 * k + 1] = TT<T>::from_f32(o[1]);
        }
        u32x4 o4;
        __builtin_memcpy(&o4, vo, 16);
        if (!(xin && oy >= 0 && oy < p.OH)) o4 = u32x4{0u, 0u, 0u, 0u};        // the convolution's zero padding
        *reinterpret_cast<u32x4*>(lds_dt + W_B + (pr * PC + pc) * 256 + ((ch ^ ((pc + 6 * pr) & 7)) << 4)) = o4;
      }
    }
#endif
    __syncthreads();
    const int tn = t + gridDim.x;
#if !(OVG_DT_SKIP & 4)
    if (tn < ntiles) prefetch(tn);                                            // in flight under phase 2
#endif

    if (wave < 7 && !(OVG_DT_SKIP & 1)) {
      // ---- phase 2: 3x3 convolution on the matrix pipe. B fragment of (tap, 32-channel group kc), block b: patch pixel p0 + tap shift,
      // chunk (4 kc + g) ^ fs with fs = the pixel's swizzle class = ((g ^ fs) & 3) | ((kc ^ (fs >> 2)) << 2): one XOR per read (byte offset kc << 6)
      int q0[2] = {p0[0], p0[1]}, qf[2] = {f0[0], f0[1]};
      asm volatile("" : "+v"(q0[0]), "+v"(q0[1]), "+v"(qf[0]), "+v"(qf[1]));   // per-tile opaque: keeps the 72 fragment addresses out of registers / scratch
      f32x4 acc[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        uint32_t tbx[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int fs = (qf[b] + kx + 6 * ky) & 7;
          tbx[b] = (patch_a + (q0[b] + ky * PC + kx) * 256 + (((g ^ fs) & 3) << 4)) ^ ((fs & 4) << 4);
        }
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
          const int ks = tap * 4 + kc;
          const u32x4 a0 = *reinterpret_cast<const u32x4*>(lds_dt + (wa + (ks * 2 + 0) * 1024));
          const u32x4 a1 = *reinterpret_cast<const u32x4*>(lds_dt + (wa + (ks * 2 + 1) * 1024));
          const u32x4 b0 = *reinterpret_cast<const u32x4*>(lds_dt + (tbx[0] ^ (kc << 6)));
          const u32x4 b1 = *reinterpret_cast<const u32x4*>(lds_dt + (tbx[1] ^ (kc << 6)));
          TT<T>::mma(acc[0][0], a0, b0);
          TT<T>::mma(acc[1][0], a1, b0);
          TT<T>::mma(acc[0][1], a0, b1);
          TT<T>::mma(acc[1][1], a1, b1);
        }
      }

      // ---- epilogue in registers: bias + ReLU, conv1x1, activation. Lane (lr, g) holds channels 16 nt + 4 g + r of pixel lr of each block.
      const int nv = p.out_dim - 1;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const f32x4 bias = *reinterpret_cast<const f32x4*>(ow_l + 128 + 16 * nt + 4 * g);
          f32x4 v = acc[nt][b] + bias;
          v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(ow_l + d * 32 + 16 * nt + 4 * g);
            part[d] += w[0] * v[0] + w[1] * v[1] + w[2] * v[2] + w[3] * v[3];
          }
        }
        float mine = 0.f;                                                     // output d ends up in lane group g == d
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const float s = quad16_sum(part[d]) + ow_l[160 + d];
          mine = g == d ? s : mine;
        }
        const int oy = oy0 + 1 + my[b], ox = ox0 + 1 + mx[b];
        if (oy < p.OH && ox < p.OW && g < p.out_dim) {
          const int64_t pix = ((int64_t)img * p.OH + oy) * p.OW + ox;
          if (g < nv) p.val[pix * nv + g] = p.activation == 0 ? expf(mine) : copysignf(expm1f(fabsf(mine)), mine);
          else p.conf[pix] = 1.0f + expf(mine);
        }
      }
    }
    stash_pos();                                                              // the next tile's table (its loads were issued before phase 2)
    __syncthreads();
  }
}

}  // namespace dtail
