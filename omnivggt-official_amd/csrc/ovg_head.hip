// DPT dense-prediction head on gfx950: the convolutions of
// heads/dpt_head.py as NHWC implicit GEMMs on the MFMA, with the element-wise glue the reference runs
// as separate ATen ops (bias, in-place ReLU of the ResidualConvUnits, residual / skip sums, UV position
// embedding, ConvTranspose2d pixel scatter) folded into the GEMM epilogue; plus LayerNorm(2048),
// align_corners bilinear resize and the 1x1 + activation output stage. All three dtypes: the 16-bit modes and (r03) the f32 parity mode
// on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32 through TT<float>::mma, 32-channel k chunks instead of 64).
#include "ovg_common.h"

namespace {

// ---------------------------------------------------------------------------
// LayerNorm over 2048-wide rows of the aggregator output (dpt_head.py:219): one wave per row,
// 8 x float4 per lane, two-pass statistics in registers.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void head_layernorm_kernel(ovg_head_layernorm_params p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f32x4* wv = reinterpret_cast<const f32x4*>(p.weight);
  const f32x4* bv = reinterpret_cast<const f32x4*>(p.bias);
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < p.rows; row += (int64_t)gridDim.x * 4) {
    const int64_t src = (row / p.p0) * p.p1 + p.row_off + row % p.p0;
    const f32x4* xr = reinterpret_cast<const f32x4*>(p.x + src * p.ldx);
    f32x4 v[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = xr[lane + 64 * i]; s += v[i][0] + v[i][1] + v[i][2] + v[i][3]; }
    const float mean = wave_sum(s) * (1.0f / 2048.0f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4 d = v[i] - mean;
      q += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / 2048.0f) + p.eps);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4 y = (v[i] - mean) * rstd * wv[lane + 64 * i] + bv[lane + 64 * i];
      store4<T>(static_cast<T*>(p.y) + row * p.ldy + 4 * (lane + 64 * i), y[0], y[1], y[2], y[3]);
    }
  }
}

// ---------------------------------------------------------------------------
// Implicit-GEMM convolution. GEMM view: rows m = output pixels (i, oy, ox), columns n = output
// channels, k = (tap, ci). Same 128 x 128 x 128-byte tile, register-staged double barrier loop and
// swizzled LDS as the linear kernels (ovg_gemm.hip); the X tile rows are GATHERED: for k-tile
// (tap, 64-channel chunk) the row of pixel m is x[i, oy*stride+ky-pad, ox*stride+kx-pad, chunk],
// or zeros outside the image (9-bit validity mask per staged row).
// acc[nt][mt] = C[n = n0 + 64 wn + 16 nt + 4g + r][m = m0 + 64 wm + 16 mt + (lane & 15)]
// ---------------------------------------------------------------------------
template <typename T>
OVG_DEV void conv_mainloop(const ovg_conv_params& p, const int M, const int OH, const int OW, const int m0, const int n0,
                           unsigned char* lds, f32x4 (&acc)[4][4]) {
  constexpr int BKB = 128;
  unsigned char* Ws = lds;
  unsigned char* Xs = lds + 128 * BKB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wm = wave & 1;
  const int g = lane >> 4, lr = lane & 15;
  const int ks = p.ksize, pad = ks >> 1, taps = ks * ks;
  const int64_t ktot_b = (int64_t)taps * p.Cin * (int64_t)sizeof(T);   // bytes per weight row
  const int64_t pix_b = p.ldx * (int64_t)sizeof(T);                    // bytes per input pixel

  const unsigned char* xc[4];
  const unsigned char* wg[4];
  unsigned vmask[4];
  int soff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (tid >> 3) + 32 * i, ch = tid & 7;
    wg[i] = static_cast<const unsigned char*>(p.w) + (int64_t)(n0 + row) * ktot_b + ch * 16;
    int m = m0 + row; m = m < M ? m : M - 1;
    const int img = m / (OH * OW), rem = m - img * (OH * OW);
    const int oy = rem / OW, ox = rem - oy * OW;
    const int iy0 = oy * p.stride - pad, ix0 = ox * p.stride - pad;
    xc[i] = static_cast<const unsigned char*>(p.x) + (((int64_t)img * p.H + iy0) * p.W + ix0) * pix_b + ch * 16;
    unsigned vm = 0;
    for (int ky = 0; ky < ks; ++ky)
      for (int kx = 0; kx < ks; ++kx)
        if (iy0 + ky >= 0 && iy0 + ky < p.H && ix0 + kx >= 0 && ix0 + kx < p.W) vm |= 1u << (ky * ks + kx);
    vmask[i] = vm;
    soff[i] = swz_off<128>(row, ch);
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int cpt = p.Cin / (BKB / (int)sizeof(T));             // k-tiles (128 B of channels) per tap
  const int nk = taps * cpt;
  u32x4 rx[4], rw[4];
  int tap = 0, cc = 0;                                        // (tap, chunk) of the k-tile being fetched
  auto fetch = [&](int kt) {
    const int ky = tap / ks, kx = tap - ky * ks;
    const int64_t toff = ((int64_t)ky * p.W + kx) * pix_b + (int64_t)cc * BKB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      rw[i] = *reinterpret_cast<const u32x4*>(wg[i] + (int64_t)kt * BKB);
      if ((vmask[i] >> tap) & 1u) rx[i] = *reinterpret_cast<const u32x4*>(xc[i] + toff);
      else rx[i] = u32x4{0u, 0u, 0u, 0u};
    }
    if (++cc == cpt) { cc = 0; ++tap; }
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<u32x4*>(Xs + soff[i]) = rx[i];
      *reinterpret_cast<u32x4*>(Ws + soff[i]) = rw[i];
    }
  };
  fetch(0);
  stash();
  __syncthreads();

  const int sx = lr >> 1;
  const int wrow = (wn * 64 + lr) * 128, xrow = (wm * 64 + lr) * 128;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + 1) < nk;
    if (more) fetch(kt + 1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int coff = ((kk * 4 + g) ^ sx) << 4;
      u32x4 a[4], b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[t] = *reinterpret_cast<const u32x4*>(Ws + wrow + t * 16 * 128 + coff);
        b[t] = *reinterpret_cast<const u32x4*>(Xs + xrow + t * 16 * 128 + coff);
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) TT<T>::mma(acc[nt][mt], a[nt], b[mt]);
    }
    __syncthreads();
    if (more) {
      stash();
      __syncthreads();
    }
  }
}

template <typename T> OVG_DEV f32x4 load4(const T* src) {
  typedef T vec4 __attribute__((ext_vector_type(4)));
  const vec4 v = *reinterpret_cast<const vec4*>(src);
  return f32x4{static_cast<float>(v[0]), static_cast<float>(v[1]), static_cast<float>(v[2]), static_cast<float>(v[3])};
}

// Epilogue of both conv kernels on a wave's 64 (n) x 16 MT (m) accumulator block: bias, UV position embedding, residual / skip sums, ReLU,
// ConvTranspose pixel scatter. Everything that depends on the column only (output channel, pixel-shuffle offset, bias) is resolved once per
// 16-column block, outside the row loop.
template <typename T, bool OUT_F32, int MT>
OVG_DEV void conv_epilogue(const ovg_conv_params& p, const f32x4 (&acc)[4][MT], const int m_w0, const int n_w0, const int OH, const int OW, const int M) {
  const int lane = threadIdx.x & 63, g = lane >> 4, lr = lane & 15;
  const int s = p.upshuffle > 1 ? p.upshuffle : 1;
  const int half = p.Cout >> 1;
  int co[4], dy[4], dx[4];
  bool live[4];
  f32x4 bias[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int n = n_w0 + nt * 16 + 4 * g;
    co[nt] = n; dy[nt] = 0; dx[nt] = 0;
    live[nt] = true;
    if (s > 1) {
      const int q = n / p.Cout;
      co[nt] = n - q * p.Cout;
      dy[nt] = q / s; dx[nt] = q - dy[nt] * s;
    } else if (n >= p.Cout) {
      live[nt] = false;                                 // zero-padded weight rows (Cout < the tile's columns)
      co[nt] = 0;
    }
    bias[nt] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + co[nt]) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m_w0 + mt * 16 + lr;
    if (m >= M) continue;
    const int img = m / (OH * OW), rem2 = m - img * (OH * OW);
    const int oy = rem2 / OW, ox = rem2 - oy * OW;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      if (!live[nt]) continue;
      const int c = co[nt];
      const int64_t opix = s > 1 ? ((int64_t)img * OH * s + (oy * s + dy[nt])) * ((int64_t)OW * s) + (ox * s + dx[nt]) : (int64_t)m;
      f32x4 v = acc[nt][mt] + bias[nt];
      if (p.pos_x) {
        if (c < half) v += *reinterpret_cast<const f32x4*>(p.pos_x + (int64_t)ox * half + c);
        else v += *reinterpret_cast<const f32x4*>(p.pos_y + (int64_t)oy * half + (c - half));
      }
      if (p.add1) v += load4<T>(static_cast<const T*>(p.add1) + opix * p.ld1 + c);
      if (p.add2) v += load4<T>(static_cast<const T*>(p.add2) + opix * p.ld2 + c);
      if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      if constexpr (OUT_F32) *reinterpret_cast<f32x4*>(static_cast<float*>(p.y) + opix * p.ldy + c) = v;
      else store4<T>(static_cast<T*>(p.y) + opix * p.ldy + c, v[0], v[1], v[2], v[3]);
    }
  }
}

template <typename T, bool OUT_F32>
__global__ __launch_bounds__(256, 2) void conv_kernel(ovg_conv_params p, int OH, int OW, int M, int ntiles_n) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 128 * 128];
  // m-fastest inside groups of 8 m-tiles x all n-tiles (same idea as ovg_gemm.hip tile_coords): the
  // gathered X rows of a group stay in the XCD's L2 while the (small) weight matrix streams
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int mtiles = (M + 127) / 128;
  const int per_group = 8 * ntiles_n;
  const int grp = lid / per_group, rem = lid - grp * per_group;
  const int m_first = grp * 8;
  const int gsz = (mtiles - m_first) < 8 ? (mtiles - m_first) : 8;
  const int tm = m_first + rem % gsz, tn = rem / gsz;
  const int m0 = tm * 128, n0 = tn * 128;

  f32x4 acc[4][4];
  conv_mainloop<T>(p, M, OH, OW, m0, n0, lds, acc);
  const int wave = threadIdx.x >> 6;
  conv_epilogue<T, OUT_F32, 4>(p, acc, m0 + (wave & 1) * 64, n0 + (wave >> 1) * 64, OH, OW, M);
}

#include "ovg_conv256.h"

// 256-pixel tiles on the LDS-DMA ring (16-bit modes, two images within 32-bit byte offsets; ovg_conv256.h): WN = 4 -- 256 GEMM columns per tile,
// 8 waves, one workgroup per CU; WN = 2 -- 128 columns per tile, 4 waves, two workgroups per CU (output_conv1: 128 output channels)
template <typename T, int WN>
__global__ __launch_bounds__(128 * WN, WN == 4 ? 1 : 2) void conv256_kernel(ovg_conv_params p, int OH, int OW, int M, int ntiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_c256[];
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int mtiles = (M + 255) / 256;
  const int per_group = 4 * ntiles_n;                   // m-fastest inside groups of 4 m-tiles x all n-tiles (ovg_gemm.hip TILE_GROUP256)
  const int grp = lid / per_group, rem = lid - grp * per_group;
  const int m_first = grp * 4;
  const int gsz = (mtiles - m_first) < 4 ? (mtiles - m_first) : 4;
  const int tm = m_first + rem % gsz, tn = rem / gsz;
  const int m0 = tm * 256, n0 = tn * c256::Geo<WN>::BN;
  f32x4 acc[4][8];
  c256::mainloop<T, WN>(p, M, OH, OW, m0, n0, lds_c256, acc);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  conv_epilogue<T, false, 8>(p, acc, m0 + (wave / WN) * 128, n0 + (wave & (WN - 1)) * 64, OH, OW, M);
}

// ---------------------------------------------------------------------------
// Bilinear resize, align_corners = True (ATen upsample_bilinear2d semantics: src = dst * (in-1)/(out-1),
// weights (1-l, l), index clamped at in-1), NHWC, one thread = 8 channels of one output pixel.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void upsample_kernel(ovg_upsample_params p, float sy, float sxr, int64_t total) {
  const int c8n = p.C / 8, half = p.C / 2;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c0 = (int)(idx % c8n) * 8;
    const int64_t pix = idx / c8n;
    const int ox = (int)(pix % p.OW);
    const int64_t t = pix / p.OW;
    const int oy = (int)(t % p.OH);
    const int64_t img = t / p.OH;
    const float fy = sy * oy, fx = sxr * ox;
    int y0 = (int)fy, x0 = (int)fx;
    y0 = y0 < p.H - 1 ? y0 : p.H - 1;
    x0 = x0 < p.W - 1 ? x0 : p.W - 1;
    const int y1 = y0 + (y0 < p.H - 1), x1 = x0 + (x0 < p.W - 1);
    const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
    const T* base = static_cast<const T*>(p.x) + img * p.H * p.W * p.ldx + c0;
    typedef T vec8 __attribute__((ext_vector_type(8)));
    const vec8 a = *reinterpret_cast<const vec8*>(base + ((int64_t)y0 * p.W + x0) * p.ldx);
    const vec8 b = *reinterpret_cast<const vec8*>(base + ((int64_t)y0 * p.W + x1) * p.ldx);
    const vec8 c = *reinterpret_cast<const vec8*>(base + ((int64_t)y1 * p.W + x0) * p.ldx);
    const vec8 d = *reinterpret_cast<const vec8*>(base + ((int64_t)y1 * p.W + x1) * p.ldx);
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      o[j] = hy * (hx * static_cast<float>(a[j]) + lx * static_cast<float>(b[j])) + ly * (hx * static_cast<float>(c[j]) + lx * static_cast<float>(d[j]));
    if (p.pos_x) {
      const float* pe = c0 < half ? p.pos_x + (int64_t)ox * half + c0 : p.pos_y + (int64_t)oy * half + (c0 - half);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += pe[j];
    }
    T* dst = static_cast<T*>(p.y) + pix * p.ldy + c0;
    store4<T>(dst, o[0], o[1], o[2], o[3]);
    store4<T>(dst + 4, o[4], o[5], o[6], o[7]);
  }
}

// ---------------------------------------------------------------------------
// conv1x1(32 -> out_dim) + activation (head_act.py:61-125), one thread per pixel
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dpt_out_kernel(ovg_dpt_out_params p) {
  __shared__ float w2s[4 * 32 + 4];
  if (threadIdx.x < p.out_dim * 32) w2s[threadIdx.x] = p.w2[threadIdx.x];
  if (threadIdx.x < p.out_dim) w2s[128 + threadIdx.x] = p.b2[threadIdx.x];
  __syncthreads();
  const int nv = p.out_dim - 1;
  for (int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x; pix < p.npix; pix += (int64_t)gridDim.x * 256) {
    const f32x4* hr = reinterpret_cast<const f32x4*>(p.h + pix * 32);
    float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c4 = 0; c4 < 8; ++c4) {
      const f32x4 h = hr[c4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < p.out_dim) o[j] += w2s[j * 32 + 4 * c4] * h[0] + w2s[j * 32 + 4 * c4 + 1] * h[1] + w2s[j * 32 + 4 * c4 + 2] * h[2] + w2s[j * 32 + 4 * c4 + 3] * h[3];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < p.out_dim) o[j] += w2s[128 + j];
    for (int j = 0; j < nv; ++j) {
      const float v = o[j];
      p.val[pix * nv + j] = p.activation == 0 ? expf(v) : copysignf(expm1f(fabsf(v)), v);
    }
    p.conf[pix] = 1.0f + expf(o[nv]);
  }
}

#include "ovg_dpt_tail.h"

// ---------------------------------------------------------------------------
// depth -> world points (utils/geometry.py:151-266); HBM bound: 4 B read + 12 B written per pixel
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unproject_kernel(ovg_unproject_params p, int64_t total) {
  const int64_t hw = (int64_t)p.H * p.W;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t s = idx / hw, rem = idx - s * hw;
    const int v = (int)(rem / p.W), u = (int)(rem - (int64_t)v * p.W);
    const float* c = p.cam + s * 16;
    const float d = p.depth[idx];
    const float xc = (float)(((double)u - (double)c[14]) * (double)d / (double)c[12]);
    const float yc = (float)(((double)v - (double)c[15]) * (double)d / (double)c[13]);
    // the reference's inverse pose is float64 (np.eye), so np.dot(cam_f32, R^T) + t runs in double
    const double X = xc, Y = yc, Z = d;
    float* o = p.out + idx * 3;
    o[0] = (float)((double)c[0] * X + (double)c[1] * Y + (double)c[2] * Z + (double)c[9]);
    o[1] = (float)((double)c[3] * X + (double)c[4] * Y + (double)c[5] * Z + (double)c[10]);
    o[2] = (float)((double)c[6] * X + (double)c[7] * Y + (double)c[8] * Z + (double)c[11]);
  }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

unsigned grid_1d(int64_t work, int per_block, int cap = 1 << 16) {
  const int64_t b = (work + per_block - 1) / per_block;
  return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" int ovg_head_layernorm(const ovg_head_layernorm_params* p, void* stream) {
  if (!p || !p->x || !p->y || !p->weight || !p->bias || p->rows <= 0 || p->p0 <= 0 || p->p1 < p->p0 || p->row_off < 0) return OVG_E_ARG;
  if (!al16(p->x) || !al16(p->y) || !al16(p->weight) || !al16(p->bias) || (p->ldx % 4) || (p->ldy % 4)) return OVG_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(grid_1d(p->rows, 4)), block(256);
  switch (p->dtype) {
    case OVG_BF16: OVG_LAUNCH((head_layernorm_kernel<bf16_t>), grid, block, 0, st, *p); break;
    case OVG_F16: OVG_LAUNCH((head_layernorm_kernel<f16_t>), grid, block, 0, st, *p); break;
    case OVG_F32: OVG_LAUNCH((head_layernorm_kernel<float>), grid, block, 0, st, *p); break;
    default: return OVG_E_DTYPE;
  }
  OVG_CHECK_LAUNCH();
  return OVG_OK;
}

extern "C" int ovg_conv(const ovg_conv_params* p, void* stream) {
  if (!p || !p->x || !p->w || !p->y) return OVG_E_ARG;
  if (p->n_img <= 0 || p->H <= 0 || p->W <= 0 || p->Cin <= 0 || p->Cout <= 0) return OVG_E_ARG;
  if ((p->ksize != 1 && p->ksize != 3) || (p->stride != 1 && p->stride != 2)) return OVG_E_ARG;
  if (p->dtype != OVG_BF16 && p->dtype != OVG_F16 && p->dtype != OVG_F32) return OVG_E_DTYPE;
  const int kchunk = p->dtype == OVG_F32 ? 32 : 64;           // channels per 128-byte k chunk
  if (p->Cin % kchunk || p->Cout % 4 || p->w_rows <= 0 || p->w_rows % 128) return OVG_E_ARG;
  const int s = p->upshuffle > 1 ? p->upshuffle : 1;
  if (s > 1 && (p->ksize != 1 || p->stride != 1 || p->pos_x || p->add1 || p->add2 || p->w_rows != s * s * p->Cout)) return OVG_E_ARG;
  if (s == 1 && p->w_rows < p->Cout) return OVG_E_ARG;
  if ((p->pos_x == nullptr) != (p->pos_y == nullptr) || (p->pos_x && (p->Cout % 8))) return OVG_E_ARG;
  if (p->ldx < p->Cin || (p->ldx % (p->dtype == OVG_F32 ? 4 : 8)) || p->ldy < p->Cout || (p->ldy % 4)) return OVG_E_ARG;
  if (!al16(p->x) || !al16(p->w) || !al16(p->y) || (p->bias && !al16(p->bias))) return OVG_E_ARG;
  if ((p->add1 && (!al16(p->add1) || (p->ld1 % 4))) || (p->add2 && (!al16(p->add2) || (p->ld2 % 4)))) return OVG_E_ARG;
  const int pad = p->ksize / 2;
  const int OH = (p->H + 2 * pad - p->ksize) / p->stride + 1, OW = (p->W + 2 * pad - p->ksize) / p->stride + 1;
  const int64_t M64 = p->n_img * OH * OW;
  if (M64 <= 0 || M64 > (1 << 30)) return OVG_E_ARG;
  const int M = (int)M64, nt = p->w_rows / 128;
  const dim3 grid((unsigned)(((M + 127) / 128) * nt)), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  // 256-pixel LDS-DMA forms (ovg_conv256.h): 16-bit, 16-bit output, every GEMM column real (w_rows == the GEMM's columns), 32-channel k-stages in
  // pairs (the free-running loop walks two k-stages per iteration), enough pixels for at least one round of the chip, and two images within
  // the 32-bit byte offsets of its descriptor. 256-column tiles where the columns allow, else 128-column tiles at two workgroups per CU.
  const int64_t two_images = 2 * (int64_t)p->H * p->W * p->ldx * 2;
  const int gemm_cols = s * s * p->Cout;
  if (p->dtype != OVG_F32 && !p->out_f32 && p->w_rows == gemm_cols && gemm_cols % 128 == 0 && p->Cin % 32 == 0 && (p->ksize * p->ksize * (p->Cin / 32)) % 2 == 0 &&
      M >= 16384 && two_images < ((int64_t)1 << 32) - 65536 && (int64_t)p->w_rows * p->ksize * p->ksize * p->Cin * 2 < ((int64_t)1 << 32)) {
    // (round 6 lab, profiles/r06_heads_conv_choice_e2e_ab.txt: taking the 37^2 / 19^2 convolutions of short view counts on these kernels too -- from
    // 2048 pixels, with 128-column tiles below 200 workgroups -- changes the 8-view end-to-end time by < 0.5 %, inside the run-to-run spread)
    const bool wide = gemm_cols % 256 == 0;
    const int nt2 = p->w_rows / (wide ? 256 : 128);
    const dim3 grid2((unsigned)(((M + 255) / 256) * nt2));
    // > 64 KB of dynamic LDS is a per-device opt-in of the kernel: done once per (kernel, device), remembered in a bit mask (idempotent; a
    // race between two threads sets it twice)
    static unsigned opted[4] = {0u, 0u, 0u, 0u};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return OVG_E_LAUNCH;
    const int which = (p->dtype == OVG_BF16 ? 0 : 1) + (wide ? 0 : 2);
    const int lds_b = wide ? c256::Geo<4>::LDS_BYTES : c256::Geo<2>::LDS_BYTES;
    if (dev >= 32 || !((opted[which] >> dev) & 1u)) {
      const void* fn = which == 0 ? reinterpret_cast<const void*>(conv256_kernel<bf16_t, 4>) : (which == 1 ? reinterpret_cast<const void*>(conv256_kernel<f16_t, 4>)
                       : (which == 2 ? reinterpret_cast<const void*>(conv256_kernel<bf16_t, 2>) : reinterpret_cast<const void*>(conv256_kernel<f16_t, 2>)));
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b) != hipSuccess) return OVG_E_LAUNCH;
      if (dev < 32) opted[which] |= 1u << dev;
    }
    switch (which) {
      case 0: OVG_LAUNCH((conv256_kernel<bf16_t, 4>), grid2, dim3(512), lds_b, st, *p, OH, OW, M, nt2); break;
      case 1: OVG_LAUNCH((conv256_kernel<f16_t, 4>), grid2, dim3(512), lds_b, st, *p, OH, OW, M, nt2); break;
      case 2: OVG_LAUNCH((conv256_kernel<bf16_t, 2>), grid2, dim3(256), lds_b, st, *p, OH, OW, M, nt2); break;
      default: OVG_LAUNCH((conv256_kernel<f16_t, 2>), grid2, dim3(256), lds_b, st, *p, OH, OW, M, nt2); break;
    }
    OVG_CHECK_LAUNCH();
    return OVG_OK;
  }
  if (p->dtype == OVG_BF16) {
    if (p->out_f32) OVG_LAUNCH((conv_kernel<bf16_t, true>), grid, block, 0, st, *p, OH, OW, M, nt);
    else OVG_LAUNCH((conv_kernel<bf16_t, false>), grid, block, 0, st, *p, OH, OW, M, nt);
  } else if (p->dtype == OVG_F16) {
    if (p->out_f32) OVG_LAUNCH((conv_kernel<f16_t, true>), grid, block, 0, st, *p, OH, OW, M, nt);
    else OVG_LAUNCH((conv_kernel<f16_t, false>), grid, block, 0, st, *p, OH, OW, M, nt);
  } else {
    OVG_LAUNCH((conv_kernel<float, true>), grid, block, 0, st, *p, OH, OW, M, nt);   // f32 activations: y is f32 whatever out_f32 says
  }
  OVG_CHECK_LAUNCH();
  return OVG_OK;
}

extern "C" int ovg_upsample(const ovg_upsample_params* p, void* stream) {
  if (!p || !p->x || !p->y || p->n_img <= 0 || p->H <= 0 || p->W <= 0 || p->OH <= 1 || p->OW <= 1) return OVG_E_ARG;
  if (p->C <= 0 || p->C % 8 || p->ldx < p->C || p->ldy < p->C || (p->ldx % 8) || (p->ldy % 8)) return OVG_E_ARG;
  if ((p->pos_x == nullptr) != (p->pos_y == nullptr) || (p->pos_x && (p->C % 16))) return OVG_E_ARG;
  if (!al16(p->x) || !al16(p->y)) return OVG_E_ARG;
  const int64_t total = p->n_img * p->OH * p->OW * (p->C / 8);
  const float sy = (float)(p->H - 1) / (float)(p->OH - 1), sx = (float)(p->W - 1) / (float)(p->OW - 1);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(grid_1d(total, 256, 1 << 20)), block(256);
  switch (p->dtype) {
    case OVG_BF16: OVG_LAUNCH((upsample_kernel<bf16_t>), grid, block, 0, st, *p, sy, sx, total); break;
    case OVG_F16: OVG_LAUNCH((upsample_kernel<f16_t>), grid, block, 0, st, *p, sy, sx, total); break;
    case OVG_F32: OVG_LAUNCH((upsample_kernel<float>), grid, block, 0, st, *p, sy, sx, total); break;
    default: return OVG_E_DTYPE;
  }
  OVG_CHECK_LAUNCH();
  return OVG_OK;
}

extern "C" int ovg_dpt_out(const ovg_dpt_out_params* p, void* stream) {
  if (!p || !p->h || !p->w2 || !p->b2 || !p->val || !p->conf || p->npix <= 0) return OVG_E_ARG;
  if (p->out_dim < 2 || p->out_dim > 4 || (p->activation != 0 && p->activation != 1) || !al16(p->h)) return OVG_E_ARG;
  OVG_LAUNCH(dpt_out_kernel, dim3(grid_1d(p->npix, 256, 1 << 20)), dim3(256), 0, static_cast<hipStream_t>(stream), *p);
  OVG_CHECK_LAUNCH();
  return OVG_OK;
}

extern "C" int ovg_dpt_tail(const ovg_dpt_tail_params* p, void* stream) {
  if (!p || !p->x || !p->w1 || !p->w2 || !p->b2 || !p->val || !p->conf) return OVG_E_ARG;
  if (p->n_img <= 0 || p->H <= 0 || p->W <= 0 || p->OH <= 1 || p->OW <= 1) return OVG_E_ARG;
  if (p->out_dim < 2 || p->out_dim > 4 || (p->activation != 0 && p->activation != 1)) return OVG_E_ARG;
  if ((p->pos_x == nullptr) != (p->pos_y == nullptr)) return OVG_E_ARG;
  if (p->dtype != OVG_BF16 && p->dtype != OVG_F16) return p->dtype == OVG_F32 || p->dtype == OVG_F16X2 ? OVG_E_UNSUPPORTED : OVG_E_DTYPE;
  if (p->C != dtail::CI) return OVG_E_UNSUPPORTED;
  if ((int64_t)p->H * p->W * p->ldx >= ((int64_t)1 << 31)) return OVG_E_UNSUPPORTED;   // the kernel indexes one source image with 32-bit element offsets
  if (p->ldx < p->C || (p->ldx % 8) || p->ldw1 < 9 * p->C || (p->ldw1 % 8) || !al16(p->x) || !al16(p->w1)) return OVG_E_ARG;
  if ((p->pos_x && (!al16(p->pos_x) || !al16(p->pos_y))) || (p->b1 && !al16(p->b1))) return OVG_E_ARG;
  const int tiles_x = (p->OW + dtail::TW - 1) / dtail::TW, tiles_y = (p->OH + dtail::TH - 1) / dtail::TH;
  const int64_t nt64 = p->n_img * tiles_x * tiles_y;
  if (nt64 > (1 << 30)) return OVG_E_ARG;
  const float sy = (float)(p->H - 1) / (float)(p->OH - 1), sx = (float)(p->W - 1) / (float)(p->OW - 1);   // as ovg_upsample
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return OVG_E_LAUNCH;
  static unsigned opted[2] = {0u, 0u};                       // > 64 KB of dynamic LDS: per-device opt-in, once (as in ovg_conv)
  const int which = p->dtype == OVG_BF16 ? 0 : 1;
  if (dev >= 32 || !((opted[which] >> dev) & 1u)) {
    const void* fn = which == 0 ? reinterpret_cast<const void*>(dtail::dpt_tail_kernel<bf16_t>) : reinterpret_cast<const void*>(dtail::dpt_tail_kernel<f16_t>);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, dtail::LDS_B) != hipSuccess) return OVG_E_LAUNCH;
    if (dev < 32) opted[which] |= 1u << dev;
  }
  const int ntiles = (int)nt64;
  const dim3 grid((unsigned)(ntiles < cus ? ntiles : cus)), block(dtail::NT);   // persistent: one workgroup per CU (156 KB of LDS each)
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (which == 0) OVG_LAUNCH((dtail::dpt_tail_kernel<bf16_t>), grid, block, dtail::LDS_B, st, *p, sy, sx, tiles_x, tiles_y, ntiles);
  else OVG_LAUNCH((dtail::dpt_tail_kernel<f16_t>), grid, block, dtail::LDS_B, st, *p, sy, sx, tiles_x, tiles_y, ntiles);
  OVG_CHECK_LAUNCH();
  return OVG_OK;
}

extern "C" int ovg_unproject(const ovg_unproject_params* p, void* stream) {
  if (!p || !p->depth || !p->cam || !p->out || p->S <= 0 || p->H <= 0 || p->W <= 0) return OVG_E_ARG;
  const int64_t total = p->S * p->H * p->W;
  OVG_LAUNCH(unproject_kernel, dim3(grid_1d(total, 256, 1 << 20)), dim3(256), 0, static_cast<hipStream_t>(stream), *p, total);
  OVG_CHECK_LAUNCH();
  return OVG_OK;
}
