// HBM-bound kernels of the aggregator path (gfx950): LayerNorm-1024, patch im2col,
// masked depth statistics, DINOv2 special rows, AA-trunk token assembly, row copies,
// and the MFMA lane-map probe used by tools/selftest.py.
// One wave (64 lanes) owns one 1024-wide row: 4 x float4 per lane, fully coalesced.
#include "ovg_common.h"

namespace {

// row statistics of a 1024-wide f32 row held as v[4] float4 per lane (two-pass, like
// ATen's native_layer_norm: mean, then biased variance of the centred values)
OVG_DEV void row_stats(const f32x4 (&v)[4], float& mean, float& rstd, float eps) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  mean = wave_sum(s) * (1.0f / 1024.0f);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float d = v[i][r] - mean; q += d * d; }
  rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / 1024.0f) + eps);
}

template <typename T, bool OUT_F32, bool X3 = false>   // X3 (OVG_F16X2): T = f16_t, the row goes out as two f16 planes y (hi) / y_lo
__global__ __launch_bounds__(256) void layernorm_kernel(ovg_layernorm_params p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f32x4* wv = reinterpret_cast<const f32x4*>(p.weight);
  const f32x4* bv = reinterpret_cast<const f32x4*>(p.bias);
  f32x4 w[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { w[i] = wv[lane + 64 * i]; b[i] = bv[lane + 64 * i]; }
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < p.rows; row += (int64_t)gridDim.x * 4) {
    const f32x4* xr = reinterpret_cast<const f32x4*>(p.x + row * p.ldx);
    f32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = xr[lane + 64 * i];
    float mean, rstd;
    row_stats(v, mean, rstd, p.eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 y = (v[i] - mean) * rstd * w[i] + b[i];
      const int col = 4 * (lane + 64 * i);
      if constexpr (OUT_F32) *reinterpret_cast<f32x4*>(static_cast<float*>(p.y) + row * p.ldy + col) = y;
      else if constexpr (X3) store4_hilo(static_cast<f16_t*>(p.y) + row * p.ldy + col, static_cast<f16_t*>(p.y_lo) + row * p.ldy + col, y[0], y[1], y[2], y[3]);
      else store4<T>(static_cast<T*>(p.y) + row * p.ldy + col, y[0], y[1], y[2], y[3]);
    }
  }
}

// --------------------------------------------------------------------------
// im2col for the k=14,s=14 patch convolutions.  One thread = 8 consecutive k of
// one patch row (16 B of bf16/f16 output, 32 B of f32).
// --------------------------------------------------------------------------
template <typename T, bool X3 = false>
__global__ __launch_bounds__(256) void im2col_kernel(ovg_im2col_params p, int gh, int gw, int64_t total) {
  const int kc = (int)(p.k_pad / 8);
  const int kvalid = p.C * 196;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(idx % kc);
    const int64_t prow = idx / kc;
    const int pp = (int)(prow % (gh * gw));
    const int64_t v = prow / (gh * gw);
    const int py = pp / gw, px = pp % gw;
    float o[8];
    float inv_den = 0.f; bool has = true; float den = 1.f;
    if (p.mode == 1) {
      const int64_t b = v / p.views_per_batch;
      const double sum = p.depth_stats[2 * b], cnt = p.depth_stats[2 * b + 1];
      has = cnt > 0.0;
      den = has ? (static_cast<float>(sum / cnt) + 1e-8f) : 1.f;
      (void)inv_den;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = c8 * 8 + e;
      float val = 0.f;
      if (k < kvalid) {
        const int c = k / 196, rem = k % 196, ky = rem / 14, kx = rem % 14;
        const int64_t pix = (int64_t)(py * 14 + ky) * p.Wpx + (px * 14 + kx);
        if (p.mode == 0) {
          const float x = p.img[(v * p.C + c) * (int64_t)p.Hpx * p.Wpx + pix];
          val = (x - p.mean[c]) / p.std[c];
        } else {
          const int64_t off = v * (int64_t)p.Hpx * p.Wpx + pix;
          const float mk = p.img2[off];
          if (c == 0) val = has ? (p.img[off] / den) * mk : 0.f;
          else val = mk;
        }
      }
      o[e] = val;
    }
    T* dst = static_cast<T*>(p.out) + prow * p.k_pad + c8 * 8;
    if constexpr (X3) {
      f16_t* dlo = static_cast<f16_t*>(p.out_lo) + prow * p.k_pad + c8 * 8;
      store4_hilo(dst, dlo, o[0], o[1], o[2], o[3]);
      store4_hilo(dst + 4, dlo + 4, o[4], o[5], o[6], o[7]);
    } else {
      store4<T>(dst, o[0], o[1], o[2], o[3]);
      store4<T>(dst + 4, o[4], o[5], o[6], o[7]);
    }
  }
}

// masked sum / count, two deterministic stages (double accumulation)
__global__ __launch_bounds__(256) void depth_stats_stage1(ovg_depth_stats_params p) {
  __shared__ double ss[4], sc[4];
  const int b = blockIdx.y;
  const float* d = p.depth + (int64_t)b * p.n_per_batch;
  const float* m = p.mask + (int64_t)b * p.n_per_batch;
  double s = 0.0, c = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.n_per_batch; i += (int64_t)gridDim.x * 256) {
    if (m[i] > 0.f) { s += (double)d[i]; c += 1.0; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); c += __shfl_xor(c, o, 64); }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { ss[wave] = s; sc[wave] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* out = p.partial + ((int64_t)b * gridDim.x + blockIdx.x) * 2;
    out[0] = (ss[0] + ss[1]) + (ss[2] + ss[3]);
    out[1] = (sc[0] + sc[1]) + (sc[2] + sc[3]);
  }
}
__global__ __launch_bounds__(64) void depth_stats_stage2(ovg_depth_stats_params p) {
  const int b = blockIdx.x;
  double s = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < p.nblocks; i += 64) {
    s += p.partial[((int64_t)b * p.nblocks + i) * 2];
    c += p.partial[((int64_t)b * p.nblocks + i) * 2 + 1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); c += __shfl_xor(c, o, 64); }
  if (threadIdx.x == 0) { p.stats[2 * b] = s; p.stats[2 * b + 1] = c; }
}

__global__ __launch_bounds__(256) void dino_specials_kernel(ovg_dino_specials_params p) {
  // one block per (view, special row); 256 threads x float4 = 1024
  const int64_t v = blockIdx.x / (1 + p.n_reg);
  const int t = blockIdx.x % (1 + p.n_reg);
  const int c = threadIdx.x * 4;
  f32x4 val;
  if (t == 0) val = *reinterpret_cast<const f32x4*>(p.cls + c) + *reinterpret_cast<const f32x4*>(p.pos0 + c);
  else val = *reinterpret_cast<const f32x4*>(p.reg + (int64_t)(t - 1) * OVG_C + c);
  *reinterpret_cast<f32x4*>(p.x + (v * p.tokens_per_view + t) * p.ldx + c) = val;
}

__global__ __launch_bounds__(256) void assemble_kernel(ovg_assemble_params p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t rows = p.V * p.tokens_per_view;
  const int64_t p0 = p.tokens_per_view - p.n_special;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const int64_t v = row / p.tokens_per_view;
    const int t = (int)(row % p.tokens_per_view);
    const int slot = ((p.view0 + v) % p.S) == 0 ? 0 : 1;
    float* dst = p.out + row * p.ldo;
    if (t == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int col = 4 * (lane + 64 * i);
        *reinterpret_cast<f32x4*>(dst + col) = *reinterpret_cast<const f32x4*>(p.camera_token + slot * OVG_C + col) +
                                               *reinterpret_cast<const f32x4*>(p.cam_add + v * OVG_C + col);
      }
    } else if (t < p.n_special) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int col = 4 * (lane + 64 * i);
        *reinterpret_cast<f32x4*>(dst + col) =
            *reinterpret_cast<const f32x4*>(p.register_token + ((int64_t)slot * (p.n_special - 1) + (t - 1)) * OVG_C + col);
      }
    } else {
      const f32x4* xr = reinterpret_cast<const f32x4*>(p.xd + row * p.ldxd);
      f32x4 x[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) x[i] = xr[lane + 64 * i];
      float mean, rstd;
      row_stats(x, mean, rstd, p.eps);
      const int dr = p.depth_row ? p.depth_row[v] : -1;
      const float* add = (dr >= 0 && p.depth_tok) ? p.depth_tok + ((int64_t)dr * p0 + (t - p.n_special)) * OVG_C : p.placeholder;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int col = 4 * (lane + 64 * i);
        const f32x4 w = *reinterpret_cast<const f32x4*>(p.norm_w + col);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.norm_b + col);
        const f32x4 a = *reinterpret_cast<const f32x4*>(add + col);
        *reinterpret_cast<f32x4*>(dst + col) = ((x[i] - mean) * rstd * w + b) + a;
      }
    }
  }
}

__global__ __launch_bounds__(256) void copy_rows_kernel(ovg_copy_rows_params p) {
  const int64_t n4 = p.n / 4, total = p.rows * n4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / n4, c = (i % n4) * 4;
    *reinterpret_cast<f32x4*>(p.y + r * p.ldy + c) = *reinterpret_cast<const f32x4*>(p.x + r * p.ldx + c);
  }
}

template <typename T>
__global__ __launch_bounds__(64) void probe_kernel(const u32x4* a, const u32x4* b, f32x4* out) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  TT<T>::mma(acc, a[threadIdx.x], b[threadIdx.x]);
  out[threadIdx.x] = acc;
}

int grid_for(int64_t items, int per_block, int cap = 8192) {
  int64_t g = (items + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}
bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// head-major [heads, n_pad, 64] -> token-major [n, heads*64]; one thread = 16 bytes (8 elements)
__global__ __launch_bounds__(256) void heads_to_tokens_kernel(ovg_heads_to_tokens_params p, int64_t total) {
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c = (int)(idx & 7);                       // 16-byte chunk of the 64-wide head row
    const int64_t t = idx >> 3;
    const int h = (int)(t % p.heads);
    const int64_t row = t / p.heads;
    const u32x4 v = *reinterpret_cast<const u32x4*>(static_cast<const unsigned char*>(p.x) + (((int64_t)h * p.n_pad + row) * 64 + c * 8) * 2);
    *reinterpret_cast<u32x4*>(static_cast<unsigned char*>(p.y) + ((row * p.ldy) + h * 64 + c * 8) * 2) = v;
  }
}

extern "C" int ovg_heads_to_tokens(const ovg_heads_to_tokens_params* p, void* stream) {
  if (!p || !p->x || !p->y || p->n <= 0 || p->n_pad < p->n || p->heads <= 0 || p->ldy < p->heads * 64 || (p->ldy % 8)) return OVG_E_ARG;
  if (p->dtype != OVG_BF16 && p->dtype != OVG_F16) return OVG_E_DTYPE;
  if (!al16(p->x) || !al16(p->y)) return OVG_E_ARG;
  const int64_t total = p->n * p->heads * 8;
  OVG_LAUNCH(heads_to_tokens_kernel, dim3(grid_for(total, 256, 1 << 16)), dim3(256), 0, static_cast<hipStream_t>(stream), *p, total);
  OVG_CHECK_LAUNCH();
  return OVG_OK;
}

// Exact combination of two attention results over disjoint key sets (header: ovg_attn_merge). One thread = 4
// consecutive d of one (token, head); lse is head-major [16, n_pad] as written by the attention kernels.
template <typename T, bool X3 = false>   // X3 (OVG_F16X2): a, b, out are (hi, lo) f16 plane pairs
__global__ __launch_bounds__(256) void attn_merge_kernel(ovg_attn_merge_params p, int64_t total) {
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c = (int)(idx & 255);                     // 4-element chunk of the 1024-wide row
    const int64_t row = idx >> 8;
    const int h = c >> 4;
    const float la = p.lse_a[(int64_t)h * p.n_pad + row], lb = p.lse_b[(int64_t)h * p.n_pad + row];
    const float m = fmaxf(la, lb);
    const float wa = __builtin_amdgcn_exp2f(la - m), wb = __builtin_amdgcn_exp2f(lb - m);
    const float inv = 1.0f / (wa + wb);
    const T* a = static_cast<const T*>(p.a) + row * p.lda + c * 4;
    const T* b = static_cast<const T*>(p.b) + row * p.ldb + c * 4;
    float r[4];
    if constexpr (X3) {
      const f16_t* al = static_cast<const f16_t*>(p.a_lo) + row * p.lda + c * 4;
      const f16_t* bl = static_cast<const f16_t*>(p.b_lo) + row * p.ldb + c * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        r[i] = (wa * (static_cast<float>(a[i]) + static_cast<float>(al[i])) + wb * (static_cast<float>(b[i]) + static_cast<float>(bl[i]))) * inv;
      store4_hilo(static_cast<f16_t*>(p.out) + row * p.ldo + c * 4, static_cast<f16_t*>(p.out_lo) + row * p.ldo + c * 4, r[0], r[1], r[2], r[3]);
      continue;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (wa * TT<T>::to_f32(a[i]) + wb * TT<T>::to_f32(b[i])) * inv;
    store4<T>(static_cast<T*>(p.out) + row * p.ldo + c * 4, r[0], r[1], r[2], r[3]);
  }
}

extern "C" int ovg_attn_merge(const ovg_attn_merge_params* p, void* stream) {
  if (!p || !p->a || !p->b || !p->out || !p->lse_a || !p->lse_b || p->rows <= 0 || p->n_pad < p->rows) return OVG_E_ARG;
  if (p->lda < OVG_C || p->ldb < OVG_C || p->ldo < OVG_C || (p->lda % 4) || (p->ldb % 4) || (p->ldo % 4)) return OVG_E_ARG;
  if (!al16(p->a) || !al16(p->b) || !al16(p->out)) return OVG_E_ARG;
  const int64_t total = p->rows * 256;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(grid_for(total, 256, 1 << 16)), block(256);
  switch (p->dtype) {
    case OVG_BF16: OVG_LAUNCH((attn_merge_kernel<bf16_t>), grid, block, 0, st, *p, total); break;
    case OVG_F16: OVG_LAUNCH((attn_merge_kernel<f16_t>), grid, block, 0, st, *p, total); break;
    case OVG_F32: OVG_LAUNCH((attn_merge_kernel<float>), grid, block, 0, st, *p, total); break;
    case OVG_F16X2:
      if (!p->a_lo || !p->b_lo || !p->out_lo || !al16(p->a_lo) || !al16(p->b_lo) || !al16(p->out_lo)) return OVG_E_ARG;
      OVG_LAUNCH((attn_merge_kernel<f16_t, true>), grid, block, 0, st, *p, total); break;
    default: return OVG_E_DTYPE;
  }
  OVG_CHECK_LAUNCH();
  return OVG_OK;
}

// f32 [rows, k] -> dtype [rows, k_pad], zero beyond k (header: ovg_pack_weights); one thread = 8 output elements
template <typename T, bool X3 = false>
__global__ __launch_bounds__(256) void pack_weights_kernel(ovg_pack_weights_params p, int64_t total, int cpr) {
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t row = idx / cpr;
    const int c0 = (int)(idx - row * cpr) * 8;
    const float* src = p.src + row * p.lds;
    T* dst = static_cast<T*>(p.dst) + row * p.ldd + c0;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (c0 + i) < p.k ? src[c0 + i] : 0.f;
    if constexpr (X3) {
      f16_t* dlo = static_cast<f16_t*>(p.dst_lo) + row * p.ldd + c0;
      store4_hilo(dst, dlo, v[0], v[1], v[2], v[3]);
      store4_hilo(dst + 4, dlo + 4, v[4], v[5], v[6], v[7]);
    } else {
      store4<T>(dst, v[0], v[1], v[2], v[3]);
      store4<T>(dst + 4, v[4], v[5], v[6], v[7]);
    }
  }
}

extern "C" int ovg_pack_weights(const ovg_pack_weights_params* p, void* stream) {
  if (!p || !p->src || !p->dst || p->rows <= 0 || p->k <= 0 || p->k_pad < p->k || (p->k_pad % 8) || p->lds < p->k || p->ldd < p->k_pad) return OVG_E_ARG;
  if (!al16(p->dst) || (p->ldd % 8)) return OVG_E_ARG;
  const int cpr = (int)(p->k_pad / 8);
  const int64_t total = p->rows * cpr;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(grid_for(total, 256, 1 << 16)), block(256);
  switch (p->dtype) {
    case OVG_BF16: OVG_LAUNCH((pack_weights_kernel<bf16_t>), grid, block, 0, st, *p, total, cpr); break;
    case OVG_F16: OVG_LAUNCH((pack_weights_kernel<f16_t>), grid, block, 0, st, *p, total, cpr); break;
    case OVG_F32: OVG_LAUNCH((pack_weights_kernel<float>), grid, block, 0, st, *p, total, cpr); break;
    case OVG_F16X2:
      if (!p->dst_lo || !al16(p->dst_lo)) return OVG_E_ARG;
      OVG_LAUNCH((pack_weights_kernel<f16_t, true>), grid, block, 0, st, *p, total, cpr); break;
    default: return OVG_E_DTYPE;
  }
  OVG_CHECK_LAUNCH();
  return OVG_OK;
}

extern "C" int ovg_layernorm(const ovg_layernorm_params* p, void* stream) {
  if (!p || !p->x || !p->y || !p->weight || !p->bias || p->rows <= 0) return OVG_E_ARG;
  if (!al16(p->x) || !al16(p->y) || !al16(p->weight) || !al16(p->bias) || (p->ldx % 4) || (p->ldy % 4)) return OVG_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(grid_for(p->rows, 4)), block(256);
  if (p->out_f32) OVG_LAUNCH((layernorm_kernel<float, true>), grid, block, 0, st, *p);
  else switch (p->dtype) {
    case OVG_BF16: OVG_LAUNCH((layernorm_kernel<bf16_t, false>), grid, block, 0, st, *p); break;
    case OVG_F16: OVG_LAUNCH((layernorm_kernel<f16_t, false>), grid, block, 0, st, *p); break;
    case OVG_F32: OVG_LAUNCH((layernorm_kernel<float, true>), grid, block, 0, st, *p); break;
    case OVG_F16X2:
      if (!p->y_lo || !al16(p->y_lo)) return OVG_E_ARG;
      OVG_LAUNCH((layernorm_kernel<f16_t, false, true>), grid, block, 0, st, *p); break;
    default: return OVG_E_DTYPE;
  }
  OVG_CHECK_LAUNCH();
  return OVG_OK;
}

extern "C" int ovg_im2col(const ovg_im2col_params* p, void* stream) {
  if (!p || !p->img || !p->out || p->V <= 0) return OVG_E_ARG;
  if (p->Hpx % 14 || p->Wpx % 14 || p->k_pad % 64 || p->k_pad < p->C * 196) return OVG_E_ARG;
  if (p->mode == 0 ? (p->C != 3) : (p->C != 2 || !p->img2 || !p->depth_stats || p->views_per_batch <= 0)) return OVG_E_ARG;
  if (!al16(p->out)) return OVG_E_ARG;
  const int gh = p->Hpx / 14, gw = p->Wpx / 14;
  const int64_t total = p->V * gh * gw * (p->k_pad / 8);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(grid_for(total, 256, 65536)), block(256);
  switch (p->dtype) {
    case OVG_BF16: OVG_LAUNCH((im2col_kernel<bf16_t>), grid, block, 0, st, *p, gh, gw, total); break;
    case OVG_F16: OVG_LAUNCH((im2col_kernel<f16_t>), grid, block, 0, st, *p, gh, gw, total); break;
    case OVG_F32: OVG_LAUNCH((im2col_kernel<float>), grid, block, 0, st, *p, gh, gw, total); break;
    case OVG_F16X2:
      if (!p->out_lo || !al16(p->out_lo)) return OVG_E_ARG;
      OVG_LAUNCH((im2col_kernel<f16_t, true>), grid, block, 0, st, *p, gh, gw, total); break;
    default: return OVG_E_DTYPE;
  }
  OVG_CHECK_LAUNCH();
  return OVG_OK;
}

extern "C" int ovg_depth_stats(const ovg_depth_stats_params* p, void* stream) {
  if (!p || !p->depth || !p->mask || !p->stats || !p->partial || p->B <= 0 || p->n_per_batch <= 0 || p->nblocks <= 0) return OVG_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  OVG_LAUNCH(depth_stats_stage1, dim3(p->nblocks, (unsigned)p->B), dim3(256), 0, st, *p);
  OVG_CHECK_LAUNCH();
  OVG_LAUNCH(depth_stats_stage2, dim3((unsigned)p->B), dim3(64), 0, st, *p);
  OVG_CHECK_LAUNCH();
  return OVG_OK;
}

extern "C" int ovg_dino_specials(const ovg_dino_specials_params* p, void* stream) {
  if (!p || !p->x || !p->cls || !p->pos0 || (p->n_reg > 0 && !p->reg) || p->V <= 0 || (p->ldx % 4)) return OVG_E_ARG;
  OVG_LAUNCH(dino_specials_kernel, dim3((unsigned)(p->V * (1 + p->n_reg))), dim3(256), 0, static_cast<hipStream_t>(stream), *p);
  OVG_CHECK_LAUNCH();
  return OVG_OK;
}

extern "C" int ovg_assemble_tokens(const ovg_assemble_params* p, void* stream) {
  if (!p || !p->xd || !p->norm_w || !p->norm_b || !p->camera_token || !p->register_token || !p->cam_add || !p->placeholder || !p->out)
    return OVG_E_ARG;
  if (p->V <= 0 || p->S <= 0 || p->view0 < 0 || p->tokens_per_view <= p->n_special || (p->ldo % 4) || (p->ldxd % 4)) return OVG_E_ARG;
  OVG_LAUNCH(assemble_kernel, dim3(grid_for(p->V * p->tokens_per_view, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), *p);
  OVG_CHECK_LAUNCH();
  return OVG_OK;
}

extern "C" int ovg_copy_rows(const ovg_copy_rows_params* p, void* stream) {
  if (!p || !p->x || !p->y || p->rows <= 0 || p->n <= 0 || (p->n % 4) || (p->ldx % 4) || (p->ldy % 4)) return OVG_E_ARG;
  OVG_LAUNCH(copy_rows_kernel, dim3(grid_for(p->rows * (p->n / 4), 256)), dim3(256), 0, static_cast<hipStream_t>(stream), *p);
  OVG_CHECK_LAUNCH();
  return OVG_OK;
}

extern "C" int ovg_probe_mfma(const void* a, const void* b, float* out, int dtype, void* stream) {
  if (!a || !b || !out) return OVG_E_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const u32x4* A = static_cast<const u32x4*>(a);
  const u32x4* B = static_cast<const u32x4*>(b);
  f32x4* O = reinterpret_cast<f32x4*>(out);
  switch (dtype) {
    case OVG_BF16: OVG_LAUNCH((probe_kernel<bf16_t>), dim3(1), dim3(64), 0, st, A, B, O); break;
    case OVG_F16: OVG_LAUNCH((probe_kernel<f16_t>), dim3(1), dim3(64), 0, st, A, B, O); break;
    case OVG_F32: OVG_LAUNCH((probe_kernel<float>), dim3(1), dim3(64), 0, st, A, B, O); break;
    default: return OVG_E_DTYPE;
  }
  OVG_CHECK_LAUNCH();
  return OVG_OK;
}

extern "C" int ovg_abi_version(void) { return OVG_ABI_VERSION; }
#define OVG_STR2(x) #x
#define OVG_STR(x) OVG_STR2(x)
extern "C" const char* ovg_build_info(void) { return "libomnivggt_hip gfx950 abi=" OVG_STR(OVG_ABI_VERSION) " " __DATE__ " " __TIME__; }
