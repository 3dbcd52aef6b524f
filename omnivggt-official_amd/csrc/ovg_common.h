// Device-side helpers shared by the gfx950 kernels of libomnivggt_hip.so.
// CDNA4 only: wave64, MFMA 16x16 tiles, 16-byte operand fragments.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/omnivggt_hip.h"

typedef __bf16 bf16_t;
typedef _Float16 f16_t;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(8 * sizeof(_Float16)))) _Float16 f16x8;

#define OVG_DEV __device__ __forceinline__

// ---------------------------------------------------------------------------
// Type traits.  A "fragment" is 16 raw bytes per lane: 8 bf16/f16 (one
// 16x16x32 MFMA) or 4 f32 (four 16x16x4 MFMAs).  Lane l supplies row (l&15)
// of its operand and the 16-byte chunk (4*kk + (l>>4)) of that row; because A
// and B use the same chunk->k assignment the contraction is exact for any
// hardware k ordering inside the chunk.
// ---------------------------------------------------------------------------
template <typename T> struct TT;
template <> struct TT<bf16_t> {
  static constexpr int kDtype = OVG_BF16;
  static constexpr int kPerChunk = 8;  // elements per 16 B
  static OVG_DEV void mma(f32x4& c, const u32x4& a, const u32x4& b) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  static OVG_DEV bf16_t from_f32(float f) { return static_cast<bf16_t>(f); }
  static OVG_DEV float to_f32(bf16_t v) { return static_cast<float>(v); }
};
template <> struct TT<f16_t> {
  static constexpr int kDtype = OVG_F16;
  static constexpr int kPerChunk = 8;
  static OVG_DEV void mma(f32x4& c, const u32x4& a, const u32x4& b) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static OVG_DEV f16_t from_f32(float f) { return static_cast<f16_t>(f); }
  static OVG_DEV float to_f32(f16_t v) { return static_cast<float>(v); }
};
template <> struct TT<float> {
  static constexpr int kDtype = OVG_F32;
  static constexpr int kPerChunk = 4;
  static OVG_DEV void mma(f32x4& c, const u32x4& a, const u32x4& b) {
    f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], c, 0, 0, 0);
  }
  static OVG_DEV float from_f32(float f) { return f; }
  static OVG_DEV float to_f32(float v) { return v; }
};

// pack 4 f32 -> 4 T (8 B for 16-bit types, 16 B for f32) and store
template <typename T> OVG_DEV void store4(T* dst, float a, float b, float c, float d);
template <> OVG_DEV void store4<float>(float* dst, float a, float b, float c, float d) {
  f32x4 v = {a, b, c, d};
  *reinterpret_cast<f32x4*>(dst) = v;
}
template <> OVG_DEV void store4<bf16_t>(bf16_t* dst, float a, float b, float c, float d) {
  typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4;
  bf16x4 v = {static_cast<bf16_t>(a), static_cast<bf16_t>(b), static_cast<bf16_t>(c), static_cast<bf16_t>(d)};
  *reinterpret_cast<bf16x4*>(dst) = v;
}
template <> OVG_DEV void store4<f16_t>(f16_t* dst, float a, float b, float c, float d) {
  typedef __attribute__((__vector_size__(4 * sizeof(_Float16)))) _Float16 f16x4;
  f16x4 v = {static_cast<f16_t>(a), static_cast<f16_t>(b), static_cast<f16_t>(c), static_cast<f16_t>(d)};
  *reinterpret_cast<f16x4*>(dst) = v;
}

// ---------------------------------------------------------------------------
// Split-f16 mode (OVG_F16X2, "f32x"): x ~ hi + lo with hi = f16(x) (saturated at the largest finite f16) and lo = f16(x - hi),
// stored in two f16 tensors of the same shape. |x - hi - lo| <= 2^-22 |x| while lo is a normal f16 (|x| >= 2^-3), <= 2^-25 ABSOLUTE
// below: for |x| < 2^-3 -- typical ViT weights (|w| ~ 0.01-0.05), most LayerNorm outputs -- lo is a SUBNORMAL f16 and the split is good to
// ~2^-25 / |x| relative (1e-6 at |x| = 0.03, not 2^-22). The mode therefore relies on the f16 MFMA and the f32 <-> f16 conversions keeping
// denormals (gfx950: they do; denormal mode is not touched anywhere in this library) -- gpu_selftest.py test_f32x feeds planes whose lo is
// subnormal everywhere through ovg_linear on both tile sizes and requires their contribution in the result. Kernels take the mode as a `bool X3` template flag next to T = f16_t:
// every contraction runs three MFMAs (lo*hi, hi*lo, hi*hi -- small terms first) into the same f32 accumulator.
// ---------------------------------------------------------------------------
OVG_DEV float f16_sat(float x) { return __builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f); }
OVG_DEV void store4_hilo(f16_t* hi, f16_t* lo, float a, float b, float c, float d) {
  typedef __attribute__((__vector_size__(4 * sizeof(_Float16)))) _Float16 f16x4;
  const f16x4 h = {static_cast<f16_t>(f16_sat(a)), static_cast<f16_t>(f16_sat(b)), static_cast<f16_t>(f16_sat(c)), static_cast<f16_t>(f16_sat(d))};
  const f16x4 l = {static_cast<f16_t>(f16_sat(a - static_cast<float>(h[0]))), static_cast<f16_t>(f16_sat(b - static_cast<float>(h[1]))),
                   static_cast<f16_t>(f16_sat(c - static_cast<float>(h[2]))), static_cast<f16_t>(f16_sat(d - static_cast<float>(h[3])))};
  *reinterpret_cast<f16x4*>(hi) = h;
  *reinterpret_cast<f16x4*>(lo) = l;
}
// one value -> (hi, lo)
OVG_DEV void split_hilo(float x, f16_t& hi, f16_t& lo) {
  hi = static_cast<f16_t>(f16_sat(x));
  lo = static_cast<f16_t>(f16_sat(x - static_cast<float>(hi)));
}

// LDS tile of rows of RB bytes (RB = 128 or 256), 16-byte chunks XOR-swizzled so
// that ds_read_b128 of {16 rows x one chunk column} is bank-conflict free:
//   128 B rows: chunk ^= (row>>1)&7   (two rows share one 256 B bank row)
//   256 B rows: chunk ^= row&15
template <int RB> OVG_DEV int swz_off(int row, int chunk) {
  if constexpr (RB == 128) return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
  else return row * 256 + ((chunk ^ (row & 15)) << 4);
}

// Column of key n inside a 16-bit V^T row (ovg_qkv writes it, the 16-bit attention kernels read it): inside every block of 32
// keys the order is pos = 8 g + 4 h + i for key = 16 h + 4 g + i (g < 4, h < 2, i < 4), i.e. the 16-byte chunk g of a block
// holds keys {4g .. 4g+3, 16+4g .. 16+4g+3} -- exactly the PV MFMA B^T fragment of lane group g, so a K / V^T tile goes from
// global memory to its LDS image by LDS-DMA (16-byte granules, no register pass) and a fragment is one ds_read_b128.
// f32 V^T rows stay in natural order.
OVG_DEV int vt_pos16(int n) {
  const int k = n & 31;
  return (n & ~31) | ((k & 12) << 1) | ((k & 16) >> 2) | (k & 3);
}

// x / d and x % d for 0 <= x < 2^24, 1 <= d: one v_rcp_f32 per divisor (hoisted by the caller), then ~8 VALU per
// division instead of the ~30 of the generic 32-bit sequence hipcc expands (the epilogues divide token indices by the
// sequence length / tokens per view / patch-grid width once per 16-row block: r02 profile of the QKV kernel showed
// 790 v_mul_lo_u32 + 730 v_cndmask_b32 of division code against 256 MFMAs).
struct FastDiv {
  int d; float rd;
  OVG_DEV explicit FastDiv(int d_) : d(d_), rd(__builtin_amdgcn_rcpf((float)d_)) {}
  OVG_DEV void divmod(int x, int& q, int& r) const {
    q = (int)((float)x * rd);                  // off by at most one either way
    r = x - q * d;
    if (r < 0) { r += d; --q; }
    if (r >= d) { r -= d; ++q; }
  }
};

// sum over the 4 lanes {l, l^16, l^32, l^48} with the gfx950 swap instructions (no LDS crossbar, see ovg_attn16.h)
OVG_DEV float quad16_sum(float v) {
  unsigned u = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  unsigned a = r[0], b = r[1];
  const float s = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
  u = __builtin_bit_cast(unsigned, s);
  r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  a = r[0]; b = r[1];
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}

OVG_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// XCD-aware 1-D block remap (8 XCDs, block b runs on XCD b%8 -- speed only):
// consecutive logical ids share an XCD/L2.  Bijective for any n.
OVG_DEV int xcd_remap(int b, int n) {
  const int q = n >> 3, r = n & 7, x = b & 7, i = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// LDS-DMA of one 16-byte granule per lane: lane l of the wave lands at lds_dst + 16 l (lds_dst wave-uniform, in M0).
// Inline asm on purpose: hipcc treats the builtin form as a FLAT access that may touch LDS ("pending flat"), after which
// EVERY s_waitcnt it generates nearby becomes vmcnt(0) / lgkmcnt(0) -- the prefetch is drained where it was issued and the
// ds_read pipelining around it is lost. Opaque to the compiler, the transfers are covered by explicit counted s_waitcnt vmcnt
// at the use sites (attention: run_tiles in ovg_attn16.h; GEMM: the residual epilogue of the 256 x 256 kernels).
OVG_DEV void lds_dma16(const void* gsrc, uint32_t lds_dst) {
  // M0 is written without being declared (it is a reserved register: hipcc rejects it as a clobber); hipcc itself never keeps a
  // value live in M0 on gfx9+ -- it materialises M0 immediately in front of each of its own uses (LDS-DMA builtin, s_sendmsg).
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_dst), "v"(gsrc) : "memory");
}

// The same with the address split into a wave-uniform 64-bit base (SGPR pair) and a 32-bit per-lane byte offset: a k-stage then advances by ONE
// scalar add on the base instead of a 64-bit vector add per request, and a request's address costs one VGPR instead of two.
OVG_DEV void lds_dma16_s(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
}

// Raw buffer descriptor (4 SGPRs): base, stride 0, num_records = bytes, untyped dword format. Lanes whose offset (VGPR offset + immediate;
// the SGPR offset is NOT range-checked) reaches num_records read as zero and fetch nothing.
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
OVG_DEV i32x4 make_srd(const void* base, uint32_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int32_t)(uint32_t)a);
  r[1] = __builtin_amdgcn_readfirstlane((int32_t)((uint32_t)(a >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int32_t)bytes);
  r[3] = 0x00020000;
  return r;
}
// LDS-DMA of one 16-byte granule per lane through a buffer descriptor: lane l lands at lds_dst + 16 l (lds_dst wave-uniform, in M0), its
// source is srd.base + voff + soff; a lane whose voff is out of range deposits zeros (the implicit-GEMM convolution's border taps, rows past
// the end of a GEMM piece). Inline asm for the reason given at lds_dma16 -- measured: with the builtin form
// (__builtin_amdgcn_raw_ptr_buffer_load_lds) hipcc put s_waitcnt vmcnt(0) in front of the fragment reads of EVERY k-stage, draining the
// ring it cannot tell apart from the slot being read. The transfers are covered by the explicit counted waits of the loops that issue them.
OVG_DEV void buffer_dma16(uint32_t lds_dst, uint32_t voff, const i32x4 srd, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_dst), "v"(voff), "s"(srd), "s"(soff) : "memory");
}

// hipGetLastError() is per-thread and sticky until read: the host framework's own benign failures
// (hipEventQuery -> NotReady, hipPointerGetAttributes on pageable memory, ...) must not be mistaken
// for a failed launch of ours, so the slot is drained immediately before every launch.
#define OVG_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)
#define OVG_CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return OVG_E_LAUNCH; } while (0)
