/*
 * omnivggt_hip.h -- C ABI of libomnivggt_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the OmniVGGT multi-view aggregator hot path
 * (reference: omnivggt/models/omnivggt_aggregator.py:130-305,
 * omnivggt/models/aggregator.py:312-341, omnivggt/layers/{block,attention,
 * mlp,rope,patch_embed,vision_transformer}.py).  The reference has no FFI of
 * its own (pure PyTorch); each entry below names the reference call site it
 * replaces.  See INTEGRATION.md for the ctypes binding a maintainer would add.
 *
 * Conventions (all entries):
 *   - `int fn(const <params>*, void* hip_stream)`; returns OVG_OK (0) or a
 *     negative OVG_E_* code.  Never throws, never allocates or frees, never
 *     synchronises the device.
 *   - every pointer is a DEVICE pointer owned by the caller and must stay valid
 *     until the stream reaches the call; 16-byte aligned unless noted.
 *   - `dtype` selects the storage/MFMA input type of activations and weights:
 *     OVG_BF16 / OVG_F16 (throughput modes, f32 accumulate), OVG_F32
 *     (parity mode, exact-f32 MFMA) or OVG_F16X2 (split-f16 parity mode, see the enum).  The residual stream, LayerNorm
 *     statistics, softmax statistics, biases, LayerScale gammas, q/k-norm
 *     affine parameters and the RoPE table are always f32.
 *   - thread-safe for distinct streams; the library holds NO mutable state: every tuning choice is either
 *     derived from the call's shapes or passed in the parameter struct (`tile`, `variant`).
 *   - workspace is caller-provided; ovg_block_workspace_bytes() answers how much a block call needs.
 */
#ifndef OMNIVGGT_HIP_H
#define OMNIVGGT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history
 * 2: + DPT-head entries (ovg_head_layernorm, ovg_conv, ovg_upsample, ovg_dpt_out) and ovg_unproject
 * 3: + head-parallel sharding (ovg_attn_params.kv_heads / out_bh_stride, ovg_block_params.skip_attention, ovg_heads_to_tokens)
 * 4: per-call GEMM tile selector (`tile`) replacing the process-global debug setter of ABI 3, optional
 *    log-sum-exp output of ovg_flash_attn + ovg_attn_merge (two-launch local-first sharded attention),
 *    ovg_block_workspace_bytes, ovg_pack_weights, split-KV attention (kv_splits / ws_part / ws_lse, ovg_attn_plan)
 * 5-6: camera head entry; 16-bit V^T rows in the PV fragment order (LDS-DMA staged attention)
 * 7: split-KV workspace SIZES travel with the pointers (ws_part_bytes / ws_lse_bytes: an undersized workspace is OVG_E_ARG
 *    instead of an out-of-bounds write), ovg_camera_tables (camera-modality injection tables built on the device), OVG_F32 in the DPT-head
 *    entries, ovg_attn_plan_out.main_rows / tail_q_tile (the tail split of long attention launches is part of the queryable plan)
 * 8: OVG_F16X2 -- the split-f16 compute mode ("f32x": every 16-bit operand tensor is a PAIR of f16 planes hi + lo, products run as
 *    three f16 MFMAs hi*hi + hi*lo + lo*hi with f32 accumulation: ~2^-22 per product instead of 2^-8 (bf16) / 2^-11 (f16) at
 *    a third of the 16-bit MFMA rate; the `*_lo` pointers below, NULL / ignored for the other dtypes; single GPU and the K / V^T
 *    all-gather sharded form incl. ovg_attn_merge);
 *    ovg_attn_params.fallback_count / ovg_block_params.attn_fallback_count (telemetry of the speculative bf16 softmax)
 * 9: split-KV partials are f32 (ovg_attn_plan_out.part_bytes doubles) and a launch may split only the rows beyond its last full round
 *    along the keys (key-split tail, reported through main_rows / tail_q_tile); OVG_TILE_256P / OVG_TILE_DMA_M name the round-5 lab GEMM
 *    forms (OVG_E_UNSUPPORTED unless the library was built with -DOVG_LAB_GEMM)
 * 10: ovg_dpt_tail -- the output stage of the DPT head (upsample + position embedding + conv3x3 + ReLU + conv1x1 + activation) as one launch
 * 11: the lab GEMM selectors of ABI 9 (OVG_TILE_256P / OVG_TILE_DMA_M) are removed (OVG_E_ARG); the key-split tail of ABI 9 now also follows
 *     the 512-row attention launches (ovg_attn_plan_out: q_tile == tail_q_tile == 512, splits = key ranges of the tail rows, partials sized
 *     for the tail rows) when the caller passes a split workspace */
#define OVG_ABI_VERSION 11

enum { OVG_BF16 = 0, OVG_F16 = 1, OVG_F32 = 2,
       /* split-f16 ("f32x", the <= 1e-4 mode with throughput): a value x is stored as hi = f16(x) (saturated at +-65504) in the tensor the
        * ordinary pointer names and lo = f16(x - hi) in a second f16 tensor of the same shape / strides named by the matching `*_lo` pointer.
        * x ~ hi + lo to 2^-22 relative (|x| >= 2^-3; absolute 2^-25 below, where lo is an f16 subnormal). A GEMM / attention contraction over
        * such operands is hi*hi + hi*lo + lo*hi on the f16 MFMA (f16 x f16 products are exact in f32) with f32 accumulation, the dropped
        * lo*lo term is 2^-22 relative. Everything that is f32 in the other modes stays f32. */
       OVG_F16X2 = 3 };

enum {
  OVG_OK = 0,
  OVG_E_ARG = -1,      /* null pointer / bad shape / misalignment            */
  OVG_E_DTYPE = -2,    /* unsupported dtype                                   */
  OVG_E_LAUNCH = -3,   /* hipGetLastError() != hipSuccess after a launch      */
  OVG_E_UNSUPPORTED = -4
};

/* Model constants of the path (omnivggt_aggregator.py:19-37). */
#define OVG_C 1024      /* embed dim            */
#define OVG_H 16        /* heads                */
#define OVG_D 64        /* head dim             */
#define OVG_HID 4096    /* MLP hidden           */
#define OVG_KV_TILE 64  /* key tile: K/V^T buffers are padded to this */
#define OVG_MAX_SEG 8   /* K/V segments per attention call (= ranks)  */

int ovg_abi_version(void);
/* human readable build string (static storage) */
const char* ovg_build_info(void);

/* ------------------------------------------------------------------ *
 * LayerNorm over rows of 1024 (nn.LayerNorm: block.py:50,67 norm1/norm2,
 * vision_transformer.py:264 final DINO norm).  x is the f32 residual
 * stream with row stride ldx (elements); y is [rows,1024] in `dtype`
 * (out_f32 != 0: y is f32 regardless of dtype).
 * ------------------------------------------------------------------ */
typedef struct {
  const float* x; int64_t ldx;
  void* y; int64_t ldy;
  const float* weight; const float* bias;
  int64_t rows; float eps; int dtype; int out_f32;
  void* y_lo;                    /* OVG_F16X2: lo plane of y (same ldy) */
} ovg_layernorm_params;
int ovg_layernorm(const ovg_layernorm_params*, void* stream);

/* ------------------------------------------------------------------ *
 * Generic linear  Y = epilogue(X @ W^T + bias)   (nn.Linear / addmm).
 * X [M,K] ld=ldx, W [N,K] ld=ldw, both `dtype`, K a multiple of 64.
 * epilogue:
 *   OVG_EPI_STORE : y[m,n] = acc + bias               (y dtype, or f32 if out_f32)
 *   OVG_EPI_GELU  : y[m,n] = gelu_erf(acc + bias)     (mlp.py:35-36)
 *   OVG_EPI_RES   : y_f32[m,n] = res_f32[m,n] + gamma[n]*(acc+bias)
 *                   (+ inject[(m/inj_period),n] when m % inj_period == 0)
 *                   (attention.py:75 + layer_scale.py:27 + block.py:105-106,
 *                    omnivggt_aggregator.py:284-301 camera injection)
 *   OVG_EPI_PATCH : y_f32[(m/p0)*p1 + off + m%p0, n] = acc + bias + table[(m%p0)+1, n]
 *                   (patch_embed.py:75-77 + vision_transformer.py:220-224)
 * ------------------------------------------------------------------ */
enum { OVG_EPI_STORE = 0, OVG_EPI_GELU = 1, OVG_EPI_RES = 2, OVG_EPI_PATCH = 3 };
/* workgroup tile of the GEMM kernels: 128 x 128 (4 waves, register-staged, up to 3 workgroups per CU) or 256 x 256 (8 waves, 4-slot LDS-DMA
 * ring, software-pipelined "free-running" main loop with one barrier per k-stage, 1 workgroup per CU, 16-bit dtypes, N % 256 == 0);
 * AUTO picks by shape (ovg_gemm.hip: choose_256). ABI 11: the round-5 lab selectors (4 = persistent 256 x 256, 8 = DMA-in-M flag) are gone --
 * any other value is OVG_E_ARG. */
enum { OVG_TILE_AUTO = 0, OVG_TILE_128 = 1, OVG_TILE_256 = 2,
       /* A/B flag, OR-ed onto any of the three: the same kernels with the r02 epilogue forms (erf_as GELU; per-lane 8- / 16-byte stores in
        * the accumulator layout instead of whole lines staged through the idle LDS -- ovg_gemm.hip); -DOVG_AB_VARIANTS builds only */
       OVG_TILE_R02_EPILOGUE = 16, OVG_TILE_128X = 17, OVG_TILE_256X = 18 };
typedef struct {
  const void* x; int64_t ldx;
  const void* w; int64_t ldw;
  const float* bias;             /* [N] or NULL */
  void* y; int64_t ldy;
  int64_t M; int64_t N; int64_t K;
  int dtype; int epilogue; int out_f32;
  /* RES */
  const float* res; int64_t ldres; const float* gamma;
  const float* inject; int64_t inj_period;   /* inject may be NULL */
  /* PATCH */
  const float* table; int64_t p0; int64_t p1; int64_t row_off;
  int tile;   /* OVG_TILE_AUTO (shape heuristic), OVG_TILE_128 or OVG_TILE_256 (16-bit dtypes, N % 256 == 0); an impossible request is OVG_E_ARG */
  /* OVG_F16X2: lo planes of x / w (same ldx / ldw) and, for 16-bit outputs (STORE / GELU without out_f32), of y (same ldy) */
  const void* x_lo; const void* w_lo; void* y_lo;
} ovg_linear_params;
int ovg_linear(const ovg_linear_params*, void* stream);

/* ------------------------------------------------------------------ *
 * Fused QKV projection (attention.py:52-58): qkv = X @ Wqkv^T + b, then per
 * head LayerNorm(64) on q,k (if qk_norm), 2-D RoPE on q,k (if rope),
 * q *= q_scale, and head-major stores:
 *   q  [B*H, nq_pad, 64]     k [B*H, nk_pad, 64]     vt [B*H, 64, nk_pad]
 * Row m of X is token n = m % seq of batch b = m / seq; RoPE positions are
 * derived from t = m % tokens_per_view: t < 5 -> (0,0) else
 * (1 + (t-5)/grid_w, 1 + (t-5)%grid_w)   (omnivggt_aggregator.py:215-224).
 * rope_cos/rope_sin: f32 [max_pos,16] (rope.py:86-117, 16 unique freqs), max_pos <= 128 (staged in LDS).
 * Padding rows/cols of q,k,vt are never written (caller zero-fills once).
 * Column order of a vt row (ABI 6). OVG_F32: natural (column n = key n). OVG_BF16 / OVG_F16: inside every block of 32 keys
 * column 8 g + 4 h + i holds key 16 h + 4 g + i (g < 4, h < 2, i < 4) -- each 16-byte chunk is then exactly the B^T fragment
 * of one lane group of the PV MFMA, so the attention kernels move K / V^T tiles global -> LDS by LDS-DMA (no register pass)
 * and read a fragment with one ds_read_b128. ovg_qkv writes this order and ovg_flash_attn expects it; the permutation is
 * local to 32-key blocks, so slicing / exchanging vt buffers at 64-key granularity (segments, ranks, heads) is unaffected.
 * ------------------------------------------------------------------ */
typedef struct {
  const void* x; int64_t ldx;       /* [M,1024] dtype */
  const void* w;                    /* [3072,1024] dtype */
  const float* bias;                /* [3072] */
  void* q; void* k; void* vt;
  int64_t M; int64_t seq; int64_t nq_pad; int64_t nk_pad;
  int dtype;
  int qk_norm; const float* qn_w; const float* qn_b; const float* kn_w; const float* kn_b; float qk_eps;
  int rope; const float* rope_cos; const float* rope_sin; int max_pos;
  int64_t tokens_per_view; int grid_w; int n_special;
  float q_scale;
  int part;   /* 0 = q,k,v; 1 = k and v only; 2 = q only (sharded path: K/V first, all-gather || Q) */
  int tile;   /* OVG_TILE_* as in ovg_linear_params */
  /* OVG_F16X2: lo planes (same shapes / strides as their hi tensors) */
  const void* x_lo; const void* w_lo; void* q_lo; void* k_lo; void* vt_lo;
} ovg_qkv_params;
int ovg_qkv(const ovg_qkv_params*, void* stream);

/* ------------------------------------------------------------------ *
 * Flash attention forward, D=64, no mask, non-causal
 * (F.scaled_dot_product_attention, attention.py:61-66).  q must already be
 * multiplied by softmax_scale*log2(e) (ovg_qkv does it): the kernel uses
 * exp2.  K/V^T arrive as `nseg` segments (1 on a single GPU; one per rank
 * after the view-sharded all-gather) -- softmax runs across all of them.
 *   q   [BH, nq_pad, 64]
 *   seg[i].k [BH, nk_pad_i, 64], seg[i].vt [BH, 64, nk_pad_i] (16-bit: columns in the ovg_qkv order above), nk_i valid keys
 *   out [B*nq, H*64] token-major (row = (bh/H)*nq + n, col = (bh%H)*64 + d), ld = ldo
 * ------------------------------------------------------------------ */
typedef struct { const void* k; const void* vt; int64_t nk; int64_t nk_pad;
                 const void* k_lo; const void* vt_lo;   /* OVG_F16X2: lo planes of k / vt (same shapes) */
} ovg_kv_segment;
typedef struct {
  const void* q; int64_t nq; int64_t nq_pad;
  ovg_kv_segment seg[OVG_MAX_SEG]; int nseg;
  void* out; int64_t ldo;
  int64_t BH; int dtype;
  int variant;   /* 0 = default; >0 selects tuning variants */
  /* head-parallel (all-to-all) sharding, both 0 otherwise:
   *   kv_heads > 0: K / V^T segments hold kv_heads heads; batch entry bh reads head bh % kv_heads (the BH entries
   *                 are (source rank, head) pairs of queries that share this rank's heads);
   *   out_bh_stride > 0: head-major output, out + bh * out_bh_stride + q * ldo + d (ldo >= 64) instead of the
   *                 token-major row (bh / 16) * nq + q, column (bh % 16) * 64 + d. */
  int kv_heads; int64_t out_bh_stride;
  /* optional f32 [BH, nq_pad]: lse[bh, q] = log2(sum_k exp2(s[q, k])) over the keys of THIS call (s = the
   * pre-scaled logits). With it, two calls over disjoint key sets are combined exactly by ovg_attn_merge --
   * the view-sharded all-gather path runs the local keys while the remote ones are still in flight. */
  float* lse;
  /* Split-KV (16-bit dtypes): a launch whose (batch entry, q tile) units do not fill the chip evenly -- 688 workgroups
   * on 512 resident slots at 8 views, the same per rank of an 8-GPU run -- is cut along the KEY axis into kv_splits
   * passes per unit; every pass writes a normalised partial result + its log-sum-exp into the caller's workspace and
   * a second (tiny) launch combines them exactly. kv_splits: 0 = the library decides (ovg_attn_plan; never splits
   * when ws_part / ws_lse are NULL), 1 = never, 2..8 = force. ws_part: `part_bytes`, ws_lse: `lse_bytes` of ovg_attn_plan.
   * Units of one (batch entry, split) run next to each other, so the K / V^T range an XCD streams shrinks by kv_splits.
   * ws_part_bytes / ws_lse_bytes: sizes of the two buffers; a call whose plan needs more than it was given is OVG_E_ARG.
   * ABI 9: the partials are f32 (normalised O per key range + its log-sum-exp: a split launch now agrees with the unsplit one to ~1e-6
   * before the final rounding; the 16-bit partials of ABI <= 8 were 6.9e-3 apart), so part_bytes doubled; and with kv_splits == 0 a launch
   * with a fractional last round may run its full rounds unsplit and only the remaining rows cut along the keys ("key-split tail":
   * ovg_attn_plan_out.main_rows < nq with tail_q_tile == q_tile), whose workspace covers those rows only. */
  int kv_splits; void* ws_part; float* ws_lse; int64_t ws_part_bytes; int64_t ws_lse_bytes;
  /* OVG_F16X2: lo planes of q and out (same shapes / strides). That mode runs one launch of 256-row tiles: no split-KV, no kv_heads /
   * head-major output (OVG_E_UNSUPPORTED), lse is available. */
  const void* q_lo; void* out_lo;
  /* Telemetry of the speculative softmax (OVG_BF16 default kernels): optional DEVICE counter; every workgroup whose speculative pass
   * failed its verification and re-ran with the lazy-rescale body adds 1 (one atomic per such workgroup; nothing is written otherwise,
   * the caller zeroes it). The counter says how often the fast path did not pay: a workgroup that re-ran returns the lazy-rescale result, one
   * that did not returns the speculative pass's, accepted by its row-sum / finiteness check (and, for the order-pinned bf16 body, by the
   * build-time disassembly guard of build.py). NULL = not counted. */
  uint32_t* fallback_count;
  /* ABI 9: CUs the launch plan may count on (0 = all the device has). One process per GPU shares its chip with RCCL's kernels in the
   * sharded run -- every channel of an exchange in flight holds a workgroup slot while the attention launch it overlaps runs -- so the q tile,
   * the tail split and the split-KV factor are planned for `cus` CUs instead of quantising against slots that are not there. ovg_attn_plan
   * answers for the same value. */
  int cus;
} ovg_attn_params;
int ovg_flash_attn(const ovg_attn_params*, void* stream);

/* Host-only query: how ovg_flash_attn would run this call with kv_splits == 0 (needs nq, BH, dtype, variant, the
 * segments' nk; pointers are ignored) and how much split workspace the caller should provide for it. */
typedef struct {
  int splits; int q_tile; int64_t part_bytes; int64_t lse_bytes;
  /* tail split of long launches (ABI 7): rows [0, main_rows) of every batch entry run as q_tile-row tiles in a first launch, the rest as
   * tail_q_tile-row tiles in a second one; main_rows == nq and tail_q_tile == 0 when the call is one launch */
  int64_t main_rows; int tail_q_tile;
} ovg_attn_plan_out;
int ovg_attn_plan(const ovg_attn_params*, ovg_attn_plan_out* out);

/* Combine two attention results over disjoint key sets (same queries):
 *   w_a = 2^(lse_a - m), w_b = 2^(lse_b - m), m = max(lse_a, lse_b);  out = (w_a * a + w_b * b) / (w_a + w_b)
 * a, b, out: [rows, 1024] `dtype` token-major (row strides lda / ldb / ldo; out may alias a or b);
 * lse_a, lse_b: f32 [16, n_pad] head-major as written by ovg_flash_attn (row = token n, B = 1). */
typedef struct {
  const void* a; int64_t lda; const float* lse_a;
  const void* b; int64_t ldb; const float* lse_b;
  void* out; int64_t ldo;
  int64_t rows; int64_t n_pad; int dtype;
  const void* a_lo; const void* b_lo; void* out_lo;   /* OVG_F16X2: lo planes of a / b / out (same strides) */
} ovg_attn_merge_params;
int ovg_attn_merge(const ovg_attn_merge_params*, void* stream);

/* ------------------------------------------------------------------ *
 * One pre-LN transformer block (block.py:81-107):
 *   x = x + ls1 * proj(attn(qkv(norm1(x))));  x = x + ls2 * fc2(gelu(fc1(norm2(x))))
 * run as LN -> QKV -> flash-attn -> proj(RES) -> LN -> fc1(GELU) -> fc2(RES [+inject]).
 * x_in/x_out are f32 with row strides (x_out may alias x_in; they may also be
 * the two halves of a (.., 2C) concat buffer, omnivggt_aggregator.py:250).
 * ------------------------------------------------------------------ */
typedef struct {
  const void* n1_w; const void* n1_b;     /* f32 [1024] */
  const void* qkv_w; const float* qkv_b;  /* dtype [3072,1024]; f32 [3072] */
  const float* qn_w; const float* qn_b; const float* kn_w; const float* kn_b; /* f32 [64] or NULL */
  const void* proj_w; const float* proj_b;
  const float* ls1;                        /* f32 [1024] (ones if no LayerScale) */
  const void* n2_w; const void* n2_b;
  const void* fc1_w; const float* fc1_b;   /* dtype [4096,1024] */
  const void* fc2_w; const float* fc2_b;   /* dtype [1024,4096] */
  const float* ls2;
  const void* qkv_w_lo; const void* proj_w_lo; const void* fc1_w_lo; const void* fc2_w_lo;   /* OVG_F16X2: lo planes of the four GEMM weights */
} ovg_block_weights;

typedef struct {
  ovg_block_weights w;
  const float* x_in; int64_t ld_in;
  float* x_out; int64_t ld_out;
  int64_t M;                 /* tokens processed by this rank            */
  int64_t seq;               /* attention sequence length (per batch)     */
  int64_t BH;                /* (M/seq) * 16                               */
  int64_t nq_pad; int64_t nk_pad;
  int dtype; float ln_eps; int qk_norm; int rope; float qk_eps;
  const float* rope_cos; const float* rope_sin; int max_pos;
  int64_t tokens_per_view; int grid_w; int n_special;
  const float* inject; int64_t inj_period;     /* fc2 epilogue; NULL = none  */
  /* caller-provided workspace */
  void* ws_xn;    /* [M,1024] dtype */
  void* ws_q; void* ws_k; void* ws_vt;
  void* ws_attn;  /* [M,1024] dtype */
  void* ws_hid;   /* [M,4096] dtype */
  /* remote K/V segments (view-sharded global attention): when nseg_extra > 0 the
   * attention step uses {local K/V^T} + extra[]; the caller fills `extra`
   * between ovg_block_attn_prologue and ovg_block_attn_epilogue. */
  ovg_kv_segment extra[OVG_MAX_SEG]; int nseg_extra; int local_seg_index;
  int attn_variant;
  int qkv_part;  /* ovg_block_attn_prologue only: 0 = LN1 + q,k,v; 1 = LN1 + k,v; 2 = q only (no LN) */
  /* optional hipEvent_t handles recorded on `stream` immediately before / after the
   * flash-attention launch (bench.py: live per-kernel timing); NULL = not recorded */
  void* ev_attn_start; void* ev_attn_stop;
  int skip_attention;  /* ovg_block_attn_epilogue only: ws_attn already holds the attention output (head-parallel sharding) */
  int gemm_tile;       /* OVG_TILE_* forwarded to the four GEMMs of the block (tests force a tile; 0 in production) */
  /* optional split-KV workspace of the block's attention launch (ovg_attn_params.ws_part / ws_lse + their sizes; NULL = never split) */
  void* ws_attn_part; float* ws_attn_lse; int attn_kv_splits; int64_t ws_attn_part_bytes; int64_t ws_attn_lse_bytes;
  /* OVG_F16X2: lo planes of the six scratch tensors (same sizes as their hi tensors; ovg_block_workspace_bytes reports the size of ONE plane) */
  void* ws_xn_lo; void* ws_q_lo; void* ws_k_lo; void* ws_vt_lo; void* ws_attn_lo; void* ws_hid_lo;
  uint32_t* attn_fallback_count;   /* forwarded to ovg_attn_params.fallback_count of the block's attention launch (NULL = not counted) */
  int attn_cus;                    /* forwarded to ovg_attn_params.cus (0 = the whole device) */
} ovg_block_params;
/* whole block */
int ovg_block_forward(const ovg_block_params*, void* stream);
/* split form for the sharded path: prologue = LN1 + QKV (writes ws_q/k/vt);
 * epilogue = attention (over local+extra segments) + proj + MLP. */
int ovg_block_attn_prologue(const ovg_block_params*, void* stream);
int ovg_block_attn_epilogue(const ovg_block_params*, void* stream);

/* Workspace query (host only, no device work): bytes of each caller-provided scratch buffer of a block call
 * with the given M, seq, BH, nq_pad, nk_pad and dtype (all other fields ignored). */
typedef struct { int64_t xn, q, k, vt, attn, hid, total; } ovg_block_workspace;
int ovg_block_workspace_bytes(const ovg_block_params*, ovg_block_workspace* out);

/* Weight pre-pack (inference.py:321-325 loads f32 checkpoints): dst[r, :k] = convert(src[r, :k]) to `dtype`,
 * dst[r, k:k_pad] = 0.  src f32 [rows, k] (ld lds), dst `dtype` [rows, k_pad] (ld ldd, k_pad % 8 == 0).
 * nn.Linear weights pack with k_pad = k; the Conv2d(k=14,s=14) patch weights with k = C_in*196, k_pad = 640 / 448. */
typedef struct {
  const float* src; int64_t lds; void* dst; int64_t ldd;
  int64_t rows; int64_t k; int64_t k_pad; int dtype;
  void* dst_lo;                  /* OVG_F16X2: lo plane (same ldd) */
} ovg_pack_weights_params;
int ovg_pack_weights(const ovg_pack_weights_params*, void* stream);

/* ------------------------------------------------------------------ *
 * Patch im2col (Conv2d k=14,s=14 as a GEMM: patch_embed.py:65,75-77).
 * img f32 [V,C,Hpx,Wpx]; out [V*gh*gw, k_pad] dtype, element k = c*196+ky*14+kx,
 * zero for k >= C*196.  mode 0: (img[c]-mean[c])/std[c]  (omnivggt_aggregator.py:143)
 * mode 1: depth/mask: c=0 -> depth/(mean_b+1e-8)*mask, c=1 -> mask
 *         (omnivggt_aggregator.py:107-128,197); depth_stats = {sum,count} per batch
 * ------------------------------------------------------------------ */
typedef struct {
  const float* img; const float* img2;   /* mode1: img=depth [V,H,W], img2=mask [V,H,W] */
  void* out; int64_t k_pad;
  int64_t V; int C; int Hpx; int Wpx; int dtype; int mode;
  float mean[3]; float std[3];
  const double* depth_stats;  /* mode1: [B][2] = {sum, count}; V = B*views_per_batch */
  int64_t views_per_batch;
  void* out_lo;               /* OVG_F16X2: lo plane of out */
} ovg_im2col_params;
int ovg_im2col(const ovg_im2col_params*, void* stream);

/* masked depth statistics (omnivggt_aggregator.py:118-123): stats[b] = {sum of
 * depth where mask>0, count}.  partial: workspace of 2*nblocks doubles. */
typedef struct {
  const float* depth; const float* mask; int64_t B; int64_t n_per_batch;
  double* stats; double* partial; int nblocks;
} ovg_depth_stats_params;
int ovg_depth_stats(const ovg_depth_stats_params*, void* stream);

/* DINOv2 special rows (vision_transformer.py:220-224): x[v,0]=cls+pos[0], x[v,1..4]=reg */
typedef struct {
  float* x; int64_t ldx; int64_t V; int64_t tokens_per_view;
  const float* cls; const float* pos0; const float* reg; int n_reg;
} ovg_dino_specials_params;
int ovg_dino_specials(const ovg_dino_specials_params*, void* stream);

/* Token assembly before the AA trunk (omnivggt_aggregator.py:147-156,202-213 and
 * vision_transformer.py:264-268): for each view v and token t
 *   t == 0     : camera_token[slot] + cam_add[v]
 *   1 <= t < 5 : register_token[slot][t-1]
 *   t >= 5     : LayerNorm_eps(xd[v,t]) + (depth_row[v] >= 0 ? depth_tok[depth_row[v]*P0+t-5] : placeholder)
 * slot = ((view0 + v) % S == 0) ? 0 : 1   (aggregator.py:343-366). */
typedef struct {
  const float* xd; int64_t ldxd;        /* DINO residual stream [V*P,1024] */
  const float* norm_w; const float* norm_b; float eps;
  const float* camera_token;            /* [2,1024] */
  const float* register_token;          /* [2,4,1024] */
  const float* cam_add;                 /* [V,1024] */
  const float* depth_tok;               /* [Sd*P0,1024] or NULL */
  const int32_t* depth_row;             /* [V] index into depth_tok views or -1 */
  const float* placeholder;             /* [1024] */
  float* out; int64_t ldo;
  int64_t V; int64_t S; int64_t tokens_per_view; int n_special;
  int64_t view0;                        /* global index of local view 0 (view-sharded ranks) */
} ovg_assemble_params;
int ovg_assemble_tokens(const ovg_assemble_params*, void* stream);

/* strided f32 row copy/add helper: y[r, :n] = x[r, :n] (+ add[r / period, :n] on rows r%period==0) */
typedef struct {
  const float* x; int64_t ldx; float* y; int64_t ldy; int64_t rows; int64_t n;
} ovg_copy_rows_params;
int ovg_copy_rows(const ovg_copy_rows_params*, void* stream);

/* ================================================================== *
 * DPT dense-prediction head (reference heads/dpt_head.py:185-304,
 * heads/head_act.py:61-125).  OVG_BF16 / OVG_F16 (f32 accumulate) and, since ABI 7, OVG_F32 (the parity mode: exact-f32 MFMA,
 * activations / weights / outputs all f32, Cin % 32 == 0 instead of % 64).  Activations are NHWC: [n_img, H, W, C] with a pixel
 * stride (ld, in elements) >= C.
 * ================================================================== */

/* LayerNorm over rows of 2048 of the aggregator output list (dpt_head.py:219 `self.norm`):
 * input row of output row r is x + ((r / p0) * p1 + row_off + r % p0) * ldx  (p0 patch tokens kept
 * per view out of p1 tokens per view, skipping the row_off special tokens); y is [rows, 2048] dtype. */
typedef struct {
  const float* x; int64_t ldx;
  void* y; int64_t ldy;
  const float* weight; const float* bias;
  int64_t rows; int64_t p0; int64_t p1; int64_t row_off;
  float eps; int dtype;
} ovg_head_layernorm_params;
int ovg_head_layernorm(const ovg_head_layernorm_params*, void* stream);

/* NHWC convolution as an implicit GEMM on the MFMA (nn.Conv2d k=1 / k=3 stride 1|2 pad k/2, and
 * nn.ConvTranspose2d with kernel == stride via `upshuffle`; dpt_head.py:221-240 projects /
 * resize_layers, :274-304 scratch convs, :357-399 ResidualConvUnit, :445-470 FeatureFusionBlock):
 *   y[i, oy, ox, co] = act( sum_{ky,kx,ci} x[i, oy*stride+ky-pad, ox*stride+kx-pad, ci] * w[co][(ky*k+kx)*Cin + ci]
 *                           + bias[co] + pos(ox, oy, co) + add1[i,oy,ox,co] + add2[i,oy,ox,co] )
 *   x  [n_img, H, W, Cin]  (ldx);   w [Cout_gemm, k*k*Cin] dtype, taps-major then channels;   bias f32 or NULL
 *   pos_x [OW, Cout/2], pos_y [OH, Cout/2] f32 or both NULL: the UV position embedding of dpt_head.py:262-272
 *         (channels [0, Cout/2) depend on ox only, [Cout/2, Cout) on oy only)
 *   add1 / add2: optional tensors in output geometry and dtype (ld1 / ld2) -- residual / skip sums
 *   relu != 0: clamp at 0 after all additions (the in-place ReLU that opens every ResidualConvUnit is folded
 *         into the producer of its input)
 *   upshuffle = s > 1 (requires ksize == 1, stride == 1): ConvTranspose2d(k = s, stride = s); w is
 *         [s*s*Cout, Cin] ordered (dy, dx, co), bias is [Cout]; GEMM row (i, oy, ox), column (dy, dx, co)
 *         is stored at y[i, oy*s+dy, ox*s+dx, co].  pos / add are not supported together with upshuffle.
 *   out_f32 != 0: y is f32.
 * Constraints: Cin % 64 == 0 (OVG_F32: % 32); w holds w_rows rows (a multiple of 128, zero rows beyond the Cout_gemm real ones,
 * Cout_gemm = Cout, or s*s*Cout with upshuffle: then it must itself be the multiple of 128); Cout % 4 == 0. */
typedef struct {
  const void* x; int64_t ldx;
  const void* w; const float* bias;
  void* y; int64_t ldy;
  const void* add1; int64_t ld1; const void* add2; int64_t ld2;
  const float* pos_x; const float* pos_y;
  int64_t n_img; int H; int W; int Cin; int Cout; int w_rows; int ksize; int stride; int upshuffle;
  int relu; int out_f32; int dtype;
} ovg_conv_params;
int ovg_conv(const ovg_conv_params*, void* stream);

/* Bilinear resize, align_corners = True (F.interpolate at dpt_head.py:242-247, :466), NHWC, C % 8 == 0:
 * y[i, oy, ox, :] = lerp of x[i, :, :, :] at (oy*(H-1)/(OH-1), ox*(W-1)/(OW-1)), plus the optional UV
 * position embedding (pos_x [OW, C/2], pos_y [OH, C/2], dpt_head.py:249-250) -- both in `dtype`. */
typedef struct {
  const void* x; int64_t ldx; void* y; int64_t ldy;
  const float* pos_x; const float* pos_y;
  int64_t n_img; int H; int W; int OH; int OW; int C; int dtype;
} ovg_upsample_params;
int ovg_upsample(const ovg_upsample_params*, void* stream);

/* Output stage of the DPT head (second half of scratch.output_conv2, dpt_head.py:252-258, and
 * head_act.py:61-125): conv1x1(32 -> out_dim) + activation on the ReLU'd 32-channel map that
 * ovg_conv (k = 3, Cout = 32, relu, out_f32) produced; f32 in, f32 out:
 *   activation 0 = "exp" (depth head, out_dim 2: val = exp(v0)), 1 = "inv_log" (point head, out_dim 4:
 *   val_j = sign(v_j) * expm1(|v_j|), j < 3); confidence = 1 + exp(v_last) ("expp1") in both.
 *   h [npix, 32] f32;  w2 [out_dim, 32] f32, b2 [out_dim] f32;  val [npix, out_dim-1] f32;  conf [npix] f32 */
typedef struct {
  const float* h; const float* w2; const float* b2;
  float* val; float* conf;
  int64_t npix; int out_dim; int activation;
} ovg_dpt_out_params;
int ovg_dpt_out(const ovg_dpt_out_params*, void* stream);

/* The whole output stage of the DPT head in one launch (ABI 10; 16-bit dtypes): what ovg_upsample (with the UV position tables) ->
 * ovg_conv (k = 3, Cin = 128, Cout = 32, relu, out_f32) -> ovg_dpt_out compute, without the upsampled map (n x OH x OW x 128) or the 32-channel
 * map ever reaching HBM (dpt_head.py:242-258, head_act.py:61-125):
 *   x [n, H, W, 128] dtype (pixel stride ldx elements)  --bilinear, align_corners-->  [n, OH, OW, 128] (+ pos_x [OW, 64] / pos_y [OH, 64] f32,
 *   both or neither), rounded to dtype as ovg_upsample does;  w1 [>= 32 rows, 9 * 128] dtype taps-major (row stride ldw1 elements: the
 *   zero-padded matrix ovg_conv takes is fine), b1 f32 [32] or NULL;  w2 f32 [out_dim, 32], b2 f32 [out_dim];
 *   val f32 [n, OH, OW, out_dim - 1], conf f32 [n, OH, OW]; activation as ovg_dpt_out_params.
 * Any other channel count / dtype is OVG_E_UNSUPPORTED (callers run the three-launch form). */
typedef struct {
  const void* x; int64_t ldx;
  const float* pos_x; const float* pos_y;
  const void* w1; int64_t ldw1; const float* b1;
  const float* w2; const float* b2;
  float* val; float* conf;
  int64_t n_img; int H; int W; int OH; int OW; int C; int out_dim; int activation; int dtype;
} ovg_dpt_tail_params;
int ovg_dpt_tail(const ovg_dpt_tail_params*, void* stream);

/* ------------------------------------------------------------------ *
 * Post-processing on the device: depth maps -> world-frame point maps,
 * utils/geometry.py:151-266 (unproject_depth_map_to_point_map -> depth_to_world_coords_points ->
 * depth_to_cam_coords_points), which the reference runs as a per-frame numpy loop on the host.
 *   depth [S, H, W] f32;  cam [S, 16] f32 per frame: cam-to-world rotation row-major (9), cam-to-world
 *   translation (3), fu, fv, cu, cv (closed_form_inverse_se3 of the extrinsic and the intrinsic entries,
 *   prepared by the caller);  out [S, H, W, 3] f32.
 *   x_cam = (u - cu) * d / fu, y_cam = (v - cv) * d / fv evaluated in double and rounded to f32 exactly like
 *   the numpy expression (int64 pixel grid promotes it to float64), then world = R * cam + t in double (the
 *   reference's inverse pose is float64) rounded to f32 on store (the reference returns float64).
 * ------------------------------------------------------------------ */
typedef struct {
  const float* depth; const float* cam; float* out;
  int64_t S; int H; int W;
} ovg_unproject_params;
int ovg_unproject(const ovg_unproject_params*, void* stream);

/* head-major -> token-major: x [heads, n_pad, 64] dtype -> y [n, heads*64] dtype (row stride ldy), the layout the
 * proj GEMM reads; used after the return all-to-all of the head-parallel sharded attention. */
typedef struct {
  const void* x; int64_t n_pad; void* y; int64_t ldy; int64_t n; int heads; int dtype;
} ovg_heads_to_tokens_params;
int ovg_heads_to_tokens(const ovg_heads_to_tokens_params*, void* stream);

/* ------------------------------------------------------------------
 * Camera head: replaces CameraHead.forward / trunk_fn, omnivggt/heads/camera_head.py:84-154, for
 * one batch element: S camera tokens (row m of `tokens`, 2048 f32, row stride ld_tokens elements: the slot-0 token of
 * every view in the LAST aggregator layer, camera_head.py:96-100) -> out [iters][S][9] f32, the activated pose
 * encodings of every refinement round (absT_quaR_FoV: translation and quaternion linear, field of view ReLU,
 * heads/head_act.py:12-35). One call issues every launch of every round; nothing is read back in between.
 *   GEMM weights (mod_w [6144,2048], blk[i].qkv_w [6144,2048], proj_w [2048,2048], fc1_w [8192,2048],
 *   fc2_w [2048,8192], pb1_w [1024,2048]) in `dtype` (OVG_BF16 / OVG_F16 / OVG_F32; nn.Linear layout, K contiguous,
 *   16-byte aligned); every vector, the 9-wide embed_pose [2048,9] and pose_branch.fc2 [9,1024] matrices and all
 *   activations that carry state (residual stream, statistics, softmax, pose) f32. With OVG_F32 (the parity mode) the
 *   GEMM operands and the activation buffers between kernels are f32 too and the products run on the exact-f32 MFMA:
 *   no rounding point below f32 anywhere. dim must be 2048, heads 16 (head dim 128), trunk_depth <= OVG_CAMERA_MAX_TRUNK,
 *   S <= 4096. ws: caller-owned scratch of >= ovg_camera_head_workspace_bytes(S, dtype) bytes (returns -1 on bad args).
 * ------------------------------------------------------------------ */
#define OVG_CAMERA_MAX_TRUNK 4
typedef struct {
  const float *n1_w, *n1_b, *n2_w, *n2_b, *ls1, *ls2;
  const void* qkv_w; const float* qkv_b;
  const void* proj_w; const float* proj_b;
  const void* fc1_w; const float* fc1_b;
  const void* fc2_w; const float* fc2_b;
} ovg_camera_block_weights;
typedef struct {
  const float* tokens; int64_t ld_tokens;
  int32_t S; int32_t iters; int32_t dtype; int32_t trunk_depth; int32_t dim; int32_t heads;
  const float *token_norm_w, *token_norm_b, *trunk_norm_w, *trunk_norm_b;
  const float* empty_pose;
  const float *embed_w, *embed_b;
  const void* mod_w; const float* mod_b;
  ovg_camera_block_weights blk[OVG_CAMERA_MAX_TRUNK];
  const void* pb1_w; const float* pb1_b;
  const float *pb2_w, *pb2_b;
  void* ws; int64_t ws_bytes;
  float* out;
} ovg_camera_head_params;
int64_t ovg_camera_head_workspace_bytes(int32_t S, int32_t dtype);
int ovg_camera_head(const ovg_camera_head_params*, void* stream);

/* ------------------------------------------------------------------
 * Camera-modality injection tables, built on the device without a host round trip (replaces, per forward:
 * ZeroAggregator.normalize_extrinsics omnivggt_aggregator.py:85-105 with closed_form_inverse_se3 utils/geometry.py:269-318,
 * extri_intri_to_pose_encoding utils/pose_enc.py:11-62 with mat_to_quat utils/rotation.py:47-109, the 25 pose_embeddings /
 * camera_adapters Linear pairs omnivggt_aggregator.py:62-75,172,211,277,286 and the zero-padded scatter :174-178,278-282):
 *   enc[b, r]       = pose encoding (t, quat xyzw, fov_h, fov_w) of camera index[r] of batch b after normalisation
 *                     (first selected camera -> identity, translations / mean distance of the others to it)
 *   emb[g, b*Sc+r]  = pose_w[g] enc[b, r] + pose_b[g]                                   g < G (= depth + 1 tables)
 *   tables[g, b*S+s] = adapt_w[g] emb[g, b*Sc+r] + adapt_b[g]   if s == index[r]        (exact-f32 MFMA)
 *                      adapt_b[g]                               otherwise (Linear of a zero row)
 * extrinsics [B,S,3,4] (world-to-camera), intrinsics [B,S,3,3] f32; index: DEVICE int32 [Sc], strictly the caller's
 * camera_gt_index (values in [0, S): the array is on the device, so the entry cannot reject it -- reads through an
 * out-of-range entry are clamped into [0, S) and its scatter is skipped, never an out-of-bounds access; a view may appear more than once,
 * as in the reference: the statistics run over the list as given and the duplicate entries scatter identical rows); pose_w [G*1024, 9] f32 row-major, pose_b [G*1024]; adapt_w [G,1024,1024] f32
 * (nn.Linear layout, 16-byte aligned), adapt_b [G,1024]; enc [B*Sc, 9] and emb [G, B*Sc, 1024] are caller-owned scratch
 * (NULL allowed when Sc == 0); tables [G, B*S, 1024] f32. Three launches (one when Sc == 0); nothing is read back.
 * ------------------------------------------------------------------ */
typedef struct {
  const float* extrinsics; const float* intrinsics; const int32_t* index;
  int32_t B; int32_t S; int32_t Sc; int32_t H; int32_t W; int32_t G;
  const float* pose_w; const float* pose_b;
  const float* adapt_w; const float* adapt_b;
  float* enc; float* emb; float* tables;
} ovg_camera_tables_params;
int ovg_camera_tables(const ovg_camera_tables_params*, void* stream);

/* MFMA lane-map probe (diagnostics; tools/selftest.py): fills out[64*4] with
 * acc of one 16x16 MFMA for dtype given raw 16-byte A/B fragments per lane. */
int ovg_probe_mfma(const void* a_frag, const void* b_frag, float* out, int dtype, void* stream);


#ifdef __cplusplus
}
#endif
#endif
