// Camera-modality injection tables on the device (gfx950): entry ovg_camera_tables.
//
// Replaces, per forward, what the reference runs as ~60 tiny ATen ops + 25 full-tensor zero-padded scatters
// (omnivggt/models/omnivggt_aggregator.py:85-105 normalize_extrinsics, :158-182 pose embedding of the GT cameras,
// :273-287 per-layer injection; utils/geometry.py:269-318 closed_form_inverse_se3; utils/pose_enc.py:11-62
// extri_intri_to_pose_encoding; utils/rotation.py:47-109 mat_to_quat) and what round 2 of this repo did with a
// device -> host copy + 26 GEMM launches + a scatter. Three launches, no host round trip:
//   1. cam_encode_kernel  one workgroup per batch element: select, normalise, pose-encode         (O(Sc) scalars)
//   2. cam_embed_kernel   emb[g, r, :] = pose_w[g] enc[r] + pose_b[g]  AND  tables[g, :, :] = adapt_b[g] (the value of
//                         every view without a GT camera: Linear of a zero row)                   (HBM: G*K*4 KB written)
//   3. cam_adapt_kernel   the G adapters as ONE batched exact-f32 MFMA GEMM over the camera rows only, results
//                         scattered to their view rows                                           (HBM: G*4 MB of weights read once)
#include "ovg_common.h"
#include <cmath>

namespace {

constexpr int ENC = 9;   // absT_quaR_FoV: t(3), quat xyzw(4), fov_h, fov_w  (pose_enc.py:48-59)

// ---- 1. selection + normalisation + pose encoding -------------------------------------------------------------------
// rel = [E_r; 0 0 0 1] * inverse([E_0; 0 0 0 1])  with E_0 the FIRST selected camera (omnivggt_aggregator.py:93-97):
//   R_rel = R_r R_0^T,  t_rel = t_r - R_rel t_0 ... evaluated the way the reference does it: the 4 x 4 product with
//   inv0 = [R_0^T | -R_0^T t_0] (geometry.py:303-316), so t_rel = R_r (-R_0^T t_0) + t_r.
// The index array lives on the device and cannot be validated by the host entry: every read through it clamps the view into [0, S)
// and cam_adapt_kernel skips the scatter of an out-of-range entry (its view keeps the bias row), so a bad index can give a wrong
// table row but never an out-of-bounds access (the Python front end rejects such lists before they reach the device). Duplicate
// indices are not supported: two rows would race for one table row (the reference's scatter is last-write-wins).
OVG_DEV int cam_view(const ovg_camera_tables_params& p, int r) {
  const int v = p.index[r];
  return v < 0 ? 0 : (v >= p.S ? p.S - 1 : v);
}

__global__ __launch_bounds__(256) void cam_encode_kernel(ovg_camera_tables_params p) {
  __shared__ float red[256];
  __shared__ float inv0[12];      // R_0^T (row-major 3 x 3), then -R_0^T t_0
  __shared__ float c0[3];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* ext_b = p.extrinsics + (int64_t)b * p.S * 12;
  const float* intr_b = p.intrinsics + (int64_t)b * p.S * 9;
  if (tid == 0) {
    const float* e0 = ext_b + (int64_t)cam_view(p, 0) * 12;
    float Rt[9], t0[3] = {e0[3], e0[7], e0[11]};
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Rt[i * 3 + j] = e0[j * 4 + i];
    for (int i = 0; i < 9; ++i) inv0[i] = Rt[i];
    for (int i = 0; i < 3; ++i) inv0[9 + i] = -(Rt[i * 3] * t0[0] + Rt[i * 3 + 1] * t0[1] + Rt[i * 3 + 2] * t0[2]);
  }
  __syncthreads();
  // translation of the first selected camera after the change of frame (its own rel): every thread needs it for the distances
  if (tid == 0) {
    const float* e0 = ext_b + (int64_t)cam_view(p, 0) * 12;
    for (int i = 0; i < 3; ++i)
      c0[i] = e0[i * 4] * inv0[9] + e0[i * 4 + 1] * inv0[10] + e0[i * 4 + 2] * inv0[11] + e0[i * 4 + 3];
  }
  __syncthreads();
  // pass 1: mean distance of cameras 1 .. Sc-1 to camera 0 (translations of the relative poses, omnivggt_aggregator.py:99-103)
  float dsum = 0.f;
  for (int r = 1 + tid; r < p.Sc; r += 256) {
    const float* e = ext_b + (int64_t)cam_view(p, r) * 12;
    float d2 = 0.f;
    for (int i = 0; i < 3; ++i) {
      const float t = e[i * 4] * inv0[9] + e[i * 4 + 1] * inv0[10] + e[i * 4 + 2] * inv0[11] + e[i * 4 + 3];
      const float d = t - c0[i];
      d2 += d * d;
    }
    dsum += sqrtf(d2);
  }
  red[tid] = dsum;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  float scale = 1.0f;
  if (p.Sc > 1) scale = fmaxf(red[0] / (float)(p.Sc - 1), 1e-6f);
  // pass 2: the encoding of every selected camera
  for (int r = tid; r < p.Sc; r += 256) {
    const int view = cam_view(p, r);
    const float* e = ext_b + (int64_t)view * 12;
    const float* k = intr_b + (int64_t)view * 9;
    float R[9], t[3];
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) R[i * 3 + j] = e[i * 4] * inv0[j] + e[i * 4 + 1] * inv0[3 + j] + e[i * 4 + 2] * inv0[6 + j];
      t[i] = e[i * 4] * inv0[9] + e[i * 4 + 1] * inv0[10] + e[i * 4 + 2] * inv0[11] + e[i * 4 + 3];
      if (p.Sc > 1) t[i] = t[i] / scale;
    }
    // matrix -> quaternion, best-conditioned branch (rotation.py:47-109), xyzw, real part >= 0 (:126-138)
    const float m00 = R[0], m01 = R[1], m02 = R[2], m10 = R[3], m11 = R[4], m12 = R[5], m20 = R[6], m21 = R[7], m22 = R[8];
    const float raw[4] = {1.0f + m00 + m11 + m22, 1.0f + m00 - m11 - m22, 1.0f - m00 + m11 - m22, 1.0f - m00 - m11 + m22};
    float qa[4];
    int best = 0;
    for (int i = 0; i < 4; ++i) {
      qa[i] = raw[i] > 0.f ? sqrtf(raw[i]) : 0.f;
      if (qa[i] > qa[best]) best = i;                    // argmax: first maximum wins, like torch
    }
    float cand[4];                                       // candidate `best` in (r, i, j, k) order
    switch (best) {
      case 0: cand[0] = qa[0] * qa[0]; cand[1] = m21 - m12; cand[2] = m02 - m20; cand[3] = m10 - m01; break;
      case 1: cand[0] = m21 - m12; cand[1] = qa[1] * qa[1]; cand[2] = m10 + m01; cand[3] = m02 + m20; break;
      case 2: cand[0] = m02 - m20; cand[1] = m10 + m01; cand[2] = qa[2] * qa[2]; cand[3] = m12 + m21; break;
      default: cand[0] = m10 - m01; cand[1] = m20 + m02; cand[2] = m21 + m12; cand[3] = qa[3] * qa[3]; break;
    }
    const float den = 2.0f * fmaxf(qa[best], 0.1f);
    float q[4] = {cand[1] / den, cand[2] / den, cand[3] / den, cand[0] / den};   // -> xyzw
    if (q[3] < 0.f) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    float* o = p.enc + ((int64_t)b * p.Sc + r) * ENC;
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
    o[3] = q[0]; o[4] = q[1]; o[5] = q[2]; o[6] = q[3];
    o[7] = 2.0f * atanf(((float)p.H / 2.0f) / k[4]);     // fov_h from fy, fov_w from fx (pose_enc.py:52-53)
    o[8] = 2.0f * atanf(((float)p.W / 2.0f) / k[0]);
  }
}

// ---- 2. pose embeddings of the camera rows + bias fill of the whole table --------------------------------------------
// thread = (table g, feature n): its 9 pose weights stay in registers; consecutive threads = consecutive n = coalesced stores
__global__ __launch_bounds__(256) void cam_embed_kernel(ovg_camera_tables_params p) {
  const int g = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
  const int64_t K = (int64_t)p.B * p.S, R = (int64_t)p.B * p.Sc;
  const float ab = p.adapt_b[(int64_t)g * OVG_C + n];
  float* trow = p.tables + (int64_t)g * K * OVG_C + n;
  for (int64_t v = 0; v < K; ++v) trow[v * OVG_C] = ab;
  if (R == 0) return;
  const float* wr = p.pose_w + ((int64_t)g * OVG_C + n) * ENC;
  float w[ENC];
#pragma unroll
  for (int k = 0; k < ENC; ++k) w[k] = wr[k];
  const float pb = p.pose_b[(int64_t)g * OVG_C + n];
  float* erow = p.emb + (int64_t)g * R * OVG_C + n;
  for (int64_t r = 0; r < R; ++r) {
    const float* e = p.enc + r * ENC;                    // wave-uniform: scalar loads
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < ENC; ++k) acc = fmaf(e[k], w[k], acc);
    erow[r * OVG_C] = acc + pb;
  }
}

// ---- 3. the G adapters as one batched exact-f32 MFMA GEMM over the camera rows ------------------------------------
// One wave = 16 output features x 64 camera rows of one adapter: the W slice (16 rows x 4 KB) is streamed once, 16 bytes per
// lane per 16-wide k step (each weight element is used by exactly this wave: no LDS), the embedding rows are the B operand
// (L2 resident). MFMA j of a step contracts k = 16 s + 4 (lane >> 4) + j on both operands (TT<float>::mma's convention).
// acc[rt][r] = out[row 16 rt + (lane & 15)][feature n0 + 4 (lane >> 4) + r]: 16-byte stores.
constexpr int CAM_RT = 4;                                 // 16-row tiles per wave (64 camera rows per blockIdx.z)
__global__ __launch_bounds__(64) void cam_adapt_kernel(ovg_camera_tables_params p) {
  const int lane = threadIdx.x, lr = lane & 15, gq = lane >> 4;
  const int n0 = blockIdx.x * 16, g = blockIdx.y, r0 = blockIdx.z * (16 * CAM_RT);
  const int R = p.B * p.Sc;
  const int64_t K = (int64_t)p.B * p.S;
  const float* wrow = p.adapt_w + ((int64_t)g * OVG_C + n0 + lr) * OVG_C + 4 * gq;
  const float* erow[CAM_RT];
#pragma unroll
  for (int rt = 0; rt < CAM_RT; ++rt) {
    int r = r0 + rt * 16 + lr;
    r = r < R ? r : R - 1;                               // dead rows read a live one (their stores are skipped)
    erow[rt] = p.emb + ((int64_t)g * R + r) * OVG_C + 4 * gq;
  }
  f32x4 acc[CAM_RT];
#pragma unroll
  for (int rt = 0; rt < CAM_RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int s = 0; s < OVG_C / 16; ++s) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(wrow + s * 16);
    f32x4 bfrag[CAM_RT];
#pragma unroll
    for (int rt = 0; rt < CAM_RT; ++rt) bfrag[rt] = *reinterpret_cast<const f32x4*>(erow[rt] + s * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int rt = 0; rt < CAM_RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], bfrag[rt][j], acc[rt], 0, 0, 0);
  }
  const f32x4 bias = *reinterpret_cast<const f32x4*>(p.adapt_b + (int64_t)g * OVG_C + n0 + 4 * gq);
#pragma unroll
  for (int rt = 0; rt < CAM_RT; ++rt) {
    const int r = r0 + rt * 16 + lr;
    if (r < R) {
      const int b = r / p.Sc, view = p.index[r - b * p.Sc];
      if (view < 0 || view >= p.S) continue;               // out-of-range entry: nothing is scattered
      float* dst = p.tables + ((int64_t)g * K + (int64_t)b * p.S + view) * OVG_C + n0 + 4 * gq;
      *reinterpret_cast<f32x4*>(dst) = acc[rt] + bias;
    }
  }
}

bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

extern "C" int ovg_camera_tables(const ovg_camera_tables_params* p, void* stream) {
  if (!p || !p->adapt_b || !p->tables || p->B <= 0 || p->S <= 0 || p->G <= 0 || p->Sc < 0 || p->Sc > 8 * p->S) return OVG_E_ARG;   // Sc > S: a view listed more than once (accepted like the reference; the bound only stops nonsense)
  if (!al16(p->adapt_b) || !al16(p->tables)) return OVG_E_ARG;
  if (p->Sc > 0) {
    if (!p->extrinsics || !p->intrinsics || !p->index || !p->pose_w || !p->pose_b || !p->adapt_w || !p->enc || !p->emb) return OVG_E_ARG;
    if (p->H <= 0 || p->W <= 0 || !al16(p->adapt_w) || !al16(p->emb)) return OVG_E_ARG;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (p->Sc > 0) {
    OVG_LAUNCH(cam_encode_kernel, dim3(p->B), dim3(256), 0, st, *p);
    OVG_CHECK_LAUNCH();
  }
  OVG_LAUNCH(cam_embed_kernel, dim3(OVG_C / 256, p->G), dim3(256), 0, st, *p);
  OVG_CHECK_LAUNCH();
  if (p->Sc > 0) {
    const int R = p->B * p->Sc;
    OVG_LAUNCH(cam_adapt_kernel, dim3(OVG_C / 16, p->G, (R + 16 * CAM_RT - 1) / (16 * CAM_RT)), dim3(64), 0, st, *p);
    OVG_CHECK_LAUNCH();
  }
  return OVG_OK;
}
