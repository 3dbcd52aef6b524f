// One pre-LN transformer block of the aggregator as a fixed launch sequence
// (reference: omnivggt/layers/block.py:81-107 eval path, attention.py:50-77, mlp.py:34-40):
//   LN1 -> QKV(+q/k-norm+RoPE) -> flash attention -> proj(+LayerScale+residual)
//   -> LN2 -> fc1(+GELU) -> fc2(+LayerScale+residual[+camera injection])
// One C call per block keeps the Python driver off the critical path (7 launches/call).
#include "ovg_common.h"

namespace {

int check_block(const ovg_block_params* p) {
  if (!p || !p->x_in || !p->x_out || !p->ws_xn || !p->ws_q || !p->ws_k || !p->ws_vt || !p->ws_attn || !p->ws_hid) return OVG_E_ARG;
  if (p->M <= 0 || p->seq <= 0 || p->M % p->seq || p->BH != (p->M / p->seq) * OVG_H) return OVG_E_ARG;
  if (p->nseg_extra < 0 || p->nseg_extra >= OVG_MAX_SEG) return OVG_E_ARG;
  if (p->local_seg_index < 0 || p->local_seg_index > p->nseg_extra) return OVG_E_ARG;
  if (p->dtype == OVG_F16X2) {      // split-f16: every 16-bit tensor has its lo plane
    if (!p->ws_xn_lo || !p->ws_q_lo || !p->ws_k_lo || !p->ws_vt_lo || !p->ws_attn_lo || !p->ws_hid_lo) return OVG_E_ARG;
    if (!p->w.qkv_w_lo || !p->w.proj_w_lo || !p->w.fc1_w_lo || !p->w.fc2_w_lo) return OVG_E_ARG;
  }
  return OVG_OK;
}

int run_prologue(const ovg_block_params* p, void* st, int part) {
  int rc = OVG_OK;
  if (part != 2) {
  ovg_layernorm_params ln{};
  ln.x = p->x_in; ln.ldx = p->ld_in; ln.y = p->ws_xn; ln.ldy = OVG_C;
  ln.weight = static_cast<const float*>(p->w.n1_w); ln.bias = static_cast<const float*>(p->w.n1_b);
  ln.rows = p->M; ln.eps = p->ln_eps; ln.dtype = p->dtype; ln.out_f32 = 0; ln.y_lo = p->ws_xn_lo;
  rc = ovg_layernorm(&ln, st);
  if (rc) return rc;
  }

  ovg_qkv_params q{};
  q.part = part;
  q.x = p->ws_xn; q.ldx = OVG_C; q.w = p->w.qkv_w; q.bias = p->w.qkv_b;
  q.q = p->ws_q; q.k = p->ws_k; q.vt = p->ws_vt;
  q.x_lo = p->ws_xn_lo; q.w_lo = p->w.qkv_w_lo; q.q_lo = p->ws_q_lo; q.k_lo = p->ws_k_lo; q.vt_lo = p->ws_vt_lo;
  q.M = p->M; q.seq = p->seq; q.nq_pad = p->nq_pad; q.nk_pad = p->nk_pad; q.dtype = p->dtype;
  q.qk_norm = p->qk_norm; q.qn_w = p->w.qn_w; q.qn_b = p->w.qn_b; q.kn_w = p->w.kn_w; q.kn_b = p->w.kn_b; q.qk_eps = p->qk_eps;
  q.rope = p->rope; q.rope_cos = p->rope_cos; q.rope_sin = p->rope_sin; q.max_pos = p->max_pos;
  q.tokens_per_view = p->tokens_per_view; q.grid_w = p->grid_w; q.n_special = p->n_special;
  q.tile = p->gemm_tile;
  q.q_scale = 0.125f * 1.4426950408889634f;   // head_dim^-0.5 * log2(e): attention uses exp2
  return ovg_qkv(&q, st);
}

int run_epilogue(const ovg_block_params* p, void* st) {
  int rc = OVG_OK;
  if (!p->skip_attention) {
  ovg_attn_params a{};
  a.q = p->ws_q; a.nq = p->seq; a.nq_pad = p->nq_pad;
  a.nseg = 1 + p->nseg_extra;
  int e = 0;
  for (int i = 0; i < a.nseg; ++i) {
    if (i == p->local_seg_index) { a.seg[i].k = p->ws_k; a.seg[i].vt = p->ws_vt; a.seg[i].nk = p->seq; a.seg[i].nk_pad = p->nk_pad;
                                   a.seg[i].k_lo = p->ws_k_lo; a.seg[i].vt_lo = p->ws_vt_lo; }
    else a.seg[i] = p->extra[e++];
  }
  a.out = p->ws_attn; a.ldo = OVG_C; a.BH = p->BH; a.dtype = p->dtype; a.variant = p->attn_variant;
  a.kv_splits = p->attn_kv_splits; a.ws_part = p->ws_attn_part; a.ws_lse = p->ws_attn_lse;
  a.ws_part_bytes = p->ws_attn_part_bytes; a.ws_lse_bytes = p->ws_attn_lse_bytes;
  a.fallback_count = p->attn_fallback_count;
  a.cus = p->attn_cus;
  a.q_lo = p->ws_q_lo; a.out_lo = p->ws_attn_lo;
  if (p->ev_attn_start) (void)hipEventRecord(static_cast<hipEvent_t>(p->ev_attn_start), static_cast<hipStream_t>(st));
  rc = ovg_flash_attn(&a, st);
  if (p->ev_attn_stop) (void)hipEventRecord(static_cast<hipEvent_t>(p->ev_attn_stop), static_cast<hipStream_t>(st));
  if (rc) return rc;
  }

  ovg_linear_params l{};
  l.x = p->ws_attn; l.ldx = OVG_C; l.w = p->w.proj_w; l.ldw = OVG_C; l.bias = p->w.proj_b;
  l.y = p->x_out; l.ldy = p->ld_out; l.M = p->M; l.N = OVG_C; l.K = OVG_C; l.dtype = p->dtype;
  l.tile = p->gemm_tile; l.x_lo = p->ws_attn_lo; l.w_lo = p->w.proj_w_lo;
  l.epilogue = OVG_EPI_RES; l.out_f32 = 1; l.res = p->x_in; l.ldres = p->ld_in; l.gamma = p->w.ls1;
  rc = ovg_linear(&l, st);
  if (rc) return rc;

  ovg_layernorm_params ln{};
  ln.x = p->x_out; ln.ldx = p->ld_out; ln.y = p->ws_xn; ln.ldy = OVG_C;
  ln.weight = static_cast<const float*>(p->w.n2_w); ln.bias = static_cast<const float*>(p->w.n2_b);
  ln.rows = p->M; ln.eps = p->ln_eps; ln.dtype = p->dtype; ln.out_f32 = 0; ln.y_lo = p->ws_xn_lo;
  rc = ovg_layernorm(&ln, st);
  if (rc) return rc;

  ovg_linear_params f1{};
  f1.x = p->ws_xn; f1.ldx = OVG_C; f1.w = p->w.fc1_w; f1.ldw = OVG_C; f1.bias = p->w.fc1_b;
  f1.y = p->ws_hid; f1.ldy = OVG_HID; f1.M = p->M; f1.N = OVG_HID; f1.K = OVG_C; f1.dtype = p->dtype;
  f1.epilogue = OVG_EPI_GELU; f1.out_f32 = 0; f1.tile = p->gemm_tile;
  f1.x_lo = p->ws_xn_lo; f1.w_lo = p->w.fc1_w_lo; f1.y_lo = p->ws_hid_lo;
  rc = ovg_linear(&f1, st);
  if (rc) return rc;

  ovg_linear_params f2{};
  f2.x = p->ws_hid; f2.ldx = OVG_HID; f2.w = p->w.fc2_w; f2.ldw = OVG_HID; f2.bias = p->w.fc2_b;
  f2.y = p->x_out; f2.ldy = p->ld_out; f2.M = p->M; f2.N = OVG_C; f2.K = OVG_HID; f2.dtype = p->dtype;
  f2.epilogue = OVG_EPI_RES; f2.out_f32 = 1; f2.res = p->x_out; f2.ldres = p->ld_out; f2.gamma = p->w.ls2;
  f2.inject = p->inject; f2.inj_period = p->inj_period; f2.tile = p->gemm_tile;
  f2.x_lo = p->ws_hid_lo; f2.w_lo = p->w.fc2_w_lo;
  return ovg_linear(&f2, st);
}

}  // namespace

extern "C" int ovg_block_workspace_bytes(const ovg_block_params* p, ovg_block_workspace* out) {
  if (!p || !out || p->M <= 0 || p->seq <= 0 || p->M % p->seq || p->BH != (p->M / p->seq) * OVG_H) return OVG_E_ARG;
  if (p->nq_pad < p->seq || p->nk_pad < p->seq || p->nk_pad % OVG_KV_TILE) return OVG_E_ARG;
  if (p->dtype != OVG_BF16 && p->dtype != OVG_F16 && p->dtype != OVG_F32 && p->dtype != OVG_F16X2) return OVG_E_DTYPE;
  const int64_t e = p->dtype == OVG_F32 ? 4 : 2;      // OVG_F16X2: bytes of ONE plane of each tensor (the caller provides hi and lo)
  out->xn = p->M * OVG_C * e;
  out->attn = p->M * OVG_C * e;
  out->hid = p->M * OVG_HID * e;
  out->q = p->BH * p->nq_pad * OVG_D * e;
  out->k = p->BH * p->nk_pad * OVG_D * e;
  out->vt = p->BH * p->nk_pad * OVG_D * e;
  out->total = out->xn + out->attn + out->hid + out->q + out->k + out->vt;
  return OVG_OK;
}

extern "C" int ovg_block_attn_prologue(const ovg_block_params* p, void* stream) {
  int rc = check_block(p);
  return rc ? rc : run_prologue(p, stream, p->qkv_part);
}
extern "C" int ovg_block_attn_epilogue(const ovg_block_params* p, void* stream) {
  int rc = check_block(p);
  return rc ? rc : run_epilogue(p, stream);
}
extern "C" int ovg_block_forward(const ovg_block_params* p, void* stream) {
  int rc = check_block(p);
  if (rc) return rc;
  rc = run_prologue(p, stream, 0);
  return rc ? rc : run_epilogue(p, stream);
}
