"""Torch-tensor front ends of the C ABI entries (one function per `ovg_*` kernel entry).

They only marshal device pointers, shapes and the current HIP stream into the parameter
structs of include/omnivggt_hip.h; every byte of compute happens in libomnivggt_hip.so.
"""
import torch

from . import lib as L

C, H, D, HID = 1024, 16, 64, 4096
KV_TILE = L.KV_TILE


def _stream():
    return torch.cuda.current_stream().cuda_stream


def pad_to(n, m):
    return (n + m - 1) // m * m


def nbytes(t):
    return 0 if t is None else t.numel() * t.element_size()


def _chk_dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.OvgError("expected a HIP device tensor (the hot path has no CPU fallback)")


class HiLo:
    """A split-f16 tensor (L.F32X mode): two f16 planes of one allocation, value ~ hi + lo (include/omnivggt_hip.h, OVG_F16X2)."""
    def __init__(self, planes):
        self.planes, self.hi, self.lo = planes, planes[0], planes[1]      # planes: f16 [2, ...] contiguous

    @property
    def shape(self):
        return self.hi.shape

    @property
    def device(self):
        return self.hi.device

    def stride(self, i):
        return self.hi.stride(i)

    def float(self):
        """f32 value of the pair (tests)."""
        return self.hi.float() + self.lo.float()


def empty_like_dtype(shape, dtype, device, zero=False):
    """Activation / weight storage for `dtype`: a plain tensor, or a HiLo pair of f16 planes in the split-f16 mode."""
    mk = torch.zeros if zero else torch.empty
    if L.is_split(dtype):
        return HiLo(mk((2,) + tuple(shape), device=device, dtype=torch.float16))
    return mk(tuple(shape), device=device, dtype=dtype)


def hi_lo(t):
    """(tensor whose pointer goes into the ordinary field, tensor for the *_lo field or None)."""
    return (t.hi, t.lo) if isinstance(t, HiLo) else (t, None)


def to_hilo(x32):
    """f32 tensor -> HiLo on the same device with the library's rounding (hi = f16(sat(x)), lo = f16(x - hi)); tests / tools."""
    hi = x32.clamp(-65504.0, 65504.0).to(torch.float16)
    lo = (x32 - hi.float()).clamp(-65504.0, 65504.0).to(torch.float16)
    return HiLo(torch.stack([hi, lo]).contiguous())


def layernorm(x, weight, bias, eps, dtype, out=None, out_f32=False):
    """x: f32 [rows, 1024] (row-strided view allowed) -> [rows,1024] in dtype (or f32)."""
    _chk_dev(x, weight, bias)
    rows = x.shape[0]
    if out is None:
        out = torch.empty(rows, C, device=x.device, dtype=torch.float32) if out_f32 else empty_like_dtype((rows, C), dtype, x.device)
    o_hi, o_lo = hi_lo(out)
    p = L.LayerNormParams(L.ptr(x), x.stride(0), L.ptr(o_hi), o_hi.stride(0), L.ptr(weight), L.ptr(bias), rows, eps,
                          L.dtype_code(dtype), 1 if out_f32 else 0, L.ptr(o_lo))
    L.call("ovg_layernorm", p, _stream())
    return out


def linear(x, w, bias, dtype, epilogue=L.EPI_STORE, out=None, out_f32=False, res=None, gamma=None, inject=None,
           inj_period=0, table=None, p0=0, p1=0, row_off=0, out_rows=None, tile=L.TILE_AUTO):
    """y = epilogue(x @ w.T + bias); x [M,K], w [N,K] in dtype."""
    (x, x_lo), (w, w_lo) = hi_lo(x), hi_lo(w)
    _chk_dev(x, w, bias, res, gamma, inject, table)
    M, K = x.shape
    N = w.shape[0]
    f32_out = out_f32 or epilogue in (L.EPI_RES, L.EPI_PATCH)
    if out is None:
        rows = out_rows if out_rows is not None else M
        out = torch.empty(rows, N, device=x.device, dtype=torch.float32) if f32_out else empty_like_dtype((rows, N), dtype, x.device)
    o_hi, o_lo = hi_lo(out)
    _chk_dev(o_hi)
    p = L.LinearParams()
    p.x, p.ldx, p.w, p.ldw, p.bias = L.ptr(x), x.stride(0), L.ptr(w), w.stride(0), L.ptr(bias)
    p.y, p.ldy, p.M, p.N, p.K = L.ptr(o_hi), o_hi.stride(0), M, N, K
    p.x_lo, p.w_lo, p.y_lo = L.ptr(x_lo), L.ptr(w_lo), L.ptr(o_lo)
    p.dtype, p.epilogue, p.out_f32 = L.dtype_code(dtype), epilogue, 1 if f32_out else 0
    if res is not None:
        p.res, p.ldres = L.ptr(res), res.stride(0)
    p.gamma, p.inject, p.inj_period = L.ptr(gamma), L.ptr(inject), inj_period
    p.table, p.p0, p.p1, p.row_off, p.tile = L.ptr(table), p0, p1, row_off, tile
    L.call("ovg_linear", p, _stream())
    return out


def vt_index(n_pad, dtype, device="cpu"):
    """Column permutation of a V^T row: idx[pos] = key stored at column pos. 16-bit dtypes: inside every block of 32 keys
    pos = 8 g + 4 h + i holds key 16 h + 4 g + i (the PV fragment order, csrc/ovg_common.h vt_pos16); f32: identity."""
    pos = torch.arange(n_pad, device=device)
    if dtype is torch.float32:
        return pos
    k = pos & 31
    return (pos & ~31) | (((k >> 2) & 1) << 4) | (((k >> 3) & 3) << 2) | (k & 3)


def set_vt(vt, v_nat):
    """Fill a V^T buffer [BH,64,nk_pad] from natural-order values v_nat [BH,64,n] (n <= nk_pad; the rest is zeroed)."""
    BH, d, n_pad = vt.shape
    full = torch.zeros(BH, d, n_pad, device=vt.device, dtype=vt.dtype)
    full[:, :, : v_nat.shape[2]] = v_nat.to(device=vt.device, dtype=vt.dtype)
    vt.copy_(full[:, :, vt_index(n_pad, vt.dtype, vt.device)])
    return vt


def get_vt(vt):
    """Natural-order view (a copy) of a V^T buffer [BH,64,nk_pad] written by ovg_qkv."""
    idx = vt_index(vt.shape[2], vt.dtype, vt.device)
    out = torch.empty_like(vt)
    out[:, :, idx] = vt
    return out


def alloc_qkv(BH, nq, nk, dtype, device):
    """Zero-filled head-major buffers q [BH,nq_pad,64], k [BH,nk_pad,64], vt [BH,64,nk_pad] (16-bit vt rows hold their keys in
    the vt_index order: fill / read them with set_vt / get_vt when they do not come from ovg_qkv)."""
    nq_pad, nk_pad = pad_to(nq, KV_TILE), pad_to(nk, KV_TILE)
    q = empty_like_dtype((BH, nq_pad, D), dtype, device, zero=True)      # split-f16 mode: HiLo pairs
    k = empty_like_dtype((BH, nk_pad, D), dtype, device, zero=True)
    vt = empty_like_dtype((BH, D, nk_pad), dtype, device, zero=True)
    return q, k, vt


def qkv(x, w, bias, seq, dtype, q, k, vt, qk_norm=None, rope=None, tokens_per_view=1374, grid_w=37, n_special=5,
        q_scale=0.125 * 1.4426950408889634, qk_eps=1e-5, part=0, tile=L.TILE_AUTO):
    """Fused QKV projection.  qk_norm = (qn_w, qn_b, kn_w, kn_b) or None; rope = (cos, sin) or None."""
    (x, x_lo), (w, w_lo), (q, q_lo), (k, k_lo), (vt, vt_lo) = hi_lo(x), hi_lo(w), hi_lo(q), hi_lo(k), hi_lo(vt)
    _chk_dev(x, w, bias, q, k, vt)
    p = L.QkvParams()
    p.x, p.ldx, p.w, p.bias = L.ptr(x), x.stride(0), L.ptr(w), L.ptr(bias)
    p.q, p.k, p.vt = L.ptr(q), L.ptr(k), L.ptr(vt)
    p.x_lo, p.w_lo, p.q_lo, p.k_lo, p.vt_lo = L.ptr(x_lo), L.ptr(w_lo), L.ptr(q_lo), L.ptr(k_lo), L.ptr(vt_lo)
    p.M, p.seq, p.nq_pad, p.nk_pad, p.dtype = x.shape[0], seq, q.shape[1], k.shape[1], L.dtype_code(dtype)
    if qk_norm is not None:
        p.qk_norm = 1
        p.qn_w, p.qn_b, p.kn_w, p.kn_b = (L.ptr(t) for t in qk_norm)
    p.qk_eps = qk_eps
    if rope is not None:
        p.rope = 1
        p.rope_cos, p.rope_sin, p.max_pos = L.ptr(rope[0]), L.ptr(rope[1]), rope[0].shape[0]
    p.tokens_per_view, p.grid_w, p.n_special, p.q_scale = tokens_per_view, grid_w, n_special, q_scale
    p.part, p.tile = part, tile
    L.call("ovg_qkv", p, _stream())


def attn_plan(BH, nq, nks, dtype, variant=0, kv_splits=0, nq_pad=None, cus=0):
    """How ovg_flash_attn would run this shape (host-only query): dict(splits, q_tile, part_bytes, lse_bytes, main_rows, tail_q_tile).
    nq_pad: row count of the q buffer the call will use (default: nq padded to 64); the partial buffers are sized with it."""
    p = L.AttnParams()
    p.nq, p.nq_pad, p.BH, p.nseg = nq, (pad_to(nq, KV_TILE) if nq_pad is None else nq_pad), BH, len(nks)
    for i, nk in enumerate(nks):
        p.seg[i].nk = nk
    p.dtype, p.variant, p.kv_splits, p.cus = L.dtype_code(dtype), variant, kv_splits, cus
    out = L.AttnPlanOut()
    L.check(L.load().ovg_attn_plan(L.C.byref(p), L.C.byref(out)), "ovg_attn_plan")
    return {"splits": out.splits, "q_tile": out.q_tile, "part_bytes": out.part_bytes, "lse_bytes": out.lse_bytes,
            "main_rows": out.main_rows, "tail_q_tile": out.tail_q_tile}


def alloc_split_ws(plan, device):
    """(ws_part, ws_lse) byte/f32 buffers for a plan with splits > 1, else (None, None)."""
    if plan["splits"] <= 1:
        return None, None
    return (torch.empty(plan["part_bytes"], device=device, dtype=torch.uint8),
            torch.empty(plan["lse_bytes"] // 4, device=device, dtype=torch.float32))


def flash_attn(q, segments, nq, dtype, out=None, variant=0, kv_heads=0, head_major=False, lse=None, kv_splits=0, split_ws=None, fallback_count=None, cus=0):
    """q [BH,nq_pad,64]; segments: list of (k [BHkv,nk_pad,64], vt [BHkv,64,nk_pad], nk).
    Returns out [B*nq, 1024] token-major, or with head_major=True out [BH, nq_pad, 64].
    kv_heads > 0: the segments hold kv_heads heads and batch entry bh attends to head bh % kv_heads
    (head-parallel sharding: the BH entries are (source rank, head) pairs).
    lse: optional f32 [BH, nq_pad] receiving log2(sum_k exp2(logit)) over the keys of this call (see attn_merge).
    kv_splits / split_ws = (ws_part, ws_lse) from alloc_split_ws(attn_plan(...)): split-KV for launches that do not fill the
    chip evenly (kv_splits 0 = library decides, and only splits when split_ws is given; 1 = never; 2..8 = force).
    fallback_count: optional int32 device tensor [1]: workgroups of the speculative bf16 kernels that re-ran (telemetry).
    Split-f16 mode (dtype = L.F32X): q / k / vt / out are HiLo pairs."""
    q, q_lo = hi_lo(q)
    _chk_dev(q)
    BH = q.shape[0]
    if out is None:
        out = empty_like_dtype((BH, q.shape[1], D) if head_major else ((BH // H) * nq, C), dtype, q.device)
    ret = out
    out, out_lo = hi_lo(out)
    p = L.AttnParams()
    p.q, p.nq, p.nq_pad, p.nseg = L.ptr(q), nq, q.shape[1], len(segments)
    p.q_lo, p.out_lo = L.ptr(q_lo), L.ptr(out_lo)
    for i, (k, vt, nk) in enumerate(segments):
        (k, k_lo), (vt, vt_lo) = hi_lo(k), hi_lo(vt)
        _chk_dev(k, vt)
        p.seg[i].k, p.seg[i].vt, p.seg[i].nk, p.seg[i].nk_pad = L.ptr(k), L.ptr(vt), nk, k.shape[1]
        p.seg[i].k_lo, p.seg[i].vt_lo = L.ptr(k_lo), L.ptr(vt_lo)
    p.out, p.BH, p.dtype, p.variant, p.kv_heads = L.ptr(out), BH, L.dtype_code(dtype), variant, kv_heads
    if head_major:
        p.ldo, p.out_bh_stride = out.stride(1), out.stride(0)
    else:
        p.ldo = out.stride(0)
    if lse is not None:
        _chk_dev(lse)
        if lse.dtype != torch.float32 or tuple(lse.shape) != (BH, q.shape[1]) or not lse.is_contiguous():
            raise L.OvgError("lse must be a contiguous f32 [BH, nq_pad] tensor")
        p.lse = L.ptr(lse)
    p.kv_splits, p.cus = kv_splits, cus            # cus: CUs the launch plan may count on (0 = all; sharded runs leave some to RCCL)
    if split_ws is not None and split_ws[0] is not None:
        _chk_dev(*split_ws)
        p.ws_part, p.ws_lse = L.ptr(split_ws[0]), L.ptr(split_ws[1])
        p.ws_part_bytes, p.ws_lse_bytes = nbytes(split_ws[0]), nbytes(split_ws[1])
    if fallback_count is not None:
        _chk_dev(fallback_count)
        p.fallback_count = L.ptr(fallback_count)
    L.call("ovg_flash_attn", p, _stream())
    return ret


def attn_merge(a, lse_a, b, lse_b, dtype, out=None):
    """Exact merge of two token-major attention results [n, 1024] over disjoint key sets (B = 1);
    lse_a / lse_b f32 [16, n_pad] from flash_attn(..., lse=...). out may alias a or b."""
    if out is None:
        out = empty_like_dtype(tuple(a.shape), dtype, a.device)
    ret = out
    (a, a_lo), (b, b_lo), (out, out_lo) = hi_lo(a), hi_lo(b), hi_lo(out)
    _chk_dev(a, b, lse_a, lse_b, out)
    p = L.AttnMergeParams(L.ptr(a), a.stride(0), L.ptr(lse_a), L.ptr(b), b.stride(0), L.ptr(lse_b), L.ptr(out), out.stride(0),
                          a.shape[0], lse_a.shape[1], L.dtype_code(dtype), L.ptr(a_lo), L.ptr(b_lo), L.ptr(out_lo))
    L.call("ovg_attn_merge", p, _stream())
    return ret


def pack_weights(src, dtype, k_pad=None):
    """f32 [rows, k] (any trailing dims flattened) -> dtype [rows, k_pad] on the device, zero padded."""
    _chk_dev(src)
    s2 = src.detach().reshape(src.shape[0], -1).float().contiguous()
    rows, k = s2.shape
    k_pad = k if k_pad is None else k_pad
    out = empty_like_dtype((rows, k_pad), dtype, src.device)
    o_hi, o_lo = hi_lo(out)
    p = L.PackWeightsParams(L.ptr(s2), s2.stride(0), L.ptr(o_hi), o_hi.stride(0), rows, k, k_pad, L.dtype_code(dtype), L.ptr(o_lo))
    L.call("ovg_pack_weights", p, _stream())
    return out


def block_workspace_bytes(M, seq, dtype):
    """dict of scratch bytes an ovg_block_forward call with these shapes needs (host-only query)."""
    p = L.BlockParams()
    p.M, p.seq, p.BH = M, seq, (M // seq) * H
    p.nq_pad = p.nk_pad = pad_to(seq, KV_TILE)
    p.dtype = L.dtype_code(dtype)
    ws = L.BlockWorkspace()
    L.check(L.load().ovg_block_workspace_bytes(L.C.byref(p), L.C.byref(ws)), "ovg_block_workspace_bytes")
    return {f: getattr(ws, f) for f, _ in L.BlockWorkspace._fields_}


def heads_to_tokens(x, n, dtype, out=None):
    """x [heads, n_pad, 64] head-major -> out [n, heads*64] token-major (16-bit dtypes)."""
    _chk_dev(x, out)
    heads, n_pad = x.shape[0], x.shape[1]
    if out is None:
        out = torch.empty(n, heads * D, device=x.device, dtype=dtype)
    p = L.HeadsToTokensParams(L.ptr(x), n_pad, L.ptr(out), out.stride(0), n, heads, L.dtype_code(dtype))
    L.call("ovg_heads_to_tokens", p, _stream())
    return out


def im2col_rgb(images, dtype, k_pad=640, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """images f32 [V,3,H,W] -> [V*gh*gw, k_pad] normalised patches."""
    _chk_dev(images)
    V, Cc, Hp, Wp = images.shape
    out = empty_like_dtype((V * (Hp // 14) * (Wp // 14), k_pad), dtype, images.device)
    o_hi, o_lo = hi_lo(out)
    p = L.Im2colParams()
    p.img, p.out, p.k_pad, p.V, p.C, p.Hpx, p.Wpx = L.ptr(images), L.ptr(o_hi), k_pad, V, Cc, Hp, Wp
    p.out_lo = L.ptr(o_lo)
    p.dtype, p.mode = L.dtype_code(dtype), 0
    for i in range(3):
        p.mean[i], p.std[i] = mean[i], std[i]
    L.call("ovg_im2col", p, _stream())
    return out


def depth_stats(depth, mask):
    """depth, mask f32 [B, n] -> f64 [B,2] = {masked sum, count} (deterministic two-stage)."""
    _chk_dev(depth, mask)
    B, n = depth.shape
    nblocks = 256
    stats = torch.empty(B, 2, device=depth.device, dtype=torch.float64)
    partial = torch.empty(B * nblocks * 2, device=depth.device, dtype=torch.float64)
    p = L.DepthStatsParams(L.ptr(depth), L.ptr(mask), B, n, L.ptr(stats), L.ptr(partial), nblocks)
    L.call("ovg_depth_stats", p, _stream())
    return stats


def im2col_depth(depth, mask, stats, views_per_batch, dtype, k_pad=448):
    """depth, mask f32 [V,H,W]; stats f64 [B,2] -> [V*gh*gw, k_pad] (channels: normalised depth, mask)."""
    _chk_dev(depth, mask, stats)
    V, Hp, Wp = depth.shape
    out = empty_like_dtype((V * (Hp // 14) * (Wp // 14), k_pad), dtype, depth.device)
    o_hi, o_lo = hi_lo(out)
    p = L.Im2colParams()
    p.img, p.img2, p.out, p.k_pad, p.V, p.C, p.Hpx, p.Wpx = L.ptr(depth), L.ptr(mask), L.ptr(o_hi), k_pad, V, 2, Hp, Wp
    p.out_lo = L.ptr(o_lo)
    p.dtype, p.mode, p.depth_stats, p.views_per_batch = L.dtype_code(dtype), 1, L.ptr(stats), views_per_batch
    L.call("ovg_im2col", p, _stream())
    return out


def dino_specials(x, V, tokens_per_view, cls, pos0, reg):
    _chk_dev(x, cls, pos0, reg)
    p = L.DinoSpecialsParams(L.ptr(x), x.stride(0), V, tokens_per_view, L.ptr(cls), L.ptr(pos0), L.ptr(reg), reg.shape[0])
    L.call("ovg_dino_specials", p, _stream())


def assemble_tokens(xd, norm_w, norm_b, eps, camera_token, register_token, cam_add, depth_tok, depth_row, placeholder,
                    out, V, S, tokens_per_view=1374, n_special=5, view0=0):
    _chk_dev(xd, out)
    p = L.AssembleParams()
    p.xd, p.ldxd, p.norm_w, p.norm_b, p.eps = L.ptr(xd), xd.stride(0), L.ptr(norm_w), L.ptr(norm_b), eps
    p.camera_token, p.register_token, p.cam_add = L.ptr(camera_token), L.ptr(register_token), L.ptr(cam_add)
    p.depth_tok, p.depth_row, p.placeholder = L.ptr(depth_tok), L.ptr(depth_row), L.ptr(placeholder)
    p.out, p.ldo, p.V, p.S, p.tokens_per_view, p.n_special = L.ptr(out), out.stride(0), V, S, tokens_per_view, n_special
    p.view0 = view0
    L.call("ovg_assemble_tokens", p, _stream())


def probe_mfma(a_frag, b_frag, dtype_code):
    """a_frag, b_frag: int32 [64,4] raw fragments -> f32 [64,4] accumulator of one 16x16 MFMA."""
    out = torch.empty(64, 4, device=a_frag.device, dtype=torch.float32)
    rc = L.load().ovg_probe_mfma(L.ptr(a_frag), L.ptr(b_frag), L.ptr(out), dtype_code, _stream())
    L.check(rc, "ovg_probe_mfma")
    return out


# ---------------------------------------------------------------------------------------------
# DPT head entries (NHWC activations, 16-bit modes)
# ---------------------------------------------------------------------------------------------
def head_layernorm(x, weight, bias, eps, dtype, views, tokens_per_view=1374, n_special=5):
    """x: f32 aggregator output [views*tokens_per_view, 2048] (row stride allowed) -> [views*(tokens_per_view-n_special), 2048]."""
    _chk_dev(x, weight, bias)
    p0 = tokens_per_view - n_special
    out = torch.empty(views * p0, 2048, device=x.device, dtype=dtype)
    p = L.HeadLayerNormParams(L.ptr(x), x.stride(0), L.ptr(out), out.stride(0), L.ptr(weight), L.ptr(bias),
                              views * p0, p0, tokens_per_view, n_special, eps, L.dtype_code(dtype))
    L.call("ovg_head_layernorm", p, _stream())
    return out


def conv(x, w, bias, dtype, cout, ksize=1, stride=1, upshuffle=0, relu=False, add1=None, add2=None, pos=None, out_f32=False):
    """NHWC convolution. x [n,H,W,Cin] dtype (contiguous), w [w_rows, k*k*Cin] dtype (taps-major), bias f32 [cout] or None.
    upshuffle = s: ConvTranspose2d(kernel = stride = s) -> [n, H*s, W*s, cout]. pos = (pos_x [OW,cout/2], pos_y [OH,cout/2])."""
    _chk_dev(x, w, bias, add1, add2)
    n, H, W, cin = x.shape
    pad = ksize // 2
    OH, OW = (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1
    s = upshuffle if upshuffle > 1 else 1
    out = torch.empty(n, OH * s, OW * s, cout, device=x.device, dtype=torch.float32 if out_f32 else dtype)
    p = L.ConvParams()
    p.x, p.ldx, p.w, p.bias, p.y, p.ldy = L.ptr(x), x.stride(2), L.ptr(w), L.ptr(bias), L.ptr(out), out.stride(2)
    if add1 is not None:
        p.add1, p.ld1 = L.ptr(add1), add1.stride(2)
    if add2 is not None:
        p.add2, p.ld2 = L.ptr(add2), add2.stride(2)
    if pos is not None:
        p.pos_x, p.pos_y = L.ptr(pos[0]), L.ptr(pos[1])
    p.n_img, p.H, p.W, p.Cin, p.Cout, p.w_rows = n, H, W, cin, cout, w.shape[0]
    p.ksize, p.stride, p.upshuffle, p.relu, p.out_f32, p.dtype = ksize, stride, upshuffle, 1 if relu else 0, 1 if out_f32 else 0, L.dtype_code(dtype)
    L.call("ovg_conv", p, _stream())
    return out


def upsample(x, OH, OW, dtype, pos=None):
    """Bilinear align_corners=True resize of NHWC x [n,H,W,C] -> [n,OH,OW,C] (+ UV position embedding tables)."""
    _chk_dev(x)
    n, H, W, c = x.shape
    out = torch.empty(n, OH, OW, c, device=x.device, dtype=dtype)
    p = L.UpsampleParams()
    p.x, p.ldx, p.y, p.ldy = L.ptr(x), x.stride(2), L.ptr(out), out.stride(2)
    if pos is not None:
        p.pos_x, p.pos_y = L.ptr(pos[0]), L.ptr(pos[1])
    p.n_img, p.H, p.W, p.OH, p.OW, p.C, p.dtype = n, H, W, OH, OW, c, L.dtype_code(dtype)
    L.call("ovg_upsample", p, _stream())
    return out


def dpt_tail_supported(x, dtype, OH=None, OW=None):
    """The one-launch output stage (ovg_dpt_tail) exists for the plain 16-bit dtypes and the model's 128-channel map. Mirrors every
    host-side limit of the C entry (csrc/ovg_head.hip: ovg_dpt_tail) that answers OVG_E_UNSUPPORTED / OVG_E_ARG for a shape the three-launch
    form still serves -- one source image within 32-bit element offsets, an output of at least 2 x 2 pixels, 16-byte aligned rows --
    so that callers fall back to upsample -> conv -> dpt_out instead of raising (round-5 advisor)."""
    if dtype not in (torch.bfloat16, torch.float16) or x.dim() != 4 or x.shape[-1] != 128:
        return False
    n, H, W, c = x.shape
    ldx = x.stride(2)
    if H * W * ldx >= (1 << 31) or ldx % 8 or (x.data_ptr() & 15):
        return False
    if OH is not None and (OH <= 1 or OW <= 1):
        return False
    return True


def dpt_tail(x, OH, OW, dtype, pos, w1, b1, w2, b2, activation):
    """x [n,H,W,128] dtype -> upsample to (OH, OW) + pos -> conv3x3(128->32)+ReLU -> conv1x1 -> activation: (val [n,OH,OW,od-1], conf [n,OH,OW]).
    w1 [>=32, 9*128] dtype taps-major (the zero-padded matrix of the ovg_conv form is fine)."""
    _chk_dev(x, w1, b1, w2, b2)
    n, H, W, c = x.shape
    od = w2.shape[0]
    val = torch.empty(n, OH, OW, od - 1, device=x.device, dtype=torch.float32)
    conf = torch.empty(n, OH, OW, device=x.device, dtype=torch.float32)
    p = L.DptTailParams()
    p.x, p.ldx, p.w1, p.ldw1, p.b1, p.w2, p.b2, p.val, p.conf = L.ptr(x), x.stride(2), L.ptr(w1), w1.stride(0), L.ptr(b1), L.ptr(w2), L.ptr(b2), L.ptr(val), L.ptr(conf)
    if pos is not None:
        p.pos_x, p.pos_y = L.ptr(pos[0]), L.ptr(pos[1])
    p.n_img, p.H, p.W, p.OH, p.OW, p.C, p.out_dim, p.activation, p.dtype = n, H, W, OH, OW, c, od, 0 if activation == "exp" else 1, L.dtype_code(dtype)
    L.call("ovg_dpt_tail", p, _stream())
    return val, conf


def dpt_out(h, w2, b2, activation):
    """h f32 [n,H,W,32] (post-ReLU) -> (val [n,H,W,out_dim-1], conf [n,H,W]); activation 'exp' | 'inv_log'."""
    _chk_dev(h, w2, b2)
    n, H, W, _ = h.shape
    od = w2.shape[0]
    val = torch.empty(n, H, W, od - 1, device=h.device, dtype=torch.float32)
    conf = torch.empty(n, H, W, device=h.device, dtype=torch.float32)
    p = L.DptOutParams(L.ptr(h), L.ptr(w2), L.ptr(b2), L.ptr(val), L.ptr(conf), n * H * W, od, 0 if activation == "exp" else 1)
    L.call("ovg_dpt_out", p, _stream())
    return val, conf


def camera_head_workspace_bytes(S, dtype):
    n = L.load().ovg_camera_head_workspace_bytes(S, L.dtype_code(dtype))
    if n < 0:
        raise L.OvgError("ovg_camera_head_workspace_bytes: unsupported (S=%d, dtype=%s)" % (S, dtype))
    return int(n)


def camera_head(tokens, W, dtype, iters=4, ws=None):
    """Whole CameraHead.forward (camera_head.py:84-154) of one batch element in one call.
    tokens: f32 [S, 2048] view of the camera tokens (row stride allowed, e.g. out[-1][b, :, 0]); W: packed weights
    (heads_hip.HipCameraHead._pack: GEMM matrices in `dtype`, everything else f32); -> [iters, S, 9] f32."""
    _chk_dev(tokens, ws)
    S = tokens.shape[0]
    if tokens.dtype != torch.float32 or tokens.shape[1] != 2048 or tokens.stride(1) != 1:
        raise L.OvgError("camera_head: tokens must be f32 [S, 2048] with unit inner stride")
    need = camera_head_workspace_bytes(S, dtype)
    if ws is None or ws.numel() * ws.element_size() < need:
        ws = torch.empty(need, device=tokens.device, dtype=torch.uint8)
    out = torch.empty(iters, S, 9, device=tokens.device, dtype=torch.float32)
    p = L.CameraHeadParams()
    p.tokens, p.ld_tokens, p.S, p.iters, p.dtype = L.ptr(tokens), tokens.stride(0), S, iters, L.dtype_code(dtype)
    p.trunk_depth, p.dim, p.heads = len(W["blocks"]), 2048, W["heads"]
    for name in ("token_norm_w", "token_norm_b", "trunk_norm_w", "trunk_norm_b", "empty_pose", "embed_w", "embed_b", "mod_w", "mod_b",
                 "pb1_w", "pb1_b", "pb2_w", "pb2_b"):
        _chk_dev(W[name])
        setattr(p, name, L.ptr(W[name]))
    for i, blk in enumerate(W["blocks"]):
        for name, _ in L.CameraBlockWeights._fields_:
            _chk_dev(blk[name])
            setattr(p.blk[i], name, L.ptr(blk[name]))
    p.ws, p.ws_bytes, p.out = L.ptr(ws), ws.numel() * ws.element_size(), L.ptr(out)
    L.call("ovg_camera_head", p, _stream())
    return out


def camera_tables(extrinsics, intrinsics, index, S, hw, pose_w, pose_b, adapt_w, adapt_b, out=None):
    """Camera-modality injection tables [G, B*S, 1024] f32 built on the device in <= 3 launches, no host round trip
    (ovg_camera_tables). extrinsics [B,S,3,4] / intrinsics [B,S,3,3] f32 device tensors (ignored when index is None);
    index: int32 DEVICE tensor [Sc] of the views that carry a GT camera, or None; pose_w [G*1024, 9], pose_b [G*1024],
    adapt_w [G,1024,1024], adapt_b [G,1024] f32."""
    _chk_dev(pose_w, pose_b, adapt_w, adapt_b, index, out)
    G = adapt_b.shape[0]
    Sc = 0 if index is None else int(index.numel())
    B = 1 if Sc == 0 else extrinsics.shape[0]
    dev = adapt_b.device
    if out is None:
        out = torch.empty(G, B * S, C, device=dev, dtype=torch.float32)
    p = L.CameraTablesParams()
    p.B, p.S, p.Sc, p.H, p.W, p.G = B, S, Sc, int(hw[0]), int(hw[1]), G
    p.pose_w, p.pose_b, p.adapt_w, p.adapt_b, p.tables = L.ptr(pose_w), L.ptr(pose_b), L.ptr(adapt_w), L.ptr(adapt_b), L.ptr(out)
    keep = None
    if Sc:
        _chk_dev(extrinsics, intrinsics)
        if index.dtype != torch.int32 or not index.is_contiguous():
            raise L.OvgError("camera_tables: index must be a contiguous int32 device tensor")
        ext = extrinsics.detach().to(torch.float32).contiguous()
        intr = intrinsics.detach().to(torch.float32).contiguous()
        if tuple(ext.shape[1:]) != (S, 3, 4) or tuple(intr.shape) != (B, S, 3, 3):
            raise L.OvgError("camera_tables: extrinsics must be [B,S,3,4] and intrinsics [B,S,3,3]")
        enc = torch.empty(B * Sc, 9, device=dev, dtype=torch.float32)
        emb = torch.empty(G, B * Sc, C, device=dev, dtype=torch.float32)
        keep = (ext, intr, enc, emb)
        p.extrinsics, p.intrinsics, p.index, p.enc, p.emb = L.ptr(ext), L.ptr(intr), L.ptr(index), L.ptr(enc), L.ptr(emb)
    L.call("ovg_camera_tables", p, _stream())
    del keep          # the caching allocator keeps freed blocks valid for work already queued on this stream
    return out


def unproject(depth, cam):
    """depth f32 [S,H,W], cam f32 [S,16] (cam-to-world R row-major, t, fu, fv, cu, cv) -> world points [S,H,W,3] f32."""
    _chk_dev(depth, cam)
    S, H, W = depth.shape
    out = torch.empty(S, H, W, 3, device=depth.device, dtype=torch.float32)
    p = L.UnprojectParams(L.ptr(depth), L.ptr(cam), L.ptr(out), S, H, W)
    L.call("ovg_unproject", p, _stream())
    return out
