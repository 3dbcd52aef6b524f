"""CPU oracle for the OmniVGGT aggregator hot path -- TEST INFRASTRUCTURE ONLY.

A functional restatement (plain PyTorch CPU fp32 ops over a state-dict, no nn.Module)
of the reference algorithm.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this file; the product path (omnivggt-official_amd/) never
does and has no CPU fallback.

Parity pinning: the reference ships no golden vectors or tests, so this
restatement is pinned against the reference itself, executed in the build container by
oracle/gen_golden.py (which imports /root/reference through oracle/ref_shim.py): the
restatement must agree with the reference on the same seeded weights/inputs, and the
sampled reference outputs are committed under tests/golden/ for the GPU box.

Third-party arithmetic: all math is ATen (torch 2.10.0 here; the reference pins
torch==2.7.0, README.md:41): F.scaled_dot_product_attention, F.linear, F.layer_norm,
F.gelu (erf), F.conv2d, F.embedding semantics are the documented ones.

Every function cites the reference file:line it follows (paths relative to
/root/reference/omnivggt/).
"""
import math

import torch
import torch.nn.functional as F

RESNET_MEAN = (0.485, 0.456, 0.406)   # models/aggregator.py:22
RESNET_STD = (0.229, 0.224, 0.225)    # models/aggregator.py:23
PATCH = 14
N_SPECIAL = 5                          # camera + 4 register tokens, models/aggregator.py:133
HEADS = 16
AA_LN_EPS = 1e-5                       # nn.LayerNorm default, layers/block.py:50,67
DINO_LN_EPS = 1e-6                     # layers/vision_transformer.py:94


# ----------------------------------------------------------------------------
# layers
# ----------------------------------------------------------------------------
def layer_norm(x, sd, prefix, eps):
    """nn.LayerNorm over the last dim (layers/block.py:50,67; attention.py:43-44)."""
    w = sd[prefix + ".weight"]
    return F.layer_norm(x, (w.numel(),), w, sd[prefix + ".bias"], eps)


def rope_tables(max_pos, half_dim=32, base=100.0):
    """cos/sin tables of layers/rope.py:86-117 for one spatial axis.

    exponents = arange(0, half_dim, 2)/half_dim; angles = pos * base**-exponents,
    duplicated (cat) to half_dim columns.  Returns (cos, sin) of shape [max_pos, half_dim].
    """
    exponents = torch.arange(0, half_dim, 2).float() / half_dim
    inv_freq = 1.0 / (base ** exponents)
    positions = torch.arange(max_pos, dtype=inv_freq.dtype)
    angles = torch.einsum("i,j->ij", positions, inv_freq)
    angles = torch.cat((angles, angles), dim=-1)
    return angles.cos(), angles.sin()


def rope_2d(t, pos, cos, sin):
    """layers/rope.py:154-188: first half of head_dim rotated by y, second half by x;
    rotate-half pairing (j, j+16) inside each half (rope.py:120-131)."""
    def one_axis(x, p):
        c = F.embedding(p, cos)[:, None, :, :]
        s = F.embedding(p, sin)[:, None, :, :]
        h = x.shape[-1] // 2
        rot = torch.cat((-x[..., h:], x[..., :h]), dim=-1)
        return x * c + rot * s
    v, h = t.chunk(2, dim=-1)
    return torch.cat((one_axis(v, pos[..., 0]), one_axis(h, pos[..., 1])), dim=-1)


def attention(x, sd, prefix, pos, rope, qk_norm):
    """layers/attention.py:50-77 (fused_attn=True path)."""
    B, N, C = x.shape
    qkv = F.linear(x, sd[prefix + ".qkv.weight"], sd[prefix + ".qkv.bias"])
    qkv = qkv.reshape(B, N, 3, HEADS, C // HEADS).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    if qk_norm:
        q = layer_norm(q, sd, prefix + ".q_norm", AA_LN_EPS)
        k = layer_norm(k, sd, prefix + ".k_norm", AA_LN_EPS)
    if rope is not None:
        q = rope_2d(q, pos, *rope)
        k = rope_2d(k, pos, *rope)
    o = F.scaled_dot_product_attention(q, k, v)
    o = o.transpose(1, 2).reshape(B, N, C)
    return F.linear(o, sd[prefix + ".proj.weight"], sd[prefix + ".proj.bias"])


def mlp(x, sd, prefix):
    """layers/mlp.py:34-40 with nn.GELU() (exact erf), mlp.py:22."""
    h = F.gelu(F.linear(x, sd[prefix + ".fc1.weight"], sd[prefix + ".fc1.bias"]))
    return F.linear(h, sd[prefix + ".fc2.weight"], sd[prefix + ".fc2.bias"])


def block(x, sd, prefix, pos=None, rope=None, qk_norm=False, eps=AA_LN_EPS):
    """layers/block.py:105-106 (eval path) with LayerScale layers/layer_scale.py:26-27."""
    a = attention(layer_norm(x, sd, prefix + ".norm1", eps), sd, prefix + ".attn", pos, rope, qk_norm)
    x = x + a * sd[prefix + ".ls1.gamma"]
    m = mlp(layer_norm(x, sd, prefix + ".norm2", eps), sd, prefix + ".mlp")
    return x + m * sd[prefix + ".ls2.gamma"]


def patch_conv(x, sd, prefix):
    """layers/patch_embed.py:68-81: Conv2d(k=14,s=14) -> flatten -> transpose."""
    y = F.conv2d(x, sd[prefix + ".proj.weight"], sd[prefix + ".proj.bias"], stride=PATCH)
    return y.flatten(2).transpose(1, 2)


def interpolate_pos_encoding(pos_embed, npatch, h_px, w_px):
    """layers/vision_transformer.py:180-212 as the aggregator configures it (aggregator.py:156-157:
    interpolate_antialias=True, interpolate_offset=0.0 -> `size=` form): identity when the patch grid is the
    trained square one, else bicubic + antialias resampling of the patch rows to (h_px/14, w_px/14)."""
    N = pos_embed.shape[1] - 1
    if npatch == N and h_px == w_px:
        return pos_embed
    pe = pos_embed.float()
    class_pos, patch_pos = pe[:, 0], pe[:, 1:]
    dim = pe.shape[-1]
    g0, g1 = h_px // PATCH, w_px // PATCH
    M = int(math.sqrt(N))
    assert N == M * M
    patch_pos = F.interpolate(patch_pos.reshape(1, M, M, dim).permute(0, 3, 1, 2), mode="bicubic", antialias=True, size=(g0, g1))
    assert (g0, g1) == tuple(patch_pos.shape[-2:])
    patch_pos = patch_pos.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((class_pos.unsqueeze(0), patch_pos), dim=1).to(pos_embed.dtype)


def dino_backbone(images_norm, sd, prefix="aggregator.patch_embed", depth=24, return_prenorm=False):
    """layers/vision_transformer.py:214-226,252-271."""
    h_px, w_px = images_norm.shape[-2:]
    x = patch_conv(images_norm, sd, prefix + ".patch_embed")
    V = x.shape[0]
    x = torch.cat((sd[prefix + ".cls_token"].expand(V, -1, -1), x), dim=1)
    x = x + interpolate_pos_encoding(sd[prefix + ".pos_embed"], x.shape[1] - 1, h_px, w_px)
    x = torch.cat((x[:, :1], sd[prefix + ".register_tokens"].expand(V, -1, -1), x[:, 1:]), dim=1)
    for i in range(depth):
        x = block(x, sd, "%s.blocks.%d" % (prefix, i), eps=DINO_LN_EPS)
    if return_prenorm:
        return x
    xn = layer_norm(x, sd, prefix + ".norm", DINO_LN_EPS)
    return xn[:, N_SPECIAL:]


# ----------------------------------------------------------------------------
# camera / depth modality preparation
# ----------------------------------------------------------------------------
def quat_from_matrix(R):
    """utils/rotation.py:47-109 (+ standardize :126-138): xyzw, real part non-negative."""
    m = R.reshape(R.shape[:-2] + (9,))
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = m.unbind(-1)
    raw = torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=-1)
    q_abs = torch.where(raw > 0, torch.sqrt(raw.clamp(min=0)), torch.zeros_like(raw))
    cand = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1),
    ], dim=-2)
    cand = cand / (2.0 * q_abs[..., None].clamp(min=0.1))
    best = q_abs.argmax(dim=-1)
    out = torch.gather(cand, -2, best[..., None, None].expand(best.shape + (1, 4))).squeeze(-2)
    out = out[..., [1, 2, 3, 0]]
    return torch.where(out[..., 3:4] < 0, -out, out)


def pose_encoding(extrinsics, intrinsics, hw):
    """utils/pose_enc.py:48-59: [t(3), quat xyzw(4), fov_h, fov_w]."""
    H, W = hw
    R, T = extrinsics[..., :3, :3], extrinsics[..., :3, 3]
    fov_h = 2 * torch.atan((H / 2) / intrinsics[..., 1, 1])
    fov_w = 2 * torch.atan((W / 2) / intrinsics[..., 0, 0])
    return torch.cat([T, quat_from_matrix(R), fov_h[..., None], fov_w[..., None]], dim=-1).float()


def normalize_extrinsics(ext):
    """models/omnivggt_aggregator.py:85-105: first selected camera -> identity, translations
    divided by the mean distance of the other cameras (clamp 1e-6)."""
    B, S = ext.shape[:2]
    bottom = torch.zeros(B, S, 1, 4)
    bottom[..., 3] = 1.0
    homog = torch.cat([ext, bottom], dim=-2)
    R0, t0 = homog[:, 0, :3, :3], homog[:, 0, :3, 3:]
    inv0 = torch.eye(4).repeat(B, 1, 1)                       # utils/geometry.py:269-318
    inv0[:, :3, :3] = R0.transpose(1, 2)
    inv0[:, :3, 3:] = -torch.bmm(R0.transpose(1, 2), t0)
    new = torch.matmul(homog, inv0.unsqueeze(1))
    if S > 1:
        centers = new[:, :, :3, 3]
        dist = torch.norm(centers - centers[:, 0:1], dim=-1)[:, 1:]
        scale = dist.mean(dim=1, keepdim=True).clamp(min=1e-6)
        new[:, :, :3, 3] = new[:, :, :3, 3] / scale.unsqueeze(-1)
    return new[:, :, :3]


def normalize_depth(depth, mask, eps=1e-8):
    """models/omnivggt_aggregator.py:107-128: per batch masked mean over ALL selected views."""
    d = depth.squeeze(-1)
    out = torch.zeros_like(d)
    for b in range(d.shape[0]):
        valid = d[b][mask[b] > 0]
        if valid.numel() == 0:
            continue
        out[b] = d[b] / (valid.mean() + eps) * mask[b]
    return out.unsqueeze(-1)


def special_tokens(tok, B, S):
    """models/aggregator.py:343-366: slot 0 for the first view of each batch, slot 1 otherwise."""
    first = tok[:, 0:1].expand(B, 1, *tok.shape[2:])
    rest = tok[:, 1:].expand(B, S - 1, *tok.shape[2:])
    return torch.cat([first, rest], dim=1).reshape(B * S, *tok.shape[2:])


def scatter_rows(values, B, S, index, K, width):
    """zero (K,1,C) tensor with rows b*S+idx filled (omnivggt_aggregator.py:174-178,278-282)."""
    full = torch.zeros(K, 1, width)
    rows = (torch.arange(B).unsqueeze(1) * S + torch.tensor(index).unsqueeze(0)).reshape(-1)
    full[rows] = values.reshape(-1, 1, width)
    return full


# ----------------------------------------------------------------------------
# the aggregator
# ----------------------------------------------------------------------------
def aggregator_prepare(sd, images, extrinsics, intrinsics, depth, mask, depth_gt_index, camera_gt_index, dino_layers=24):
    """Embedding + modality fusion, models/omnivggt_aggregator.py:130-224.
    Returns dict(tokens (K,T,C), pos (K,T,2), rope, enc, B, S)."""
    P = "aggregator"
    B, S, C_in, H, W = images.shape
    if C_in != 3:
        raise ValueError("Expected 3 input channels, got %d" % C_in)
    mean = torch.tensor(RESNET_MEAN).view(1, 1, 3, 1, 1)
    std = torch.tensor(RESNET_STD).view(1, 1, 3, 1, 1)
    x = ((images - mean) / std).view(B * S, C_in, H, W)
    patch_tokens = dino_backbone(x, sd, P + ".patch_embed", dino_layers)
    K, P0, C = patch_tokens.shape

    cam_tok = special_tokens(sd[P + ".camera_token"], B, S)
    reg_tok = special_tokens(sd[P + ".register_token"], B, S)

    if len(camera_gt_index) != 0:
        idx = torch.tensor(camera_gt_index)
        ext_n = normalize_extrinsics(torch.index_select(extrinsics, 1, idx))
        enc = pose_encoding(ext_n, torch.index_select(intrinsics, 1, idx), (H, W))
    else:
        enc = None

    if len(depth_gt_index) != 0:
        idx = torch.tensor(depth_gt_index)
        d_sel = torch.index_select(depth, 1, idx)
        m_sel = torch.index_select(mask, 1, idx)
        d_norm = normalize_depth(d_sel, m_sel)
        n = len(depth_gt_index)
        maps = torch.cat([d_norm.view(B * n, 1, H, W), m_sel.reshape(B * n, 1, H, W)], dim=1)
        d_tok = patch_conv(maps, sd, P + ".depth_patch_embed")
        depth_full = sd[P + ".depth_placeholder"].expand(K, P0, C).clone()
        rows = (torch.arange(B).unsqueeze(1) * S + idx.unsqueeze(0)).reshape(-1)
        depth_full[rows] = d_tok
    else:
        depth_full = sd[P + ".depth_placeholder"].expand(K, P0, C)

    st = dict(B=B, S=S, K=K, C=C, enc=enc, camera_gt_index=list(camera_gt_index))
    cam_tok = cam_tok + camera_injection(sd, 0, st)
    st["tokens"] = torch.cat([cam_tok, reg_tok, patch_tokens + depth_full], dim=1)

    gh, gw = H // PATCH, W // PATCH
    yx = torch.cartesian_prod(torch.arange(gh), torch.arange(gw)) + 1        # layers/rope.py:39-59 (+1: :219)
    pos = torch.cat([torch.zeros(N_SPECIAL, 2, dtype=yx.dtype), yx], dim=0)
    st["pos"] = pos.unsqueeze(0).expand(K, -1, -1)
    st["rope"] = rope_tables(int(pos.max()) + 1)
    return st


def camera_injection(sd, i, st):
    """camera_adapters[i](scatter(pose_embeddings[i](enc))) -> (K,1,C)
    (omnivggt_aggregator.py:172-178,211 for i=0; :273-287 for i>=1).  Views without a GT
    camera receive the adapter bias (Linear of a zero row)."""
    P = "aggregator"

    def lin(name, v):
        return F.linear(v, sd["%s.%s.%d.weight" % (P, name, i)], sd["%s.%s.%d.bias" % (P, name, i)])

    if st["enc"] is not None:
        src = scatter_rows(lin("pose_embeddings", st["enc"]), st["B"], st["S"], st["camera_gt_index"], st["K"], st["C"])
    else:
        src = torch.zeros(st["K"], 1, st["C"])
    return lin("camera_adapters", src)


def aggregator_forward(sd, images, extrinsics, intrinsics, depth, mask, depth_gt_index, camera_gt_index,
                       depth_layers=24, dino_layers=24, capture=None):
    """models/omnivggt_aggregator.py:130-305 + models/aggregator.py:312-341.

    Returns (list of depth_layers tensors (B,S,P,2C), patch_start_idx).  `capture`, if a
    dict, receives intermediate tensors ("tokens0").
    """
    st = aggregator_prepare(sd, images, extrinsics, intrinsics, depth, mask, depth_gt_index, camera_gt_index, dino_layers)
    B, S, K, C = st["B"], st["S"], st["K"], st["C"]
    tokens, pos, rope = st["tokens"], st["pos"], st["rope"]
    if capture is not None:
        capture["tokens0"] = tokens
    T = tokens.shape[1]
    out = []
    for i in range(depth_layers):
        # frame attention + camera injection (omnivggt_aggregator.py:258-305)
        tokens = block(tokens.view(K, T, C), sd, "aggregator.frame_blocks.%d" % i, pos, rope, True)
        tokens = torch.cat([tokens[:, :1] + camera_injection(sd, i + 1, st), tokens[:, 1:]], dim=1)
        frame_out = tokens.view(B, S, T, C)
        # global attention (aggregator.py:312-341)
        tokens = block(tokens.view(B, S * T, C), sd, "aggregator.global_blocks.%d" % i, pos.reshape(B, S * T, 2), rope, True)
        out.append(torch.cat([frame_out, tokens.view(B, S, T, C)], dim=-1))
    return out, N_SPECIAL


# ----------------------------------------------------------------------------
# heads (consumers of the hot path; heads/camera_head.py, heads/dpt_head.py)
# ----------------------------------------------------------------------------
def camera_head_forward(sd, tokens_last, iterations=4, prefix="camera_head"):
    """heads/camera_head.py:83-154."""
    x = layer_norm(tokens_last[:, :, 0], sd, prefix + ".token_norm", 1e-5)
    B, S, C = x.shape
    pred, outs = None, []
    for _ in range(iterations):
        src = sd[prefix + ".empty_pose_tokens"].expand(B, S, -1) if pred is None else pred
        emb = F.linear(src, sd[prefix + ".embed_pose.weight"], sd[prefix + ".embed_pose.bias"])
        mod = F.linear(F.silu(emb), sd[prefix + ".poseLN_modulation.1.weight"], sd[prefix + ".poseLN_modulation.1.bias"])
        shift, scale, gate = mod.chunk(3, dim=-1)
        y = gate * (F.layer_norm(x, (C,), None, None, 1e-6) * (1 + scale) + shift) + x
        for j in range(4):
            y = block(y, sd, "%s.trunk.%d" % (prefix, j))
        y = layer_norm(y, sd, prefix + ".trunk_norm", 1e-5)
        h = F.gelu(F.linear(y, sd[prefix + ".pose_branch.fc1.weight"], sd[prefix + ".pose_branch.fc1.bias"]))
        delta = F.linear(h, sd[prefix + ".pose_branch.fc2.weight"], sd[prefix + ".pose_branch.fc2.bias"])
        pred = delta if pred is None else pred + delta
        outs.append(torch.cat([pred[..., :7], F.relu(pred[..., 7:])], dim=-1))   # heads/head_act.py:12-35
    return outs


def _uv_embed(x, W, H, ratio=0.1):
    """heads/dpt_head.py:262-272 + heads/utils.py:11-108."""
    pw, ph, Cc = x.shape[-1], x.shape[-2], x.shape[1]
    aspect = W / H
    diag = (aspect ** 2 + 1.0) ** 0.5
    sx, sy = aspect / diag, 1.0 / diag
    xs = torch.linspace(-sx * (pw - 1) / pw, sx * (pw - 1) / pw, steps=pw, dtype=x.dtype)
    ys = torch.linspace(-sy * (ph - 1) / ph, sy * (ph - 1) / ph, steps=ph, dtype=x.dtype)
    uu, vv = torch.meshgrid(xs, ys, indexing="xy")

    def sincos(p, dim):
        omega = torch.arange(dim // 2, dtype=torch.double) / (dim / 2.0)
        omega = 1.0 / 100 ** omega
        o = torch.einsum("m,d->md", p.reshape(-1), omega)
        return torch.cat([torch.sin(o), torch.cos(o)], dim=1).float()

    emb = torch.cat([sincos(uu, Cc // 2), sincos(vv, Cc // 2)], dim=-1).view(ph, pw, Cc)
    return x + (emb * ratio).permute(2, 0, 1)[None]


def _conv(x, sd, name, stride=1, padding=0):
    return F.conv2d(x, sd[name + ".weight"], sd.get(name + ".bias"), stride=stride, padding=padding)


def _rcu(x, sd, name):
    """heads/dpt_head.py:352-395 (inplace ReLU: the skip connection sees relu(x))."""
    x = F.relu(x)
    y = _conv(x, sd, name + ".conv1", padding=1)
    y = _conv(F.relu(y), sd, name + ".conv2", padding=1)
    return y + x


def _fusion(sd, name, x, skip=None, size=None):
    """heads/dpt_head.py:398-465."""
    if skip is not None:
        x = x + _rcu(skip, sd, name + ".resConfUnit1")
    x = _rcu(x, sd, name + ".resConfUnit2")
    if size is None:
        size = (x.shape[-2] * 2, x.shape[-1] * 2)
    x = F.interpolate(x, size=size, mode="bilinear", align_corners=True)
    return _conv(x, sd, name + ".out_conv")


def dpt_head_forward(sd, prefix, tokens_list, images, patch_start=N_SPECIAL, chunk=8, activation="exp",
                     layers=(4, 11, 17, 23)):
    """heads/dpt_head.py:128-260 + heads/head_act.py:61-125."""
    B, S, _, H, W = images.shape
    ph, pw = H // PATCH, W // PATCH
    preds, confs = [], []
    for s0 in range(0, S, chunk if chunk and chunk < S else S):
        s1 = min(s0 + (chunk if chunk and chunk < S else S), S)
        n = s1 - s0
        feats = []
        for li, layer in enumerate(layers):
            x = tokens_list[layer][:, s0:s1, patch_start:].reshape(B * n, ph * pw, -1)
            x = layer_norm(x, sd, prefix + ".norm", 1e-5)
            x = x.permute(0, 2, 1).reshape(B * n, -1, ph, pw)
            x = _uv_embed(_conv(x, sd, "%s.projects.%d" % (prefix, li)), W, H)
            rn = "%s.resize_layers.%d" % (prefix, li)
            if li == 0:
                x = F.conv_transpose2d(x, sd[rn + ".weight"], sd[rn + ".bias"], stride=4)
            elif li == 1:
                x = F.conv_transpose2d(x, sd[rn + ".weight"], sd[rn + ".bias"], stride=2)
            elif li == 3:
                x = _conv(x, sd, rn, stride=2, padding=1)
            feats.append(F.conv2d(x, sd["%s.scratch.layer%d_rn.weight" % (prefix, li + 1)], padding=1))
        sc = prefix + ".scratch"
        y = _fusion(sd, sc + ".refinenet4", feats[3], size=feats[2].shape[2:])
        y = _fusion(sd, sc + ".refinenet3", y, feats[2], size=feats[1].shape[2:])
        y = _fusion(sd, sc + ".refinenet2", y, feats[1], size=feats[0].shape[2:])
        y = _fusion(sd, sc + ".refinenet1", y, feats[0])
        y = _conv(y, sd, sc + ".output_conv1", padding=1)
        y = F.interpolate(y, size=(ph * PATCH, pw * PATCH), mode="bilinear", align_corners=True)
        y = _uv_embed(y, W, H)
        y = _conv(F.relu(_conv(y, sd, sc + ".output_conv2.0", padding=1)), sd, sc + ".output_conv2.2")
        y = y.permute(0, 2, 3, 1)
        xyz, conf = y[..., :-1], y[..., -1]
        if activation == "exp":
            xyz = torch.exp(xyz)
        elif activation == "inv_log":
            xyz = torch.sign(xyz) * torch.expm1(torch.abs(xyz))
        else:
            raise ValueError(activation)
        preds.append(xyz.reshape(B, n, *xyz.shape[1:]))
        confs.append((1 + conf.exp()).reshape(B, n, *conf.shape[1:]))
    return torch.cat(preds, dim=1), torch.cat(confs, dim=1)


def model_forward(sd, images, extrinsics, intrinsics, depth, mask, depth_gt_index, camera_gt_index,
                  depth_layers=24, dino_layers=24):
    """models/omnivggt.py:20-68."""
    if images.dim() == 4:
        images = images.unsqueeze(0)
    toks, start = aggregator_forward(sd, images, extrinsics, intrinsics, depth, mask, depth_gt_index,
                                     camera_gt_index, depth_layers, dino_layers)
    out = {}
    poses = camera_head_forward(sd, toks[-1])
    out["pose_enc"], out["pose_enc_list"] = poses[-1], poses
    layers = (4, 11, 17, 23) if depth_layers == 24 else tuple(min(l, depth_layers - 1) for l in (4, 11, 17, 23))
    out["depth"], out["depth_conf"] = dpt_head_forward(sd, "depth_head", toks, images, start, activation="exp", layers=layers)
    out["world_points"], out["world_points_conf"] = dpt_head_forward(sd, "point_head", toks, images, start,
                                                                     activation="inv_log", layers=layers)
    out["images"] = images
    out["_tokens"] = toks
    return out


# ----------------------------------------------------------------------------
# synthetic inputs: seeded, identical on every host
# ----------------------------------------------------------------------------
def synthetic_inputs(S, seed=1234, hw=518):
    """hw: int (square) or (H, W)."""
    g = torch.Generator().manual_seed(seed)
    Hh, Ww = (hw, hw) if isinstance(hw, int) else hw
    images = torch.rand(1, S, 3, Hh, Ww, generator=g)
    q = torch.randn(S, 4, generator=g)
    q = q / q.norm(dim=-1, keepdim=True)
    i, j, k, r = q.unbind(-1)
    two_s = 2.0 / (q * q).sum(-1)
    R = torch.stack([1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)], -1).reshape(S, 3, 3)
    t = torch.randn(S, 3, 1, generator=g)
    extrinsics = torch.cat([R, t], dim=-1).unsqueeze(0)
    f = 400 + 300 * torch.rand(S, generator=g)
    intrinsics = torch.zeros(1, S, 3, 3)
    intrinsics[0, :, 0, 0] = f
    intrinsics[0, :, 1, 1] = f
    intrinsics[0, :, 0, 2] = Ww / 2
    intrinsics[0, :, 1, 2] = Hh / 2
    intrinsics[0, :, 2, 2] = 1
    depth = 0.5 + 5 * torch.rand(1, S, Hh, Ww, 1, generator=g)
    mask = (torch.rand(1, S, Hh, Ww, generator=g) > 0.2).float()
    return dict(images=images, extrinsics=extrinsics, intrinsics=intrinsics, depth=depth, mask=mask)
