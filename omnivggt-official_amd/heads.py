"""Prediction heads that consume the aggregator output (stock PyTorch-ROCm ops).

Out of kernel scope this round: 0.3 % (camera) / 17 % (DPT) of the
reference's time, conv nets that run through MIOpen/rocBLAS on the GPU.  Written from the
behaviour of omnivggt/heads/camera_head.py:19-162, omnivggt/heads/dpt_head.py:21-497,
omnivggt/heads/head_act.py:12-125 and omnivggt/heads/utils.py:11-108 with identical
state-dict keys so `OmniVGGT.safetensors` loads with strict=True.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _Scale(nn.Module):
    def __init__(self, dim, init):
        super().__init__()
        self.gamma = nn.Parameter(init * torch.ones(dim))

    def forward(self, x):
        return x * self.gamma


class _SelfAttention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.heads = heads
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, Cc = x.shape
        q, k, v = self.qkv(x).view(B, N, 3, self.heads, Cc // self.heads).permute(2, 0, 3, 1, 4)
        y = F.scaled_dot_product_attention(q, k, v)
        return self.proj(y.transpose(1, 2).reshape(B, N, Cc))


class _FeedForward(nn.Module):
    def __init__(self, dim, hidden, out=None):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, out or dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class TrunkBlock(nn.Module):
    """Pre-LN block of the camera trunk (layers/block.py semantics, dim 2048, no RoPE/qk-norm)."""
    def __init__(self, dim, heads, init_values):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _SelfAttention(dim, heads)
        self.ls1 = _Scale(dim, init_values)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _FeedForward(dim, 4 * dim)
        self.ls2 = _Scale(dim, init_values)

    def forward(self, x):
        x = x + self.ls1(self.attn(self.norm1(x)))
        return x + self.ls2(self.mlp(self.norm2(x)))


class CameraHead(nn.Module):
    """Iterative pose regressor on the camera tokens of the last aggregator layer."""

    def __init__(self, dim_in=2048, trunk_depth=4, num_heads=16, init_values=0.01, target_dim=9):
        super().__init__()
        self.trunk = nn.Sequential(*[TrunkBlock(dim_in, num_heads, init_values) for _ in range(trunk_depth)])
        self.token_norm = nn.LayerNorm(dim_in)
        self.trunk_norm = nn.LayerNorm(dim_in)
        self.empty_pose_tokens = nn.Parameter(torch.zeros(1, 1, target_dim))
        self.embed_pose = nn.Linear(target_dim, dim_in)
        self.poseLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(dim_in, 3 * dim_in))
        self.adaln_norm = nn.LayerNorm(dim_in, elementwise_affine=False, eps=1e-6)
        self.pose_branch = _FeedForward(dim_in, dim_in // 2, target_dim)

    def forward(self, aggregated_tokens_list, num_iterations=4):
        tok = self.token_norm(aggregated_tokens_list[-1][:, :, 0])
        B, S, _ = tok.shape
        pose, history = None, []
        for _ in range(num_iterations):
            prev = self.empty_pose_tokens.expand(B, S, -1) if pose is None else pose.detach()
            shift, scale, gate = self.poseLN_modulation(self.embed_pose(prev)).chunk(3, dim=-1)
            h = gate * (self.adaln_norm(tok) * (1 + scale) + shift) + tok
            delta = self.pose_branch(self.trunk_norm(self.trunk(h)))
            pose = delta if pose is None else pose + delta
            # translation / quaternion stay linear, field of view is ReLU'd (head_act.py:12-35)
            history.append(torch.cat([pose[..., :7], F.relu(pose[..., 7:])], dim=-1))
        return history


# ----------------------------------------------------------------------------
def uv_position_embedding(x, W, H, ratio=0.1, omega_0=100):
    """Add the sinusoidal UV embedding used by the DPT head (dpt_head.py:262-272)."""
    ch, ph, pw = x.shape[1], x.shape[2], x.shape[3]
    aspect = W / H
    diag = (aspect * aspect + 1.0) ** 0.5
    span_x, span_y = aspect / diag, 1.0 / diag
    u = torch.linspace(-span_x * (pw - 1) / pw, span_x * (pw - 1) / pw, steps=pw, dtype=x.dtype, device=x.device)
    v = torch.linspace(-span_y * (ph - 1) / ph, span_y * (ph - 1) / ph, steps=ph, dtype=x.dtype, device=x.device)
    uu, vv = torch.meshgrid(u, v, indexing="xy")
    quarter = ch // 4
    omega = torch.arange(quarter, dtype=torch.double, device=x.device) / float(quarter)
    omega = 1.0 / omega_0 ** omega

    def enc(p):
        ang = torch.einsum("m,d->md", p.reshape(-1), omega)
        return torch.cat([torch.sin(ang), torch.cos(ang)], dim=1).float()

    emb = torch.cat([enc(uu), enc(vv)], dim=-1).view(ph, pw, ch)
    return x + (emb * ratio).permute(2, 0, 1).unsqueeze(0)


class _ResidualConv(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv1 = nn.Conv2d(ch, ch, 3, padding=1)
        self.conv2 = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        # the reference's ReLU is in-place, so its skip connection carries relu(x)
        x = F.relu(x)
        return self.conv2(F.relu(self.conv1(x))) + x


class _Fusion(nn.Module):
    def __init__(self, ch, with_skip=True):
        super().__init__()
        self.out_conv = nn.Conv2d(ch, ch, 1)
        if with_skip:
            self.resConfUnit1 = _ResidualConv(ch)
        self.resConfUnit2 = _ResidualConv(ch)
        self.with_skip = with_skip

    def forward(self, x, skip=None, size=None):
        if self.with_skip:
            x = x + self.resConfUnit1(skip)
        x = self.resConfUnit2(x)
        size = size if size is not None else (2 * x.shape[-2], 2 * x.shape[-1])
        return self.out_conv(F.interpolate(x, size=tuple(size), mode="bilinear", align_corners=True))


class DPTHead(nn.Module):
    """Dense prediction head (depth: output_dim=2/'exp'; points: output_dim=4/'inv_log')."""

    def __init__(self, dim_in, patch_size=14, output_dim=4, activation="inv_log", conf_activation="expp1", features=256,
                 out_channels=(256, 512, 1024, 1024), intermediate_layer_idx=(4, 11, 17, 23)):
        super().__init__()
        if conf_activation != "expp1" or activation not in ("exp", "inv_log"):
            raise ValueError("unsupported DPT activation")
        self.patch_size, self.activation = patch_size, activation
        self.intermediate_layer_idx = tuple(intermediate_layer_idx)
        oc = list(out_channels)
        self.norm = nn.LayerNorm(dim_in)
        self.projects = nn.ModuleList([nn.Conv2d(dim_in, c, 1) for c in oc])
        self.resize_layers = nn.ModuleList([
            nn.ConvTranspose2d(oc[0], oc[0], 4, stride=4), nn.ConvTranspose2d(oc[1], oc[1], 2, stride=2),
            nn.Identity(), nn.Conv2d(oc[3], oc[3], 3, stride=2, padding=1)])
        sc = nn.Module()
        for i, c in enumerate(oc):
            setattr(sc, "layer%d_rn" % (i + 1), nn.Conv2d(c, features, 3, padding=1, bias=False))
        sc.refinenet1, sc.refinenet2, sc.refinenet3 = _Fusion(features), _Fusion(features), _Fusion(features)
        sc.refinenet4 = _Fusion(features, with_skip=False)
        sc.output_conv1 = nn.Conv2d(features, features // 2, 3, padding=1)
        sc.output_conv2 = nn.Sequential(nn.Conv2d(features // 2, 32, 3, padding=1), nn.ReLU(), nn.Conv2d(32, output_dim, 1))
        self.scratch = sc

    def forward(self, aggregated_tokens_list, images, patch_start_idx, frames_chunk_size=8):
        S = images.shape[1]
        step = S if (not frames_chunk_size or frames_chunk_size >= S) else frames_chunk_size
        parts = [self._chunk(aggregated_tokens_list, images, patch_start_idx, s, min(s + step, S)) for s in range(0, S, step)]
        return torch.cat([p[0] for p in parts], dim=1), torch.cat([p[1] for p in parts], dim=1)

    def _chunk(self, toks, images, start, s0, s1):
        B, _, _, H, W = images.shape
        n, ph, pw = s1 - s0, H // self.patch_size, W // self.patch_size
        pyramid = []
        for i, layer in enumerate(self.intermediate_layer_idx):
            x = toks[layer][:, s0:s1, start:].reshape(B * n, ph * pw, -1)
            x = self.norm(x).transpose(1, 2).reshape(B * n, -1, ph, pw)
            x = self.resize_layers[i](uv_position_embedding(self.projects[i](x), W, H))
            pyramid.append(getattr(self.scratch, "layer%d_rn" % (i + 1))(x))
        sc = self.scratch
        y = sc.refinenet4(pyramid[3], size=pyramid[2].shape[2:])
        y = sc.refinenet3(y, pyramid[2], size=pyramid[1].shape[2:])
        y = sc.refinenet2(y, pyramid[1], size=pyramid[0].shape[2:])
        y = sc.output_conv1(sc.refinenet1(y, pyramid[0]))
        y = F.interpolate(y, size=(ph * self.patch_size, pw * self.patch_size), mode="bilinear", align_corners=True)
        y = sc.output_conv2(uv_position_embedding(y, W, H)).permute(0, 2, 3, 1)
        val, conf = y[..., :-1], y[..., -1]
        val = torch.exp(val) if self.activation == "exp" else torch.sign(val) * torch.expm1(val.abs())
        return val.reshape(B, n, *val.shape[1:]), (1 + conf.exp()).reshape(B, n, *conf.shape[1:])
