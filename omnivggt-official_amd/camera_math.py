"""Camera-modality input preparation (O(S) scalars, fp32, bit-faithful to the reference).

Mirrors omnivggt/models/omnivggt_aggregator.py:85-105 (normalize_extrinsics),
omnivggt/utils/geometry.py:269-318 (closed-form SE3 inverse), omnivggt/utils/pose_enc.py:11-62
and omnivggt/utils/rotation.py:47-138.  Runs wherever its inputs live; the aggregator calls
it on host copies (S x 12 floats) so no device kernels are spent on it.
"""
import torch


def se3_inverse(mat):
    """[N,4,4] rigid transforms -> inverses: [R^T | -R^T t]."""
    R = mat[:, :3, :3]
    t = mat[:, :3, 3:]
    inv = torch.eye(4, dtype=mat.dtype, device=mat.device).repeat(mat.shape[0], 1, 1)
    Rt = R.transpose(1, 2)
    inv[:, :3, :3] = Rt
    inv[:, :3, 3:] = -torch.bmm(Rt, t)
    return inv


def normalize_extrinsics(extrinsics):
    """[B,S,3,4] w2c -> first camera = identity, translations / mean distance to camera 0."""
    B, S = extrinsics.shape[:2]
    last_row = torch.zeros(B, S, 1, 4, dtype=extrinsics.dtype, device=extrinsics.device)
    last_row[..., 3] = 1.0
    full = torch.cat([extrinsics, last_row], dim=-2)
    rel = torch.matmul(full, se3_inverse(full[:, 0]).unsqueeze(1))
    if S > 1:
        c = rel[:, :, :3, 3]
        d = torch.norm(c - c[:, :1], dim=-1)[:, 1:]
        s = d.mean(dim=1, keepdim=True).clamp(min=1e-6)
        rel[:, :, :3, 3] = rel[:, :, :3, 3] / s.unsqueeze(-1)
    return rel[:, :, :3]


def rotation_to_quaternion(R):
    """[...,3,3] -> [...,4] xyzw with non-negative w (PyTorch3D-style best-conditioned branch)."""
    f = R.reshape(R.shape[:-2] + (9,))
    a, b, c, d, e, f_, g, h, i = f.unbind(-1)
    t = torch.stack([1.0 + a + e + i, 1.0 + a - e - i, 1.0 - a + e - i, 1.0 - a - e + i], dim=-1)
    mag = torch.where(t > 0, torch.sqrt(t.clamp(min=0)), torch.zeros_like(t))
    rows = [
        torch.stack([mag[..., 0] ** 2, h - f_, c - g, d - b], dim=-1),
        torch.stack([h - f_, mag[..., 1] ** 2, d + b, c + g], dim=-1),
        torch.stack([c - g, d + b, mag[..., 2] ** 2, f_ + h], dim=-1),
        torch.stack([d - b, g + c, h + f_, mag[..., 3] ** 2], dim=-1),
    ]
    cand = torch.stack(rows, dim=-2) / (2.0 * mag[..., None].clamp(min=0.1))
    pick = mag.argmax(dim=-1)
    q = torch.gather(cand, -2, pick[..., None, None].expand(pick.shape + (1, 4))).squeeze(-2)
    q = q[..., [1, 2, 3, 0]]
    return torch.where(q[..., 3:4] < 0, -q, q)


def quaternion_to_rotation(q):
    """[...,4] xyzw -> [...,3,3]."""
    x, y, z, w = q.unbind(-1)
    s = 2.0 / (q * q).sum(-1)
    m = torch.stack([1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w),
                     s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w),
                     s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y)], dim=-1)
    return m.reshape(q.shape[:-1] + (3, 3))


def pose_encoding(extrinsics, intrinsics, image_hw):
    """absT_quaR_FoV: [t(3), quat xyzw(4), fov_h, fov_w] (pose_enc.py:48-59)."""
    H, W = image_hw
    fov_h = 2 * torch.atan((H / 2) / intrinsics[..., 1, 1])
    fov_w = 2 * torch.atan((W / 2) / intrinsics[..., 0, 0])
    quat = rotation_to_quaternion(extrinsics[..., :3, :3])
    return torch.cat([extrinsics[..., :3, 3], quat, fov_h[..., None], fov_w[..., None]], dim=-1).float()


def pose_decoding(enc, image_hw=None, build_intrinsics=True):
    """Inverse of pose_encoding (pose_enc.py:65-130): -> ([B,S,3,4], [B,S,3,3] or None)."""
    R = quaternion_to_rotation(enc[..., 3:7])
    ext = torch.cat([R, enc[..., :3, None]], dim=-1)
    K = None
    if build_intrinsics:
        H, W = image_hw
        K = torch.zeros(enc.shape[:2] + (3, 3), device=enc.device)
        K[..., 0, 0] = (W / 2.0) / torch.tan(enc[..., 8] / 2.0)
        K[..., 1, 1] = (H / 2.0) / torch.tan(enc[..., 7] / 2.0)
        K[..., 0, 2] = W / 2
        K[..., 1, 2] = H / 2
        K[..., 2, 2] = 1.0
    return ext, K
