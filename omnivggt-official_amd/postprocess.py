"""Post-processing of the prediction dict on the device.

`pose_encoding_to_extri_intri` mirrors omnivggt/utils/pose_enc.py:65-130 (O(S) tensor math, any device);
`unproject_depth_map_to_point_map` mirrors omnivggt/utils/geometry.py:151-266 but keeps the maps on the GPU:
the per-frame numpy loop of the reference becomes one `ovg_unproject` launch (no device->host copy of the
(S, H, W) maps before export / visualisation).
"""
import torch

from . import camera_math, ops


def pose_encoding_to_extri_intri(pose_encoding, image_size_hw=None, pose_encoding_type="absT_quaR_FoV", build_intrinsics=True):
    """utils/pose_enc.py:65-130: (B,S,9) -> extrinsics (B,S,3,4) camera-from-world, intrinsics (B,S,3,3) in pixels."""
    if pose_encoding_type != "absT_quaR_FoV":
        raise NotImplementedError
    return camera_math.pose_decoding(pose_encoding, image_size_hw, build_intrinsics)


def unproject_depth_map_to_point_map(depth_map, extrinsics_cam, intrinsics_cam):
    """utils/geometry.py:151-266: depth (S,H,W,1) or (S,H,W), extrinsics (S,3,4) camera-from-world, intrinsics (S,3,3)
    -> world points (S,H,W,3) f32 on the device (the reference returns a numpy array after a host loop)."""
    if not depth_map.is_cuda:
        raise ops.L.OvgError("unproject_depth_map_to_point_map needs HIP device tensors: there is no CPU fallback")
    d = depth_map.squeeze(-1) if depth_map.dim() == 4 else depth_map
    d = d.float().contiguous()
    S = d.shape[0]
    ext = extrinsics_cam.detach().float().cpu().reshape(S, 3, 4)
    intr = intrinsics_cam.detach().float().cpu().reshape(S, 3, 3)
    if bool((intr[:, 0, 1] != 0).any()) or bool((intr[:, 1, 0] != 0).any()):
        raise AssertionError("Intrinsic matrix must have zero skew")          # geometry.py:251
    full = torch.eye(4).repeat(S, 1, 1)
    full[:, :3] = ext
    c2w = camera_math.se3_inverse(full)                                     # closed_form_inverse_se3, geometry.py:269-318
    cam = torch.cat([c2w[:, :3, :3].reshape(S, 9), c2w[:, :3, 3], intr[:, 0, 0:1], intr[:, 1, 1:2], intr[:, 0, 2:3], intr[:, 1, 2:3]], dim=1)
    return ops.unproject(d, cam.contiguous().to(d.device))
