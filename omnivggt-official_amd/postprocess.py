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


def unproject_depth_map_to_point_map(depth_map, extrinsics_cam, intrinsics_cam, check_skew=None):
    """utils/geometry.py:151-266: depth (S,H,W,1) or (S,H,W), extrinsics (S,3,4) camera-from-world, intrinsics (S,3,3)
    -> world points (S,H,W,3) f32 on the device (the reference returns a numpy array after a host loop).

    Nothing here waits for the device: the camera-from-world inverse (closed_form_inverse_se3, geometry.py:269-318) and the packing of
    the 16 per-frame camera numbers are O(S) tensor operations on the depth map's device, queued in front of the one `ovg_unproject`
    launch. The reference's zero-skew assertion (geometry.py:251) is evaluated where it is free -- on intrinsics that arrive as HOST
    tensors -- and on device-resident intrinsics only when `check_skew=True` (one scalar device -> host read, i.e. a sync)."""
    if not depth_map.is_cuda:
        raise ops.L.OvgError("unproject_depth_map_to_point_map needs HIP device tensors: there is no CPU fallback")
    d = depth_map.squeeze(-1) if depth_map.dim() == 4 else depth_map
    d = d.float().contiguous()
    S = d.shape[0]
    if check_skew is None:
        check_skew = not intrinsics_cam.is_cuda
    if check_skew:
        k = intrinsics_cam.detach().reshape(S, 3, 3)
        if bool(((k[:, 0, 1] != 0) | (k[:, 1, 0] != 0)).any()):
            raise AssertionError("Intrinsic matrix must have zero skew")      # geometry.py:251
    ext = extrinsics_cam.detach().to(device=d.device, dtype=torch.float32, non_blocking=True).reshape(S, 3, 4)
    intr = intrinsics_cam.detach().to(device=d.device, dtype=torch.float32, non_blocking=True).reshape(S, 3, 3)
    Rt = ext[:, :, :3].transpose(1, 2)                                       # world-from-camera rotation
    c = -torch.bmm(Rt, ext[:, :, 3:])                                        # camera centre in the world
    cam = torch.cat([Rt.reshape(S, 9), c.reshape(S, 3), intr[:, 0, 0:1], intr[:, 1, 1:2], intr[:, 0, 2:3], intr[:, 1, 2:3]], dim=1)
    return ops.unproject(d, cam.contiguous())
