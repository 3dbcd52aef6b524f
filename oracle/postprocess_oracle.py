"""CPU restatement (numpy, test infrastructure only) of the reference's depth un-projection,
omnivggt/utils/geometry.py:151-266 + :269-318, used to check `ovg_unproject`
(omnivggt-official_amd/postprocess.py). Pinned: oracle/gen_golden_postprocess.py runs the REAL reference
functions in the build container on seeded inputs, asserts agreement with this file and stores
tests/golden/unproject.npz for the GPU box.
"""
import numpy as np


def closed_form_inverse_se3(se3):
    """geometry.py:269-318 (numpy branch): [R | t] -> [R^T | -R^T t], batched (N,4,4) or (N,3,4)."""
    R = se3[:, :3, :3]
    T = se3[:, :3, 3:]
    Rt = np.transpose(R, (0, 2, 1))
    top_right = -np.matmul(Rt, T)
    inv = np.tile(np.eye(4), (len(R), 1, 1))          # float64, as in the reference: the world points come out in float64
    inv[:, :3, :3] = Rt
    inv[:, :3, 3:] = top_right
    return inv


def depth_to_cam_coords_points(depth_map, intrinsic):
    """geometry.py:231-266."""
    H, W = depth_map.shape
    fu, fv = intrinsic[0, 0], intrinsic[1, 1]
    cu, cv = intrinsic[0, 2], intrinsic[1, 2]
    u, v = np.meshgrid(np.arange(W), np.arange(H))
    x_cam = (u - cu) * depth_map / fu
    y_cam = (v - cv) * depth_map / fv
    return np.stack((x_cam, y_cam, depth_map), axis=-1).astype(np.float32)


def unproject_depth_map_to_point_map(depth_map, extrinsics_cam, intrinsics_cam):
    """geometry.py:151-229: (S,H,W[,1]) depth, (S,3,4) camera-from-world, (S,3,3) -> (S,H,W,3) world points."""
    out = []
    for i in range(depth_map.shape[0]):
        d = depth_map[i].squeeze(-1) if depth_map[i].ndim == 3 else depth_map[i]
        cam = depth_to_cam_coords_points(d, intrinsics_cam[i])
        c2w = closed_form_inverse_se3(extrinsics_cam[i][None])[0]
        out.append(np.dot(cam, c2w[:3, :3].T) + c2w[:3, 3])
    return np.stack(out, axis=0)
