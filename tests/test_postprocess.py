"""post-processing. CPU: the numpy oracle and the pose decoding against golden outputs of the
REAL reference functions (oracle/gen_golden_postprocess.py). GPU: `ovg_unproject` against the same golden."""
import os
import sys

import numpy as np
import pytest
import torch

import common
from omnivggt_official_amd import camera_math, postprocess
from omnivggt_official_amd import lib as L

sys.path.insert(0, os.path.join(common.ROOT, "oracle"))
import postprocess_oracle as ppo  # noqa: E402


def _gold():
    return dict(np.load(os.path.join(common.GOLD, "unproject.npz")))


def test_unproject_oracle_reproduces_reference_bit_exactly():
    g = _gold()
    mine = ppo.unproject_depth_map_to_point_map(g["depth"], g["extrinsics"], g["intrinsics"])
    assert mine.dtype == g["world"].dtype == np.float64 and np.array_equal(mine, g["world"])


def test_pose_decoding_matches_reference():
    g = _gold()
    ext, K = postprocess.pose_encoding_to_extri_intri(torch.from_numpy(g["pose_enc"]), (70, 98))
    assert common.max_rel(ext, g["dec_extrinsics"]) <= 1e-6 and common.max_rel(K, g["dec_intrinsics"]) <= 1e-6
    with pytest.raises(NotImplementedError):
        postprocess.pose_encoding_to_extri_intri(torch.from_numpy(g["pose_enc"]), (70, 98), pose_encoding_type="other")


def test_unproject_has_no_cpu_path():
    g = _gold()
    with pytest.raises(L.OvgError):
        postprocess.unproject_depth_map_to_point_map(torch.from_numpy(g["depth"]), torch.from_numpy(g["extrinsics"]), torch.from_numpy(g["intrinsics"]))


@pytest.mark.gpu
def test_unproject_kernel_vs_reference_golden():
    L.require_gpu()
    g = _gold()
    out = postprocess.unproject_depth_map_to_point_map(torch.from_numpy(g["depth"]).cuda(), torch.from_numpy(g["extrinsics"]),
                                                       torch.from_numpy(g["intrinsics"]))
    assert out.shape == g["world"].shape and out.dtype == torch.float32
    assert common.max_rel(out.cpu(), g["world"]) <= 1e-6       # f32 rounding of the reference's float64 points
    # size-independent property at full resolution: identity pose, unit focal length -> x = (u-cu) d, y = (v-cv) d, z = d
    S, H, W = 2, 518, 518
    d = torch.rand(S, H, W, device="cuda") + 0.5
    ext = torch.eye(4)[:3].repeat(S, 1, 1)
    K = torch.tensor([[1.0, 0, 259.0], [0, 1.0, 259.0], [0, 0, 1]]).repeat(S, 1, 1)
    pts = postprocess.unproject_depth_map_to_point_map(d, ext, K)
    u = torch.arange(W, device="cuda").view(1, 1, W).float()
    v = torch.arange(H, device="cuda").view(1, H, 1).float()
    assert torch.equal(pts[..., 2], d)
    assert torch.allclose(pts[..., 0], (u - 259.0) * d, rtol=1e-6, atol=0) and torch.allclose(pts[..., 1], (v - 259.0) * d, rtol=1e-6, atol=0)
