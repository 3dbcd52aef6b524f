"""Pin oracle/postprocess_oracle.py against the REAL reference (build container only) and write
tests/golden/unproject.npz (inputs + reference outputs, small) for the GPU box.

    python oracle/gen_golden_postprocess.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import aggregator_oracle as orc  # noqa: E402
import postprocess_oracle as ppo  # noqa: E402
import ref_shim  # noqa: E402


def main():
    ref_shim.install()
    from omnivggt.utils.geometry import unproject_depth_map_to_point_map as ref_unproject
    from omnivggt.utils.pose_enc import pose_encoding_to_extri_intri as ref_decode
    inp = orc.synthetic_inputs(3, hw=(70, 98))
    depth = inp["depth"][0].numpy()                       # (S,H,W,1)
    ext = inp["extrinsics"][0].numpy()
    intr = inp["intrinsics"][0].numpy()
    ref = ref_unproject(depth, ext, intr)
    mine = ppo.unproject_depth_map_to_point_map(depth, ext, intr)
    assert ref.dtype == mine.dtype and np.array_equal(ref, mine), float(np.abs(ref - mine).max())
    g = torch.Generator().manual_seed(3)
    enc = torch.randn(1, 3, 9, generator=g)
    enc[..., 7:] = 0.6 + 0.5 * torch.rand(1, 3, 2, generator=g)
    e_ref, k_ref = ref_decode(enc, (70, 98))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "unproject.npz"), depth=depth, extrinsics=ext, intrinsics=intr,
                        world=ref, pose_enc=enc.numpy(), dec_extrinsics=e_ref.numpy(), dec_intrinsics=k_ref.numpy())
    print("postprocess oracle == reference (bit-exact); golden written")


if __name__ == "__main__":
    main()
