"""CPU test of the HIP DPT-head HOST logic (heads_hip.py) against the PyTorch DPTHead, with the kernel
entries replaced by their torch emulation (tests/head_ops_emul.py). No GPU, no HIP code runs here; the
kernels themselves are checked against the same emulation in tests/gpu_selftest.py (`--only heads`)."""
import importlib
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import omnivggt_official_amd  # noqa: F401,E402  (root shim: registers the package under an importable name)
import head_ops_emul as emul  # noqa: E402

heads = importlib.import_module("omnivggt_official_amd.heads")
heads_hip = importlib.import_module("omnivggt_official_amd.heads_hip")


def _rand_head(output_dim, activation, seed):
    torch.manual_seed(seed)
    h = heads.DPTHead(dim_in=2048, output_dim=output_dim, activation=activation, conf_activation="expp1",
                      intermediate_layer_idx=(0, 1, 2, 3)).eval()
    with torch.no_grad():
        for name, p in h.named_parameters():          # default init leaves the deep convs nearly silent
            if p.dim() > 1:
                p.mul_(1.6)
            elif "bias" in name:
                p.uniform_(-0.2, 0.2)
        last = h.scratch.output_conv2[2]                # keep exp()/expm1() of the output stage in range
        last.weight.mul_(0.2 / float(last.weight.abs().max()))
    return h


@pytest.mark.parametrize("output_dim,activation", [(2, "exp"), (4, "inv_log")])
def test_hip_head_host_logic_matches_pytorch_head(monkeypatch, output_dim, activation):
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    head = _rand_head(output_dim, activation, seed=11 + output_dim)
    g = torch.Generator().manual_seed(5)
    S, P, Hpx = 2, 1374, 518
    toks = [torch.randn(1, S, P, 2048, generator=g) * 0.7 for _ in range(4)]
    images = torch.zeros(1, S, 3, Hpx, Hpx)
    with torch.no_grad():
        ref_val, ref_conf = head(toks, images=images, patch_start_idx=5)
        for fn in ("head_layernorm", "conv", "upsample", "dpt_out"):
            monkeypatch.setattr(heads_hip.ops, fn, getattr(emul, fn))
        hip = heads_hip.HipDPTHead(head)
        val, conf = hip(toks, images, 5, dtype=torch.float32)
    assert val.shape == ref_val.shape and conf.shape == ref_conf.shape
    for a, b, name in ((val, ref_val, "val"), (conf, ref_conf, "conf")):
        err = float((a - b).abs().max() / b.abs().max())
        assert err < 2e-4, (name, err)       # f32 both sides; only the out_conv/upsample order differs


def test_uv_tables_are_the_separable_form_of_the_reference_embedding():
    for ch, ph, pw in ((256, 37, 37), (128, 518, 518)):
        x = torch.zeros(1, ch, ph, pw)
        emb = heads.uv_position_embedding(x, 518, 518)[0]            # [ch, ph, pw]
        px, py = heads_hip.uv_tables(ch, ph, pw, 518, 518, "cpu")
        half = ch // 2
        assert torch.equal(emb[:half], px.t()[:, None, :].expand(half, ph, pw))
        assert torch.equal(emb[half:], py.t()[:, :, None].expand(half, ph, pw))


def _rand_camera_head(seed):
    torch.manual_seed(seed)
    h = heads.CameraHead(dim_in=2048).eval()
    with torch.no_grad():
        for name, p in h.named_parameters():          # default init: LayerScale 0.01, empty pose 0 -> the trunk would be nearly silent
            if name.endswith("gamma"):
                p.fill_(0.7)
            elif name == "empty_pose_tokens":
                p.normal_(0, 0.5)
            elif p.dim() > 1:
                p.mul_(1.5)
            elif "bias" in name:
                p.uniform_(-0.1, 0.1)
            elif "norm" in name and name.endswith("weight"):
                p.uniform_(0.8, 1.2)
    return h


def test_hip_camera_head_host_logic_matches_pytorch_head(monkeypatch):
    """heads_hip.HipCameraHead (weight packing, token view, batch loop, output list) on the torch restatement of the
    ovg_camera_head entry == the PyTorch CameraHead (itself oracle-checked, test_cpu_contract) in f32."""
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    head = _rand_camera_head(3)
    g = torch.Generator().manual_seed(9)
    B, S, P = 2, 5, 7
    toks = [torch.randn(B, S, P, 2048, generator=g) * 1.3]
    with torch.no_grad():
        ref = head(toks)
        monkeypatch.setattr(heads_hip.ops, "camera_head", emul.camera_head)
        monkeypatch.setattr(heads_hip.ops, "camera_head_workspace_bytes", lambda S, dtype: 16)
        got = heads_hip.HipCameraHead(head)(toks, dtype=torch.float32)
    assert len(got) == len(ref) == 4
    for a, b in zip(got, ref):
        assert a.shape == b.shape == (B, S, 9)
        assert float((a - b).abs().max() / b.abs().max()) < 2e-5
    assert float(ref[-1].abs().max()) > 1e-2 and float((ref[-1] - ref[0]).abs().max()) > 1e-3      # the refinement rounds do something
