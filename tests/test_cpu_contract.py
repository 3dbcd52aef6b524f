"""CPU-side checks (no GPU): state-dict contract, C ABI surface, struct layout, loud failure
without a device, host-side camera math and heads vs the oracle."""
import ctypes
import json
import os
import sys
import re
import subprocess
import tempfile

import pytest
import torch

import aggregator_oracle as orc
from omnivggt_official_amd import camera_math, heads, lib as L, weights
from omnivggt_official_amd.aggregator import ZeroAggregator
from omnivggt_official_amd.model import OmniVGGT

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "omnivggt_hip.h")
MANIFEST = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_manifest.json")))


def test_state_dict_keys_match_reference_manifest():
    """1505 keys / shapes of the reference checkpoint contract (inference.py:323-324)."""
    with torch.device("meta"):
        m = OmniVGGT()
    mine = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert len(MANIFEST) == 1505
    assert mine == MANIFEST


def test_depth_reduced_manifest_is_a_subset():
    with torch.device("meta"):
        m = OmniVGGT(depth=2, dino_depth=3)
    mine = {k: list(v.shape) for k, v in m.state_dict().items()}
    red = weights.reduce_manifest(MANIFEST, depth=2, dino_depth=3)
    assert mine == {k: list(v) for k, v in red.items()}


def test_library_exports_every_declared_symbol():
    text = open(HEADER).read()
    declared = set(re.findall(r"\b(ovg_[a-z0-9_]+)\s*\(", text))
    assert declared, "no prototypes parsed"
    lib = L.load()
    for name in declared:
        assert hasattr(lib, name), "missing export " + name
    assert declared == set(L.SYMBOLS), (declared ^ set(L.SYMBOLS))
    assert lib.ovg_abi_version() == L.ABI_VERSION
    assert b"gfx950" in lib.ovg_build_info()


def test_ctypes_struct_layout_matches_c():
    """sizeof() of every params struct as gcc sees the header == the ctypes mirror."""
    pairs = {"ovg_layernorm_params": L.LayerNormParams, "ovg_linear_params": L.LinearParams, "ovg_qkv_params": L.QkvParams,
             "ovg_kv_segment": L.KvSegment, "ovg_attn_params": L.AttnParams, "ovg_block_weights": L.BlockWeights,
             "ovg_block_params": L.BlockParams, "ovg_im2col_params": L.Im2colParams, "ovg_depth_stats_params": L.DepthStatsParams,
             "ovg_dino_specials_params": L.DinoSpecialsParams, "ovg_assemble_params": L.AssembleParams,
             "ovg_copy_rows_params": L.CopyRowsParams, "ovg_head_layernorm_params": L.HeadLayerNormParams,
             "ovg_conv_params": L.ConvParams, "ovg_upsample_params": L.UpsampleParams, "ovg_dpt_out_params": L.DptOutParams, "ovg_dpt_tail_params": L.DptTailParams,
             "ovg_unproject_params": L.UnprojectParams, "ovg_heads_to_tokens_params": L.HeadsToTokensParams,
             "ovg_attn_merge_params": L.AttnMergeParams, "ovg_block_workspace": L.BlockWorkspace,
             "ovg_pack_weights_params": L.PackWeightsParams, "ovg_camera_block_weights": L.CameraBlockWeights,
             "ovg_camera_head_params": L.CameraHeadParams, "ovg_camera_tables_params": L.CameraTablesParams}
    src = '#include <stdio.h>\n#include "%s"\nint main(){\n' % HEADER
    for name in pairs:
        src += 'printf("%s %%zu\\n", sizeof(%s));\n' % (name, name)
    src += "return 0;}\n"
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-std=c99", c, "-o", exe])
        out = subprocess.check_output([exe]).decode()
    sizes = dict(line.split() for line in out.strip().splitlines())
    for name, cls in pairs.items():
        assert int(sizes[name]) == ctypes.sizeof(cls), name


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device."""
    lib = L.load()
    p = L.LinearParams()
    assert lib.ovg_linear(ctypes.byref(p), None) == -1
    a = L.AttnParams()
    assert lib.ovg_flash_attn(ctypes.byref(a), None) == -1
    assert lib.ovg_qkv(None, None) == -1
    assert lib.ovg_attn_merge(ctypes.byref(L.AttnMergeParams()), None) == -1
    assert lib.ovg_pack_weights(ctypes.byref(L.PackWeightsParams()), None) == -1
    assert not hasattr(lib, "ovg_debug_set")          # ABI 4: no process-global knobs left in the library
    assert lib.ovg_camera_head(ctypes.byref(L.CameraHeadParams()), None) == -1
    assert lib.ovg_camera_head(None, None) == -1
    assert lib.ovg_camera_tables(ctypes.byref(L.CameraTablesParams()), None) == -1 and lib.ovg_camera_tables(None, None) == -1
    # the camera head's workspace query is a pure host function: S tokens x (f32 token / residual / pose / 32768 partial
    # columns + 16-bit LN, embed, qkv, attention and hidden buffers), every piece rounded up to 256 bytes
    r = lambda n: (n + 255) // 256 * 256
    for S in (1, 8, 128):
        want = 2 * r(S * 2048 * 4) + r(S * 9 * 4) + r(S * 32768 * 4) + 3 * r(S * 2048 * 2) + r(S * 6144 * 2) + r(S * 8192 * 2)
        assert lib.ovg_camera_head_workspace_bytes(S, L.OVG_BF16) == want
        want32 = 2 * r(S * 2048 * 4) + r(S * 9 * 4) + r(S * 32768 * 4) + 3 * r(S * 2048 * 4) + r(S * 6144 * 4) + r(S * 8192 * 4)
        assert lib.ovg_camera_head_workspace_bytes(S, L.OVG_F32) == want32       # f32 parity mode: f32 activation buffers
    assert lib.ovg_camera_head_workspace_bytes(8, 99) == -1 and lib.ovg_camera_head_workspace_bytes(0, L.OVG_BF16) == -1
    # ovg_dpt_tail (ABI 10): NULL / bad shapes are OVG_E_ARG, a dtype or channel count it has no kernel for is OVG_E_UNSUPPORTED (the host
    # then runs the three-launch form) -- all decided before the device is touched
    assert lib.ovg_dpt_tail(None, None) == -1 and lib.ovg_dpt_tail(ctypes.byref(L.DptTailParams()), None) == -1
    fake = 0x10000
    t = L.DptTailParams()
    t.x, t.w1, t.w2, t.b2, t.val, t.conf = fake, fake, fake, fake, fake, fake
    t.ldx, t.ldw1, t.n_img, t.H, t.W, t.OH, t.OW, t.C, t.out_dim, t.activation, t.dtype = 128, 1152, 1, 296, 296, 518, 518, 128, 4, 1, L.OVG_F32
    assert lib.ovg_dpt_tail(ctypes.byref(t), None) == -4                        # f32: three launches
    t.dtype = L.OVG_F16X2
    assert lib.ovg_dpt_tail(ctypes.byref(t), None) == -4
    t.dtype, t.C, t.ldx = L.OVG_BF16, 256, 256
    assert lib.ovg_dpt_tail(ctypes.byref(t), None) == -4                        # not the model's 128-channel map
    t.C, t.ldx, t.out_dim = 128, 128, 5
    assert lib.ovg_dpt_tail(ctypes.byref(t), None) == -1
    t.out_dim, t.pos_x = 4, fake                                                # pos_x without pos_y
    assert lib.ovg_dpt_tail(ctypes.byref(t), None) == -1
    t.pos_x, t.ldw1 = None, 1000                                                # weight rows shorter than 9 * C
    assert lib.ovg_dpt_tail(ctypes.byref(t), None) == -1
    t.ldw1, t.dtype = 1152, 7
    assert lib.ovg_dpt_tail(ctypes.byref(t), None) == -2
    from omnivggt_official_amd import ops
    assert ops.dpt_tail_supported(torch.empty(1, 4, 4, 128), torch.bfloat16) and not ops.dpt_tail_supported(torch.empty(1, 4, 4, 128), torch.float32)
    assert not ops.dpt_tail_supported(torch.empty(1, 4, 4, 256), torch.float16) and not ops.dpt_tail_supported(torch.empty(1, 4, 4, 128), L.F32X)


def test_split_f16_mode_contract_on_cpu():
    """OVG_F16X2 / lib.F32X (ABI 8): the split rule, the dtype plumbing and the argument validation of the entries that take (hi, lo)
    plane pairs -- everything that can be checked without a device."""
    from omnivggt_official_amd import ops
    assert L.dtype_code(L.F32X) == L.OVG_F16X2 == 3 and L.storage_dtype(L.F32X) is torch.float16 and L.head_dtype(L.F32X) is torch.float32
    assert L.storage_dtype(torch.bfloat16) is torch.bfloat16 and L.head_dtype(torch.float16) is torch.float16 and repr(L.F32X) == "f32x"
    # split rule: hi = f16(sat(x)), lo = f16(x - hi): 2^-22 relative while lo is a normal f16, 2^-25 absolute below, saturating above
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.randn(4096, generator=g) * s for s in (1e-3, 0.02, 1.0, 50.0, 3e3)] + [torch.tensor([0.0, 65504.0, 70000.0, -1e5, 6e-8, 1e-9])])
    h = ops.to_hilo(x)
    assert h.hi.dtype == h.lo.dtype == torch.float16 and h.planes.shape == (2,) + tuple(x.shape)
    err = (h.float().double() - x.double()).abs()
    bound = torch.maximum(x.double().abs() * 2.0 ** -21, torch.full_like(err, 2.0 ** -24))
    ok = x.abs() <= 65504.0 * (1 + 2.0 ** -11)
    assert bool((err[ok] <= bound[ok]).all()), float((err[ok] / bound[ok]).max())
    assert torch.isfinite(h.hi).all() and torch.isfinite(h.lo).all()                 # saturating, never inf
    assert abs(float(h.float()[-4]) - 70000.0) <= 32.0                                # hi = 65504, lo carries the rest
    # the error model of a split contraction, emulated in torch: three products of f16 planes (exact in f32) accumulated in f32 against a
    # float64 evaluation of the same f32 operands -- f32-roundoff class, three orders below a plain f16 GEMM
    a32, b32 = torch.randn(256, 1024, generator=g), torch.randn(192, 1024, generator=g) * 0.03
    ah, bh = ops.to_hilo(a32), ops.to_hilo(b32)
    mm = lambda u, v: u.float() @ v.float().t()
    got = mm(ah.lo, bh.hi) + mm(ah.hi, bh.lo) + mm(ah.hi, bh.hi)
    ref = a32.double() @ b32.double().t()
    rel = lambda t: float((t.double() - ref).abs().max() / ref.abs().max())
    assert rel(got) <= 2e-6 and rel(mm(a32.half(), b32.half())) >= 1e-4, (rel(got), rel(mm(a32.half(), b32.half())))
    # argument validation (host side, before any launch)
    lib = L.load()
    fake = 0x10000
    p = L.LinearParams()
    p.x, p.w, p.y, p.M, p.N, p.K, p.ldx, p.ldw, p.ldy, p.dtype = fake, fake, fake, 128, 128, 128, 128, 128, 128, L.OVG_F16X2
    assert lib.ovg_linear(ctypes.byref(p), None) == -1                               # lo planes of x / w missing
    p.x_lo, p.w_lo = fake, fake
    assert lib.ovg_linear(ctypes.byref(p), None) == -1                               # 16-bit output needs y_lo
    p.y_lo = fake + 8
    assert lib.ovg_linear(ctypes.byref(p), None) == -1                               # misaligned
    q = L.QkvParams()
    q.x, q.w, q.bias, q.q, q.k, q.vt = (fake,) * 6
    q.M, q.seq, q.nq_pad, q.nk_pad, q.ldx, q.dtype = 128, 128, 128, 128, 1024, L.OVG_F16X2
    assert lib.ovg_qkv(ctypes.byref(q), None) == -1
    a = L.AttnParams()
    a.q, a.out, a.nq, a.nq_pad, a.BH, a.nseg, a.ldo, a.dtype = fake, fake, 64, 64, 16, 1, 1024, L.OVG_F16X2
    a.seg[0].k, a.seg[0].vt, a.seg[0].nk, a.seg[0].nk_pad = fake, fake, 64, 64
    assert lib.ovg_flash_attn(ctypes.byref(a), None) == -1                           # q_lo / out_lo / segment lo planes missing
    a.q_lo, a.out_lo, a.seg[0].k_lo, a.seg[0].vt_lo, a.kv_splits = fake, fake, fake, fake, 2
    assert lib.ovg_flash_attn(ctypes.byref(a), None) == -4                           # no split-KV in the split-f16 mode
    a.kv_splits, a.kv_heads = 0, 2
    assert lib.ovg_flash_attn(ctypes.byref(a), None) == -4                           # no head-parallel form either
    plan = ops.attn_plan(16, 87936, [87936], L.F32X)
    assert plan["splits"] == 1 and plan["q_tile"] == 256 and plan["tail_q_tile"] == 0 and plan["part_bytes"] == 0
    ws = ops.block_workspace_bytes(8 * 1374, 1374, L.F32X)                           # bytes of ONE plane of each scratch tensor
    assert ws["xn"] == 8 * 1374 * 1024 * 2 and ws["hid"] == 8 * 1374 * 4096 * 2
    for cls, fields in ((L.LayerNormParams, ("y_lo",)), (L.PackWeightsParams, ("dst_lo",)), (L.Im2colParams, ("out_lo",)),
                        (L.BlockWeights, ("qkv_w_lo", "proj_w_lo", "fc1_w_lo", "fc2_w_lo")), (L.BlockParams, ("ws_xn_lo", "ws_hid_lo", "attn_fallback_count")),
                        (L.AttnParams, ("q_lo", "out_lo", "fallback_count")), (L.KvSegment, ("k_lo", "vt_lo"))):
        names = [f for f, _ in cls._fields_]
        assert all(f in names for f in fields), cls
    # the aggregator takes the mode by sentinel or by name and refuses what the mode does not support
    with torch.device("meta"):
        agg = ZeroAggregator(depth=1, dino_depth=1, compute_dtype="f32x")
    assert agg.compute_dtype is L.F32X
    agg.set_compute_dtype(torch.float32)
    agg.set_compute_dtype("f32x")
    assert agg.compute_dtype is L.F32X
    with pytest.raises(ValueError):
        agg.set_compute_dtype(torch.float64)


def test_block_workspace_query_matches_the_python_allocation():
    """ovg_block_workspace_bytes (caller-provided workspace with a size query) vs what Workspace allocates."""
    from omnivggt_official_amd import ops
    for M, seq, dt, e in ((8 * 1374, 1374, torch.bfloat16, 2), (8 * 1374, 8 * 1374, torch.float16, 2), (2 * 1374, 1374, torch.float32, 4)):
        ws = ops.block_workspace_bytes(M, seq, dt)
        pad = (seq + 63) // 64 * 64
        BH = (M // seq) * 16
        assert ws["xn"] == ws["attn"] == M * 1024 * e and ws["hid"] == M * 4096 * e
        assert ws["q"] == ws["k"] == ws["vt"] == BH * pad * 64 * e
        assert ws["total"] == sum(ws[k] for k in ("xn", "attn", "hid", "q", "k", "vt"))
    p = L.BlockParams()
    assert L.load().ovg_block_workspace_bytes(ctypes.byref(p), ctypes.byref(L.BlockWorkspace())) == -1


def test_stale_library_is_detected(tmp_path, monkeypatch):
    """lib.load() compares the build stamp with the digest of csrc/ + header + flags: a library built from
    other sources is rebuilt when hipcc is there and refused otherwise -- never loaded silently."""
    from omnivggt_official_amd import build as B
    assert B.is_current()
    monkeypatch.setattr(B, "_digest", lambda: "0" * 64)
    assert not B.is_current()
    monkeypatch.setattr(B, "have_hipcc", lambda: False)
    monkeypatch.setattr(L, "_lib", None)
    with pytest.raises(L.OvgError, match="stale"):
        L.load()


def test_shard_mode_is_decided_without_communication():
    from omnivggt_official_amd.sharding import head_groups, resolve_mode
    assert resolve_mode("auto", 64, 8, False) == "heads" and resolve_mode("auto", 64, 1, False) == "allgather"
    assert resolve_mode("auto", 63, 8, False) == "allgather" and resolve_mode("auto", 64, 8, True) == "allgather"
    assert resolve_mode("allgather", 64, 8, False) == "allgather" and resolve_mode("heads", 128, 4, False) == "heads"
    for bad in ((63, 8, False), (64, 8, True), (6, 3, False)):
        with pytest.raises(ValueError):
            resolve_mode("heads", *bad)
    assert head_groups(2) == [(0, 1), (1, 1)] and head_groups(1) == [(0, 1)] and head_groups(8) == [(0, 4), (4, 4)]
    assert head_groups(2, 8, 8 * 1374) == [(0, 2)]                    # 8 ranks x 8 views: 344 workgroups per head -> one group
    assert head_groups(8, 2, 32 * 1374) == [(0, 4), (4, 4)]           # 2 ranks x 32 views: 1376 workgroups per group -> pipelined


def test_hot_path_fails_loudly_on_cpu():
    agg = ZeroAggregator(pose_hidden_dim=9, depth=1, dino_depth=1)
    x = torch.zeros(1, 2, 3, 518, 518)
    with pytest.raises(L.OvgError):
        agg(x, None, None, None, None, [], [])
    with pytest.raises(ValueError):
        agg(torch.zeros(1, 2, 4, 518, 518), None, None, None, None, [], [])


def test_camera_math_matches_oracle():
    inp = orc.synthetic_inputs(5)
    a = camera_math.normalize_extrinsics(inp["extrinsics"])
    b = orc.normalize_extrinsics(inp["extrinsics"])
    assert torch.equal(a, b)
    assert torch.equal(camera_math.pose_encoding(a, inp["intrinsics"], (518, 518)), orc.pose_encoding(b, inp["intrinsics"], (518, 518)))
    enc = camera_math.pose_encoding(a, inp["intrinsics"], (518, 518))
    ext, K = camera_math.pose_decoding(enc, (518, 518))
    assert torch.allclose(ext, a, atol=1e-5)
    assert torch.allclose(K[..., 0, 0], inp["intrinsics"][..., 0, 0], rtol=1e-5)


def test_partition_is_contiguous_and_complete():
    from omnivggt_official_amd.sharding import partition
    for n, w in ((8, 8), (64, 8), (10, 4), (3, 2), (128, 8), (9, 8)):
        parts = partition(n, w)
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
        sizes = [h - l for l, h in parts]
        assert max(sizes) - min(sizes) <= 1 and min(sizes) >= 1


def test_heads_match_oracle_on_cpu():
    """Product heads (nn.Module) vs the oracle's functional restatement, same synthetic weights."""
    man = {k: v for k, v in MANIFEST.items() if not k.startswith("aggregator.")}
    sd = weights.synthetic_state_dict(man, seed=3)
    cam = heads.CameraHead(dim_in=2048)
    cam.load_state_dict({k[len("camera_head."):]: v for k, v in sd.items() if k.startswith("camera_head.")}, strict=True)
    dh = heads.DPTHead(dim_in=2048, output_dim=2, activation="exp")
    dh.load_state_dict({k[len("depth_head."):]: v for k, v in sd.items() if k.startswith("depth_head.")}, strict=True)
    ph = heads.DPTHead(dim_in=2048, output_dim=4, activation="inv_log")
    ph.load_state_dict({k[len("point_head."):]: v for k, v in sd.items() if k.startswith("point_head.")}, strict=True)
    g = torch.Generator().manual_seed(0)
    S = 2
    toks = [torch.randn(1, S, 1374, 2048, generator=g) for _ in range(24)]
    images = torch.rand(1, S, 3, 518, 518, generator=g)
    with torch.no_grad():
        mine = cam(toks)
        ref = orc.camera_head_forward(sd, toks[-1])
        for a, b in zip(mine, ref):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
        d, dc = dh(toks, images, 5)
        rd, rdc = orc.dpt_head_forward(sd, "depth_head", toks, images, 5, activation="exp")
        assert torch.allclose(d, rd, rtol=1e-4, atol=1e-6) and torch.allclose(dc, rdc, rtol=1e-4, atol=1e-6)
        p, pc = ph(toks, images, 5)
        rp, rpc = orc.dpt_head_forward(sd, "point_head", toks, images, 5, activation="inv_log")
        assert torch.allclose(p, rp, rtol=1e-4, atol=1e-6) and torch.allclose(pc, rpc, rtol=1e-4, atol=1e-6)
        assert d.shape == (1, S, 518, 518, 1) and pc.shape == (1, S, 518, 518)


def test_from_safetensors_roundtrip(tmp_path):
    """checkpoint path -- a safetensors file with the reference's key set loads strictly into a
    model that was never initialised (meta -> to_empty), and every tensor comes back bit-identical."""
    import json
    import os
    import torch
    from safetensors.torch import save_file
    from omnivggt_official_amd import weights
    from omnivggt_official_amd.model import OmniVGGT
    here = os.path.dirname(os.path.abspath(__file__))
    manifest = json.load(open(os.path.join(here, "golden", "state_dict_manifest.json")))
    sd = weights.synthetic_state_dict(weights.reduce_manifest(manifest, 1, 1), seed=7)
    path = str(tmp_path / "tiny.safetensors")
    save_file({k: v.contiguous() for k, v in sd.items()}, path)
    m = OmniVGGT.from_safetensors(path, device="cpu", depth=1, dino_depth=1)
    got = m.state_dict()
    assert set(got) == set(sd)
    assert all(torch.equal(got[k], sd[k]) for k in sd)
    assert not m.training


@pytest.mark.skipif(not os.path.isdir("/root/reference/omnivggt"), reason="needs the reference tree (build container only)")
def test_integration_path_a_state_dict_swap_into_the_reference_model():
    """INTEGRATION.md path A, executed: the REFERENCE OmniVGGT (reference heads, reference key names) gets this repo's
    aggregator by load_state_dict(strict=True) + attribute swap; the swapped model keeps the reference's full key set and
    the aggregator keeps the reference's forward signature."""
    import inspect
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_shim
    ref = ref_shim.build_reference_model()
    ref_keys = set(ref.state_dict())
    assert len(ref_keys) == 1505
    hip = ZeroAggregator(pose_hidden_dim=9, compute_dtype=torch.bfloat16)
    missing = hip.load_state_dict(ref.aggregator.state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    assert len(hip.state_dict()) == len(ref.aggregator.state_dict()) == 1312
    for k, v in ref.aggregator.state_dict().items():
        assert torch.equal(hip.state_dict()[k], v)
    ref_sig = list(inspect.signature(ref.aggregator.forward).parameters)
    assert list(inspect.signature(hip.forward).parameters) == ref_sig
    ref.aggregator = hip
    assert set(ref.state_dict()) == ref_keys                     # checkpoint contract of the swapped model unchanged
    with pytest.raises(L.OvgError):                              # and it fails loudly without a HIP device (no silent CPU path)
        ref(torch.zeros(1, 2, 3, 518, 518), torch.zeros(1, 2, 3, 4), torch.zeros(1, 2, 3, 3), torch.zeros(1, 2, 518, 518, 1),
            torch.zeros(1, 2, 518, 518), [], [])


def test_vt_column_order_helpers():
    """16-bit V^T rows hold their keys in the PV fragment order (include/omnivggt_hip.h, ovg_qkv): inside every 32-key block column
    8 g + 4 h + i holds key 16 h + 4 g + i; f32 rows are natural. ops.set_vt / get_vt are the converters tests and external callers use."""
    from omnivggt_official_amd import ops
    idx = ops.vt_index(64, torch.bfloat16)
    assert idx[:32].tolist() == [0, 1, 2, 3, 16, 17, 18, 19, 4, 5, 6, 7, 20, 21, 22, 23, 8, 9, 10, 11, 24, 25, 26, 27, 12, 13, 14, 15, 28, 29, 30, 31]
    assert idx[32:].tolist() == [32 + k for k in idx[:32].tolist()]
    for g in range(4):                                   # every 16-byte chunk = keys {4g..4g+3, 16+4g..16+4g+3}: one lane group's fragment
        assert idx[8 * g: 8 * g + 8].tolist() == [4 * g + i for i in range(4)] + [16 + 4 * g + i for i in range(4)]
    assert torch.equal(ops.vt_index(64, torch.float32), torch.arange(64))
    v = torch.arange(2 * 64 * 70, dtype=torch.float32).reshape(2, 64, 70)
    for dt in (torch.bfloat16, torch.float16, torch.float32):
        vt = torch.zeros(2, 64, 128, dtype=dt)
        ops.set_vt(vt, v.to(dt))
        nat = ops.get_vt(vt)
        assert torch.equal(nat[:, :, :70], v.to(dt)) and float(nat[:, :, 70:].abs().max()) == 0.0


def test_bench_watchdog_dumps_stacks_prints_the_partial_line_and_exits(tmp_path):
    """bench.py's per-rank watchdog (round-2 review: a hung collective must leave a diagnosable record): a rank that makes no
    progress for `timeout` seconds writes a rank- and stage-tagged message plus every thread's stack to stderr, rank 0 prints the
    JSON line with what was measured so far and a `watchdog` record, and the process exits with code 3."""
    import json
    import sys
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "partial = {'metric': 'm', 'value': 12.5, 'n_gpus': 2}\n"
            "wd = bench.Watchdog(0, 2, 1.5, partial)\n"
            "wd.stage('second form: allgather')\n"
            "time.sleep(30)\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] == 12.5 and line["watchdog"]["stage"] == "second form: allgather" and line["watchdog"]["rank"] == 0
    assert "[rank 0/2" in r.stderr and "WATCHDOG: no progress" in r.stderr and "Thread" in r.stderr
    # a non-zero rank dumps its stacks but prints no JSON line (rank 0 owns stdout)
    r1 = subprocess.run([sys.executable, "-c", code.replace("Watchdog(0, 2", "Watchdog(1, 2")], capture_output=True, text=True, timeout=120)
    assert r1.returncode == 3 and r1.stdout.strip() == "" and "[rank 1/2]" in r1.stderr


def test_bench_self_launch_starts_n_ranks_and_propagates_the_exit_code(tmp_path):
    """`python bench.py --gpus N` (N > 1) with no launcher around it must start its own N ranks (round-3 review: the scaling leg
    could not start). bench.self_launch re-executes the script under torch.distributed.run on 127.0.0.1 with a free port, argv
    forwarded verbatim; rank 0's line is the only stdout; the job's exit code comes back (0 / the failing rank's)."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "2"], 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5] == os.path.join(ROOT, "bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "2"]
    script = tmp_path / "fake_bench.py"
    script.write_text("import os, sys, json\n"
                      "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
                      "assert os.environ['MASTER_ADDR'] == '127.0.0.1' and os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'\n"
                      "if r == 0: print(json.dumps({'n_gpus': w, 'argv': sys.argv[1:]}), flush=True)\n"
                      "sys.exit(7 if ('--fail' in sys.argv and r == 1) else 0)\n")
    code = ("import sys; sys.path.insert(0, %r); import bench\n"
            "sys.exit(bench.self_launch(2, sys.argv[1:], script=%r, timeout=240))\n") % (ROOT, str(script))
    ok = subprocess.run([sys.executable, "-c", code, "--gpus", "2", "--views", "8"], capture_output=True, text=True, timeout=300)
    assert ok.returncode == 0, ok.stderr[-2000:]
    line = json.loads(ok.stdout.strip().splitlines()[-1])
    assert line == {"n_gpus": 2, "argv": ["--gpus", "2", "--views", "8"]} and len(ok.stdout.strip().splitlines()) == 1
    bad = subprocess.run([sys.executable, "-c", code, "--fail"], capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0
    # the real script takes that road when WORLD_SIZE is unset: on this CPU-only host the ranks fail loudly (no HIP device)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--watchdog-s", "60"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert "[bench launcher]" in r.stderr and "torch.distributed.run" in r.stderr
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "no HIP device visible" in r.stderr


def test_committed_traffic_record_matches_the_attention_sources():
    """profiles/traffic.json (what bench.py prints as roofline.traffic) is tied to the sha256 of the attention sources it was measured
    on; the committed record must describe the committed kernels -- editing ovg_attn.hip / ovg_attn16.h / ovg_common.h without
    re-taking the PMC passes (tools/validate_r03.sh -> tools/traffic_json.py) fails here instead of printing a stale figure."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    rec = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert rec["attention_source_digest"] == bench.attention_source_digest()
    for S in (8, 64):
        assert rec["global_attn_S%d_bytes_per_launch" % S] >= rec["global_attn_S%d_algorithmic_bytes" % S] == 4 * S * 1374 * 1024 * 2


def test_traffic_json_tool_sums_the_dispatches_of_one_attention_and_stamps_the_digest(tmp_path):
    """tools/traffic_json.py: bytes per global attention = sum over the (kernel, grid) dispatches of the launch plan of
    FETCH_SIZE[KiB] * 1024 * 2 (gfx950 wide-read correction, MI355X_MICROARCH.md) + WRITE_SIZE[KiB] * 1024, from the two separate
    rocprofv3 --pmc passes; kernels that are not the tuned attention (the baseline reference call of the bench tool) are ignored."""
    import csv
    import json
    import sys
    hdr = ["Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value"]
    main, tail = "void attn16_kernel<__bf16, 4, 8, 0, 2, false, 5>(ovg_attn_params)", "void attn16_kernel<__bf16, 2, 4, 0, 2, false, 3>(ovg_attn_params)"
    rows_fetch = [[main, 1310720, "FETCH_SIZE", 1000.0], [main, 1310720, "FETCH_SIZE", 1200.0], [main, 1310720, "TCC_HIT_sum", 96.0],
                  [tail, 192512, "FETCH_SIZE", 300.0], [tail, 192512, "TCC_HIT_sum", 4.0], ["void attn_kernel<float, 1>(x)", 704512, "FETCH_SIZE", 9e9]]
    rows_write = [[main, 1310720, "WRITE_SIZE", 500.0], [main, 1310720, "TCC_MISS_sum", 3.0], [tail, 192512, "WRITE_SIZE", 100.0], [tail, 192512, "TCC_MISS_sum", 1.0]]
    for name, rows in (("p3", rows_fetch), ("p4", rows_write)):
        d = tmp_path / name / "runc"
        d.mkdir(parents=True)
        with open(d / "1_counter_collection.csv", "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(hdr)
            w.writerows(rows)
    out = tmp_path / "traffic.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "traffic_json.py"), "--views", "64", str(tmp_path / "p3"), str(tmp_path / "p4"),
                        "--out", str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.load(open(out))
    want = (1100.0 * 2048 + 500.0 * 1024) + (300.0 * 2048 + 100.0 * 1024)
    assert rec["global_attn_S64_bytes_per_launch"] == round(want)
    assert rec["global_attn_S64_algorithmic_bytes"] == 4 * 64 * 1374 * 1024 * 2
    assert abs(rec["global_attn_S64_l2_hit_rate"] - 100.0 / 104.0) < 1e-3 and len(rec["global_attn_S64_dispatches"]) == 2
    sys.path.insert(0, ROOT)
    import bench
    assert rec["attention_source_digest"] == bench.attention_source_digest()


def test_cu_budget_of_the_sharded_attention_plans(monkeypatch):
    """Round-4 review item 5(ii): the launch plans of the global attention are told how many CUs RCCL's channels leave them
    (sharding.available_cus -> ovg_attn_params.cus) instead of quantising against all 256."""
    from omnivggt_official_amd import ops, sharding
    monkeypatch.delenv("NCCL_MAX_NCHANNELS", raising=False)
    assert sharding.rccl_channels() == 32 and sharding.available_cus(1) == 256 and sharding.available_cus(8) == 224
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "16")
    assert sharding.rccl_channels() == 16 and sharding.available_cus(8) == 240
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "200")                      # a mis-set environment cannot cripple the plan
    assert sharding.available_cus(8) == 192
    # head groups: 2 ranks x 32 views -> 2 groups at either budget; 8 ranks x 8 views -> one launch
    assert len(sharding.head_groups(8, 2, 32 * 1374, cus=224)) == 2 and len(sharding.head_groups(2, 8, 8 * 1374, cus=224)) == 1
    P, bf = 1374, torch.bfloat16
    full = ops.attn_plan(16, 64 * P, [64 * P], bf)
    part = ops.attn_plan(16, 64 * P, [64 * P], bf, cus=224)
    assert (full["q_tile"], full["main_rows"]) == (512, 81920) and (part["q_tile"], part["main_rows"]) == (512, 86016)   # 10 rounds of 256 CUs vs 12 of 224
    assert ops.attn_plan(16, 64 * P, [64 * P], bf, cus=999) == full      # more than the device has: the device count
    # per-rank launch of the 8-GPU run: the split workspace follows the budget it was planned with
    pr = ops.attn_plan(16, 8 * P, [8 * P] * 8, bf, cus=224)
    assert pr["splits"] >= 2 and pr["part_bytes"] == pr["splits"] * 16 * ops.pad_to(8 * P, 64) * 64 * 4


def test_checkpoint_rehearsal_key_check(tmp_path):
    """tools/validate_checkpoint.py (round-4 review item 8): the host-side half -- a synthetic checkpoint with the reference's key set is
    written and accepted; a file with a missing / renamed / reshaped tensor is reported, not loaded."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import validate_checkpoint as V
    from safetensors.torch import load_file, save_file
    path = V.write_synthetic(str(tmp_path / "synth.safetensors"), depth=1)
    kc = V.check_keys(path, depth=1)
    assert kc["ok"] and kc["keys_in_file"] == kc["keys_expected"] == 263
    sd = load_file(path)
    k0 = "aggregator.frame_blocks.0.attn.qkv.weight"
    bad = dict(sd)
    bad["aggregator.frame_blocks.0.attn.qkv_renamed.weight"] = bad.pop(k0)
    bad["aggregator.camera_token"] = bad["aggregator.camera_token"][:, :1].contiguous()
    save_file(bad, str(tmp_path / "bad.safetensors"))
    kb = V.check_keys(str(tmp_path / "bad.safetensors"), depth=1)
    assert not kb["ok"] and kb["missing"] == [k0] and kb["unexpected"] == ["aggregator.frame_blocks.0.attn.qkv_renamed.weight"]
    assert [m[0] for m in kb["shape_mismatch"]] == ["aggregator.camera_token"]
    assert not V.check_keys(path, depth=24)["ok"]                      # a depth-1 file is not the released model


def test_split_f16_sentinel_survives_copy_and_pickle():
    """lib.F32X is compared by identity everywhere; a deep copy of a model, a pickle round trip (torch.save / load,
    multiprocessing) or a second construction must give back the same object."""
    import copy
    import pickle
    assert copy.copy(L.F32X) is L.F32X and copy.deepcopy(L.F32X) is L.F32X
    assert pickle.loads(pickle.dumps(L.F32X)) is L.F32X and type(L.F32X)() is L.F32X
    assert copy.deepcopy({"compute_dtype": L.F32X})["compute_dtype"] is L.F32X
    assert L.is_split(pickle.loads(pickle.dumps([L.F32X]))[0]) and L.dtype_code(copy.deepcopy(L.F32X)) == L.dtype_code(L.F32X)


def test_attention_launch_plan_of_the_baseline_shapes():
    """ovg_attn_plan is a host-only query (256 CUs assumed where no device is visible): the launch plan of every BASELINE shape is pinned
    here so that a plan regression shows up on CPU -- q tile, the tail split (rows of the first launch + tile of the second) and the
    split-KV factor of the per-rank launches of the view-sharded run."""
    from omnivggt_official_amd import ops
    P = 1374
    bf, f16 = torch.bfloat16, torch.float16
    plan = lambda BH, nq, nks, dt, **kw: ops.attn_plan(BH, nq, nks, dt, **kw)
    # single GPU, global attention (16 heads, nq = nk = S * 1374)
    p8 = plan(16, 8 * P, [8 * P], bf)
    assert (p8["q_tile"], p8["tail_q_tile"], p8["main_rows"]) == (256, 0, 8 * P)            # 1.34 rounds: the tail split loses there (-4 %)
    # short global launches (r04 A/B): 128-row tiles where they quantise no worse than 256-row tiles and their last round is not too full
    assert [plan(16, S * P, [S * P], bf)["q_tile"] for S in (3, 4, 5, 6, 7, 8, 12, 13)] == [128, 128, 256, 128, 128, 256, 256, 256]
    assert plan(16, 4 * P, [4 * P], bf)["tail_q_tile"] == 0 and plan(16, 4 * P, [4 * P], bf)["splits"] == 1
    # key-split tail (round 5): a last round that is at most a quarter full -- 13 views, 2.19 rounds of 256-row tiles: 2 rounds (16384 rows) unsplit,
    # the remaining 1478 rows (96 units) as 5 key ranges each + merge, f32 partials sized for the tail rows only; not at 8 (1.34) / 14 (2.38) views
    p13 = plan(16, 13 * P, [13 * P], bf)
    assert (p13["q_tile"], p13["tail_q_tile"], p13["main_rows"], p13["splits"]) == (256, 256, 16384, 5), p13
    assert p13["part_bytes"] == 5 * 16 * 1536 * 64 * 4 and p13["lse_bytes"] == 5 * 16 * 1536 * 4
    assert plan(16, 13 * P, [13 * P], bf, kv_splits=1)["splits"] == 1 and plan(16, 14 * P, [14 * P], bf)["splits"] == 1
    for S in (9, 10):
        p = plan(16, S * P, [S * P], bf)
        assert (p["q_tile"], p["tail_q_tile"], p["main_rows"]) == (256, 128, 8192), (S, p)     # one full round of 512 slots, then 128-row tiles
    # 512-row tiles from 2.5 rounds on: the full rounds (2 at 16 views, 10 at 64 on 256 CUs) unsplit; the rows beyond them as 128-row tiles when
    # the caller gives no workspace (kv_splits = 1), else -- round 6 -- as 512-row tiles cut along the keys so that tail units x ranges fill whole
    # rounds: 16 views 176 units x 4, 64 views 192 units x 8 (f32 partials for the tail rows only, padded to the 512-row tile)
    for S, rows, s, tail_pad in ((16, 16384, 4, 5632), (64, 81920, 8, 6144)):
        p = plan(16, S * P, [S * P], bf, kv_splits=1)
        assert (p["q_tile"], p["tail_q_tile"], p["main_rows"], p["splits"]) == (512, 128, rows, 1), (S, p)
        p = plan(16, S * P, [S * P], bf)
        assert (p["q_tile"], p["tail_q_tile"], p["main_rows"], p["splits"]) == (512, 512, rows, s), (S, p)
        assert p["part_bytes"] == s * 16 * tail_pad * 64 * 4 and p["lse_bytes"] == s * 16 * tail_pad * 4
    assert plan(16, 17 * P, [17 * P], bf)["splits"] == 1                                       # 2.875 rounds: nothing to gain
    p128 = plan(16, 128 * P, [128 * P], f16)
    assert (p128["q_tile"], p128["tail_q_tile"], p128["splits"]) == (256, 0, 1)               # f16: the lazy-rescale kernel, 256-row tiles
    # frame-local attention (S * 16 entries of 1374 rows): the tile that pads the sequence least once the launch is long enough
    assert plan(64 * 16, P, [P], bf)["q_tile"] == 128 and plan(8 * 16, P, [P], bf)["q_tile"] == 256
    # per-rank launches of the 8-GPU run, head-parallel form: 16 (source rank, head) entries x 8 views of queries x 8 segments of keys
    pr = plan(16, 8 * P, [8 * P] * 8, bf)
    assert (pr["splits"], pr["q_tile"], pr["tail_q_tile"]) == (4, 256, 0) and pr["part_bytes"] == 4 * 16 * ops.pad_to(8 * P, 64) * 64 * 4
    # a forced factor is honoured and sized; the baseline kernel (variant 1) never splits
    assert plan(16, 8 * P, [8 * P], bf, kv_splits=3)["splits"] == 3 and plan(16, 8 * P, [8 * P], bf, variant=1)["splits"] == 1


def test_heads_run_in_order_when_not_concurrent():
    """OmniVGGT._run_heads: without a device (or with one job / in the sharded run) the head closures run in the reference's order
    on the caller's stream; the side-stream form needs a GPU (tests/test_gpu_aggregator.py exercises it through the model forward)."""
    from omnivggt_official_amd.model import OmniVGGT
    calls = []
    jobs = [("camera", lambda: calls.append("camera") or ["pose"]), ("depth", lambda: calls.append("depth") or ("d", "dc")),
            ("point", lambda: calls.append("point") or ("p", "pc"))]
    res = OmniVGGT._run_heads(object(), jobs, concurrent=False)
    assert calls == ["camera", "depth", "point"] and res == {"camera": ["pose"], "depth": ("d", "dc"), "point": ("p", "pc")}
    calls.clear()
    assert OmniVGGT._run_heads(object(), jobs[:1], concurrent=True) == {"camera": ["pose"]} and calls == ["camera"]   # one job: nothing to overlap


def test_dpt_tail_patch_swizzle_is_conflict_free_and_consistent():
    """csrc/ovg_dpt_tail.h keeps an 18 x 16-pixel patch in LDS with 256-byte pixels whose 16-byte chunks are XOR-swizzled by the pixel's
    class (col + 6 row) & 7. Host restatement of the kernel's address arithmetic (tile constants parsed from the header): (i) the one-XOR form the
    matrix phase uses equals the direct formula (chunk ^ class), (ii) the class the reader derives per tap equals the class the writer stored
    under, (iii) every 8 consecutive lanes of every B-fragment read (16 consecutive tile pixels, 14-wide rows, any tap, any channel group) hit
    8 different 16-byte bank groups."""
    import re
    src = open(os.path.join(ROOT, "omnivggt-official_amd", "csrc", "ovg_dpt_tail.h")).read()
    m = re.search(r"constexpr int TH = (\d+), TW = (\d+), PR = TH \+ 2, PC = TW \+ 2;", src)
    TH, TW = int(m.group(1)), int(m.group(2))
    PC = TW + 2
    assert (TH * TW) % 16 == 0 and "(pc + 6 * pr) & 7" in src and "(mx[b] + 6 * my[b]) & 7" in src
    for mb in range(TH * TW // 16):
        for ky in range(3):
            for kx in range(3):
                for kc in range(4):
                    addrs = []
                    for lane in range(64):
                        g, lr = lane >> 4, lane & 15
                        pix = 16 * mb + lr
                        y, x = divmod(pix, TW)
                        p0, f0 = y * PC + x, (x + 6 * y) & 7
                        fs = (f0 + kx + 6 * ky) & 7
                        assert fs == ((x + kx) + 6 * (y + ky)) & 7                                   # reader's class == writer's class of that patch pixel
                        tbx = ((p0 + ky * PC + kx) * 256 + (((g ^ fs) & 3) << 4)) ^ ((fs & 4) << 4)
                        a = tbx ^ (kc << 6)
                        assert a == (p0 + ky * PC + kx) * 256 + (((4 * kc + g) ^ fs) << 4)
                        addrs.append(a)
                    for s in range(0, 64, 8):
                        assert len({(a // 16) % 8 for a in addrs[s:s + 8]}) == 8, (mb, ky, kx, kc, s)


def test_dpt_pass_size_follows_free_memory_and_early_levels_pick_their_layers(monkeypatch):
    """Round 6 host logic of the façade: (i) `dpt_frames_chunk` is an upper bound -- the pass actually taken fits half of the free device memory
    (round-5 advisor: 64 frames x two concurrent heads x ~0.28 GB per frame and head OOMs smaller parts), never below the reference's 8 frames;
    (ii) heads_hip.EarlyLevels takes exactly the aggregator layers that feed pyramid levels 0-2 and never the layer that ends the aggregator
    (reduced-depth models clamp several levels onto the last layer: nothing to start early)."""
    from omnivggt_official_amd import heads_hip
    from omnivggt_official_amd.model import OmniVGGT
    with torch.device("meta"):
        m = OmniVGGT(compute_dtype=torch.bfloat16)
        small = OmniVGGT(depth=2, dino_depth=2, compute_dtype=torch.bfloat16)

    class FakeImages:
        is_cuda, device = True, "cuda:0"
        shape = (1, 64, 3, 518, 518)
    for free_gb, concurrent, dt, want in ((250, True, torch.bfloat16, 64), (30, True, torch.bfloat16, 16), (30, False, torch.bfloat16, 32),
                                          (30, True, torch.float32, 8), (2, True, torch.bfloat16, 8)):
        monkeypatch.setattr(torch.cuda, "mem_get_info", lambda dev=None, g=free_gb: (int(g * 1e9), int(288e9)))
        assert m._frames_per_pass(FakeImages, dt, concurrent) == want, (free_gb, concurrent, dt)
    m.dpt_frames_chunk = 8
    assert m._frames_per_pass(FakeImages, torch.bfloat16, True) == 8
    e = heads_hip.EarlyLevels(m._hip_dpt["depth"], 1, 8, 518, 518, 5, torch.bfloat16)
    assert [i for i in range(24) if e.wants(i)] == [4, 11, 17] and e.matches(1, 8, 518, 518, torch.bfloat16) and not e.matches(1, 9, 518, 518, torch.bfloat16)
    e2 = heads_hip.EarlyLevels(small._hip_dpt["depth"], 1, 2, 518, 518, 5, torch.bfloat16)
    assert not any(e2.wants(i) for i in range(2))


def test_bench_parses_rccl_channel_lines(tmp_path, monkeypatch):
    """Round-5 review item 7: bench.py at N > 1 points RCCL's INFO log of every rank at a file and reports the channel counts RCCL
    actually set up (`comm.rccl_channels_observed`) -- what sharding.available_cus() should have subtracted. The parser against the line
    formats of RCCL 2.2x; a job that configured NCCL_DEBUG itself, or a log without the lines, yields a reason instead of numbers."""
    import bench
    log = tmp_path / "rccl_rank3.log"
    monkeypatch.setattr(bench, "RCCL_LOG", str(tmp_path / "rccl_rank%d.log"))
    monkeypatch.setenv("NCCL_DEBUG_FILE", str(log))
    log.write_text("runc:77:77 [3] NCCL INFO RCCL version 2.26.6-HEAD:64f48b6\n"
                   "runc:77:102 [3] NCCL INFO Channel 00/32 : 0 1 2 3 4 5 6 7\n"
                   "runc:77:102 [3] NCCL INFO 32 coll channels, 0 collnet channels, 0 nvls channels, 32 p2p channels, 2 p2p channels per peer\n"
                   "runc:77:140 [3] NCCL INFO 16 coll channels, 0 collnet channels, 0 nvls channels, 16 p2p channels, 2 p2p channels per peer\n")
    assert bench.rccl_channels_observed(3) == {"coll_channels": 32, "p2p_channels": 32, "p2p_channels_per_peer": 2, "version": "2.26.6-HEAD:64f48b6"}
    log.write_text("nothing about channels here\n")
    assert "unavailable" in bench.rccl_channels_observed(3)
    monkeypatch.setenv("NCCL_DEBUG_FILE", "/somewhere/else.log")
    assert "unavailable" in bench.rccl_channels_observed(3)
