"""-m gpu: the HIP aggregator / OmniVGGT facade against the CPU oracle and the committed
golden vectors of the real reference.

Tolerances (max|a-b| / max|b| per tensor):
  f32 parity mode : <= 1e-4 on every aggregator layer and on pose_enc / depth / points
  bf16 / f16 modes: at full depth, against the f32 reference golden AND the reference's own bf16-autocast twin
                    (tests/golden/*_bf16twin.npz): gated at <= 2x the twin's error per tensor (SURVEY 8c Gate 2) --
                    the 1e-4 target is an fp32 statement, the twin itself moves 7e-3..1e-2.
Heads are stock PyTorch (out of kernel scope) and are exercised in two tests only.
"""
import os

import pytest
import torch

import aggregator_oracle as orc
import common
from omnivggt_official_amd import lib as L
from omnivggt_official_amd.model import OmniVGGT

pytestmark = pytest.mark.gpu
DEV = "cuda"
F32_TOL = 1e-4


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    L.require_gpu()
    torch.set_num_threads(min(32, os.cpu_count()))   # 256-thread intra-op on the GPU box is far slower than 32


def build(sd, depth, dino_depth, dtype):
    with torch.device("meta"):
        m = OmniVGGT(depth=depth, dino_depth=dino_depth, compute_dtype=dtype)
    m = m.to_empty(device="cpu")
    m.load_state_dict(sd, strict=True)
    return m.to(DEV).eval()


def run_agg(m, S, dgi, cgi, hw=518):
    inp = common.inputs_for(S, DEV, hw=hw)
    with torch.no_grad():
        return m.aggregator(inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], dgi, cgi)


def run_full(m, S, dgi, cgi, hw=518):
    inp = common.inputs_for(S, DEV, hw=hw)
    with torch.no_grad():
        return m(inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], dgi, cgi)


@pytest.fixture(scope="module")
def reduced():
    sd = common.reduced_state_dict(2, 2)
    return sd, build(sd, 2, 2, torch.float32)


@pytest.mark.parametrize("S,dgi,cgi", [(2, [], []), (2, [1], []), (3, [], [0, 2]), (3, [1], [0, 2]), (2, [0, 1], [0, 1])])
def test_f32_parity_all_modality_combos_depth2(reduced, S, dgi, cgi):
    """depth-2 / DINO-2 aggregator, every modality combination incl. partial / interleaved indices."""
    sd, m = reduced
    inp = orc.synthetic_inputs(S)
    with torch.no_grad():
        ref, _ = orc.aggregator_forward(sd, inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], dgi, cgi,
                                        depth_layers=2, dino_layers=2)
    toks, start = run_agg(m, S, dgi, cgi)
    assert start == 5 and len(toks) == 2
    for l in range(2):
        assert toks[l].shape == (1, S, 1374, 2048) and toks[l].dtype == torch.float32
        assert common.max_rel(toks[l].cpu(), ref[l]) <= F32_TOL


@pytest.mark.parametrize("S,dgi,cgi", [(2, [], []), (3, [1], [0, 2]), (2, [0, 1], [0, 1])])
def test_f32x_parity_modality_combos_depth2(reduced, S, dgi, cgi):
    """The split-f16 mode (compute_dtype = lib.F32X: (hi, lo) f16 planes, three f16 MFMAs per product) at the SAME 1e-4 gate as
    the f32 mode, against the CPU oracle; its error is printed next to the f32 mode's on the same inputs."""
    sd, m32 = reduced
    inp = orc.synthetic_inputs(S)
    with torch.no_grad():
        ref, _ = orc.aggregator_forward(sd, inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], dgi, cgi,
                                        depth_layers=2, dino_layers=2)
    mx = build(sd, 2, 2, L.F32X)
    toks, start = run_agg(mx, S, dgi, cgi)
    t32, _ = run_agg(m32, S, dgi, cgi)
    assert start == 5 and len(toks) == 2
    for l in range(2):
        assert toks[l].shape == (1, S, 1374, 2048) and toks[l].dtype == torch.float32
        e, e32 = common.max_rel(toks[l].cpu(), ref[l]), common.max_rel(t32[l].cpu(), ref[l])
        print("S=%d layer %d max-rel vs oracle: f32x %.2e (f32 mode %.2e)" % (S, l, e, e32))
        assert e <= F32_TOL


@pytest.mark.parametrize("S,dgi,cgi", [(8, [], []), (16, list(range(16)), list(range(16))), (8, [2, 5], [0, 3, 7])])
def test_parity_at_baseline_view_counts_depth2(reduced, S, dgi, cgi):
    """BASELINE configs[1] (8 views, images only), configs[2] (16 views, depth + camera on every view) and an 8-view
    partial-aux case against the CPU oracle at the real token counts (global attention over 10 992 / 21 984 keys),
    depth 2 / DINO 2 so that the oracle finishes in seconds: f32 mode at the 1e-4 gate, then the bf16 mode the bench
    times on the same inputs (gate: 3e-2, the twin's full-depth error; measured 3e-3..6e-3 at this depth)."""
    sd, m = reduced
    inp = orc.synthetic_inputs(S)
    with torch.no_grad():
        ref, _ = orc.aggregator_forward(sd, inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], dgi, cgi,
                                        depth_layers=2, dino_layers=2)
    toks, start = run_agg(m, S, dgi, cgi)
    assert start == 5 and len(toks) == 2
    for l in range(2):
        assert toks[l].shape == (1, S, 1374, 2048)
        err = common.max_rel(toks[l].cpu(), ref[l])
        print("S=%d f32 layer %d max-rel vs oracle %.2e" % (S, l, err))
        assert err <= F32_TOL
    del toks
    mx = build(sd, 2, 2, L.F32X)                      # the split-f16 mode at the same gate
    toks, _ = run_agg(mx, S, dgi, cgi)
    for l in range(2):
        err = common.max_rel(toks[l].cpu(), ref[l])
        print("S=%d f32x layer %d max-rel vs oracle %.2e" % (S, l, err))
        assert err <= F32_TOL
    del toks, mx
    mb = build(sd, 2, 2, torch.bfloat16)
    toks, _ = run_agg(mb, S, dgi, cgi)
    for l in range(2):
        assert torch.isfinite(toks[l]).all()
        err = common.max_rel(toks[l].float().cpu(), ref[l])
        print("S=%d bf16 layer %d max-rel vs oracle %.2e" % (S, l, err))
        assert err <= 3e-2


@pytest.mark.parametrize("hw", [(392, 518), (518, 392), (266, 266)])
def test_f32_parity_other_resolutions_depth2(reduced, hw):
    """non-square / non-trained grids -- resampled pos_embed (bicubic + antialias), gh != gw RoPE
    positions, depth patchify and camera FoV on the real (H, W); f32 parity vs the oracle, then the whole dict."""
    sd, m = reduced
    S, dgi, cgi = 2, [1], [0, 1]
    inp = orc.synthetic_inputs(S, hw=hw)
    with torch.no_grad():
        ref = orc.model_forward(sd, inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], dgi, cgi,
                                depth_layers=2, dino_layers=2)
    out = run_full(m, S, dgi, cgi, hw=hw)
    toks, _ = run_agg(m, S, dgi, cgi, hw=hw)
    P = (hw[0] // 14) * (hw[1] // 14) + 5
    for l in range(2):
        assert toks[l].shape == (1, S, P, 2048)
        assert common.max_rel(toks[l].cpu(), ref["_tokens"][l]) <= F32_TOL
    for key in ("pose_enc", "depth", "depth_conf", "world_points", "world_points_conf"):
        assert out[key].shape == ref[key].shape
        assert common.max_rel(out[key].cpu(), ref[key]) <= F32_TOL, key


def test_low_precision_other_resolution_depth2(reduced):
    """bf16 aggregator + HIP DPT heads on a 392 x 518 input: finite, right shapes, close to the f32 oracle."""
    sd, _ = reduced
    m = build(sd, 2, 2, torch.bfloat16)
    S, dgi, cgi, hw = 2, [1], [0, 1], (392, 518)
    inp = orc.synthetic_inputs(S, hw=hw)
    with torch.no_grad():
        ref = orc.model_forward(sd, inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], dgi, cgi,
                                depth_layers=2, dino_layers=2)
    out = run_full(m, S, dgi, cgi, hw=hw)
    for key in ("depth", "depth_conf", "world_points", "world_points_conf"):
        assert out[key].shape == ref[key].shape and torch.isfinite(out[key]).all()
        err = common.max_rel(out[key].cpu(), ref[key])
        print("bf16 392x518 %s max-rel vs f32 oracle %.2e" % (key, err))
        assert err <= 0.15, key


def test_f32_end_to_end_dict_depth2(reduced):
    """OmniVGGT.forward contract (keys, shapes, values) with the PyTorch heads, vs the oracle."""
    sd, m = reduced
    S, dgi, cgi = 2, [0], [0, 1]
    inp = orc.synthetic_inputs(S)
    with torch.no_grad():
        ref = orc.model_forward(sd, inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], dgi, cgi,
                                depth_layers=2, dino_layers=2)
    out = run_full(m, S, dgi, cgi)
    assert set(out) == {"pose_enc", "pose_enc_list", "depth", "depth_conf", "world_points", "world_points_conf", "images"}
    for key in ("pose_enc", "depth", "depth_conf", "world_points", "world_points_conf"):
        assert out[key].shape == ref[key].shape and out[key].dtype == torch.float32
        assert common.max_rel(out[key].cpu(), ref[key]) <= F32_TOL, key
    assert len(out["pose_enc_list"]) == 4 and out["images"].shape == (1, S, 3, 518, 518)


@pytest.fixture(scope="module")
def full_model():
    return build(common.full_state_dict(), 24, 24, torch.float32)


@pytest.mark.parametrize("name", ["s2_images_only", "s3_partial_aux", "s2_full_aux", "s2_392x518_aux"])
def test_f32_full_depth_vs_reference_golden(full_model, name):
    """Full 24+24+24-block aggregator in f32 parity mode against the REAL reference's tokens
    (the 392 x 518 case goes through the resampled pos_embed and a 28 x 37 RoPE grid: SURVEY 8(f) N2)."""
    full_model.set_compute_dtype(torch.float32)
    S, dgi, cgi, hw = common.case(name)
    toks, start = run_agg(full_model, S, dgi, cgi, hw=hw)
    gold = common.load_golden(name)
    worst = 0.0
    for l in common.TOK_LAYERS:
        e = common.max_rel(common.sample_tokens([t.cpu() for t in toks], l), gold["tok_L%d" % l])
        worst = max(worst, e)
        assert e <= F32_TOL, (l, e)
    absmean = torch.tensor([float(t.abs().mean()) for t in toks])
    assert common.max_rel(absmean, gold["tok_absmean"]) <= F32_TOL
    print("f32 full-depth %s: worst sampled token max-rel %.2e" % (name, worst))


@pytest.mark.parametrize("name", ["s2_images_only", "s3_partial_aux", "s2_392x518_aux"])
def test_f32x_full_depth_vs_reference_golden(full_model, name):
    """The split-f16 mode through all 24 + 24 + 24 blocks and the (exact-f32) heads against the REAL reference's tokens and
    predictions, at the f32 mode's own 1e-4 gate (north star: "outputs within 1e-4 rel of reference")."""
    full_model.set_compute_dtype(L.F32X)
    try:
        S, dgi, cgi, hw = common.case(name)
        out = run_full(full_model, S, dgi, cgi, hw=hw)
        toks, _ = run_agg(full_model, S, dgi, cgi, hw=hw)
    finally:
        full_model.set_compute_dtype(torch.float32)
    gold = common.load_golden(name)
    errs = {}
    for l in common.TOK_LAYERS:
        errs["tok_L%d" % l] = common.max_rel(common.sample_tokens([t.cpu() for t in toks], l), gold["tok_L%d" % l])
    errs["pose_enc"] = common.max_rel(out["pose_enc"].cpu(), gold["pose_enc"])
    errs["depth"] = common.max_rel(out["depth"][0, :, ::37, ::37, 0].cpu(), gold["depth"])
    errs["depth_conf"] = common.max_rel(out["depth_conf"][0, :, ::37, ::37].cpu(), gold["depth_conf"])
    errs["world_points"] = common.max_rel(out["world_points"][0, :, ::37, ::37].cpu(), gold["world_points"])
    print("f32x full-depth %s: %s" % (name, ", ".join("%s %.2e" % kv for kv in errs.items())))
    assert max(errs.values()) <= F32_TOL, errs


@pytest.mark.parametrize("name", ["s3_partial_aux", "s2_392x518_aux"])
def test_f32_full_depth_predictions_vs_reference_golden(full_model, name):
    full_model.set_compute_dtype(torch.float32)
    S, dgi, cgi, hw = common.case(name)
    out = run_full(full_model, S, dgi, cgi, hw=hw)
    gold = common.load_golden(name)
    assert common.max_rel(out["pose_enc"].cpu(), gold["pose_enc"]) <= F32_TOL
    assert common.max_rel(out["depth"][0, :, ::37, ::37, 0].cpu(), gold["depth"]) <= F32_TOL
    assert common.max_rel(out["depth_conf"][0, :, ::37, ::37].cpu(), gold["depth_conf"]) <= F32_TOL
    assert common.max_rel(out["world_points"][0, :, ::37, ::37].cpu(), gold["world_points"]) <= F32_TOL


TWIN_CASES = ("s2_images_only", "s3_partial_aux", "s2_392x518_aux")


def _rms_rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp(min=1e-30))


@pytest.mark.parametrize("name", TWIN_CASES)
def test_low_precision_modes_vs_twin(full_model, name):
    """SURVEY 8c Gate 2: the 16-bit throughput modes at FULL depth (24 + 24 + 24 blocks, HIP DPT heads) against the f32
    reference golden AND against the same-precision twin -- the reference itself under torch.autocast('cpu', bfloat16)
    on the same weights / inputs (oracle/gen_golden_bf16twin.py). Gate: on every aggregator layer and on pose_enc the
    HIP error vs the f32 reference is <= 2x the twin's own error (max-rel AND rms-rel); the dense predictions, which pass
    through the 16-bit DPT heads (a "next" row), <= 3x. The measured table is appended to gpurun_out/lowprec_parity.txt."""
    S, dgi, cgi, hw = common.case(name)
    gold, twin = common.load_golden(name), common.load_golden(name + "_bf16twin")
    keys = ["tok_L%d" % l for l in common.TOK_LAYERS] + ["pose_enc", "depth", "depth_conf", "world_points", "world_points_conf"]
    rows = {k: {"twin": (common.max_rel(twin[k], gold[k]), _rms_rel(twin[k], gold[k]))} for k in keys}
    for dtype, tag in ((torch.bfloat16, "bf16"), (torch.float16, "f16")):
        full_model.set_compute_dtype(dtype)
        full_model.hip_heads = True
        out = run_full(full_model, S, dgi, cgi, hw=hw)
        toks, _ = run_agg(full_model, S, dgi, cgi, hw=hw)
        assert all(torch.isfinite(t).all() for t in toks)
        got = {"tok_L%d" % l: common.sample_tokens([t.cpu() for t in toks], l) for l in common.TOK_LAYERS}
        got["pose_enc"] = out["pose_enc"].cpu()
        got["depth"] = out["depth"][0, :, ::37, ::37, 0].cpu()
        got["depth_conf"] = out["depth_conf"][0, :, ::37, ::37].cpu()
        got["world_points"] = out["world_points"][0, :, ::37, ::37].cpu()
        got["world_points_conf"] = out["world_points_conf"][0, :, ::37, ::37].cpu()
        for k in keys:
            rows[k][tag] = (common.max_rel(got[k], gold[k]), _rms_rel(got[k], gold[k]))
            rows[k][tag + "_vs_twin"] = common.max_rel(got[k], twin[k])
    full_model.set_compute_dtype(torch.float32)
    lines = ["case %s (S=%d, %s): error vs the f32 REFERENCE golden as max-rel / rms-rel; last columns: HIP vs the twin itself" % (name, S, hw),
             "%-18s %-21s %-21s %-21s %-10s %-10s" % ("tensor", "reference@bf16-autocast", "HIP bf16", "HIP f16", "bf16~twin", "f16~twin")]
    for k in keys:
        r = rows[k]
        lines.append("%-18s %.2e / %.2e   %.2e / %.2e   %.2e / %.2e   %.2e   %.2e"
                     % (k, r["twin"][0], r["twin"][1], r["bf16"][0], r["bf16"][1], r["f16"][0], r["f16"][1], r["bf16_vs_twin"], r["f16_vs_twin"]))
    text = "\n".join(lines)
    print(text)
    os.makedirs(os.path.join(common.ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(common.ROOT, "gpurun_out", "lowprec_parity.txt"), "a") as fh:
        fh.write(text + "\n\n")
    for k in keys:
        factor = 2.0 if (k.startswith("tok_") or k == "pose_enc") else 3.0
        for tag in ("bf16", "f16"):
            assert rows[k][tag][0] <= factor * rows[k]["twin"][0], (k, tag, "max-rel", rows[k])
            assert rows[k][tag][1] <= factor * rows[k]["twin"][1], (k, tag, "rms-rel", rows[k])


def test_view_permutation_equivariance_at_bench_size():
    """Size-independent property at the S=8 bench configuration (bf16): permuting views 1..S-1
    permutes the aggregator outputs (view 0 keeps the slot-0 special tokens, aggregator.py:343-366).
    Exact in real arithmetic; in bf16 only the softmax key order changes."""
    sd = common.reduced_state_dict(2, 1)
    m = build(sd, 2, 1, torch.bfloat16)
    S = 8
    inp = common.inputs_for(S, DEV)
    perm = [0, 5, 3, 7, 1, 6, 2, 4]
    with torch.no_grad():
        a, _ = m.aggregator(inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], [2, 5], [0, 3])
        pi = {k: v[:, perm].contiguous() for k, v in inp.items()}
        dgi = sorted([perm.index(2), perm.index(5)])
        cgi = [0, perm.index(3)]
        b, _ = m.aggregator(pi["images"], pi["extrinsics"], pi["intrinsics"], pi["depth"], pi["mask"], dgi, cgi)
    for l in range(2):
        assert torch.isfinite(a[l]).all()
        assert common.max_rel(b[l].cpu(), a[l][:, perm].cpu()) <= 3e-2


def test_packed_weights_roundtrip(tmp_path):
    """save_packed -> from_packed reproduces the forward bit for bit without f32 masters of the GEMM weights
    (the aggregator parameters of the reloaded model stay on the meta device), in half the bytes."""
    sd = common.reduced_state_dict(2, 2)
    m = build(sd, 2, 2, torch.bfloat16)
    S, dgi, cgi = 2, [1], [0, 1]
    ref_toks, _ = run_agg(m, S, dgi, cgi)
    ref_out = run_full(m, S, dgi, cgi)
    path = str(tmp_path / "packed.safetensors")
    m.save_packed(path)
    full = sum(v.numel() * 4 for v in sd.values())
    gemm = sum(v.numel() * 4 for k, v in sd.items() if k.startswith("aggregator.") and
               k.endswith(("attn.qkv.weight", "attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight")))
    # the block GEMM weights are stored in 16 bits (the heads and all vectors stay f32): the file saves half of THEIR bytes
    assert os.path.getsize(path) < full - 0.45 * gemm
    m2 = OmniVGGT.from_packed(path)
    assert m2.aggregator.frame_blocks[0].attn.qkv.weight.is_meta and m2.aggregator.compute_dtype == torch.bfloat16
    toks, start = run_agg(m2, S, dgi, cgi)
    out = run_full(m2, S, dgi, cgi)
    assert start == 5
    for a, b in zip(toks, ref_toks):
        assert torch.equal(a, b)
    for key in ("pose_enc", "depth", "world_points", "depth_conf"):
        assert torch.equal(out[key], ref_out[key])
    with pytest.raises(ValueError):
        OmniVGGT.from_packed(path).aggregator.load_packed({"global_blocks.0.attn.qkv.weight": torch.zeros(1, dtype=torch.float16)}, torch.device(DEV))


def test_rejects_bad_inputs():
    sd = common.reduced_state_dict(1, 1)
    m = build(sd, 1, 1, torch.bfloat16)
    with pytest.raises(ValueError):
        m.aggregator(torch.zeros(1, 2, 4, 518, 518, device=DEV), None, None, None, None, [], [])
    with pytest.raises(AssertionError):        # not a multiple of the patch size (reference: patch_embed.py:72-73)
        m.aggregator(torch.zeros(1, 2, 3, 400, 518, device=DEV), None, None, None, None, [], [])
    with pytest.raises(L.OvgError):
        m.aggregator(torch.zeros(1, 2, 3, 518, 518), None, None, None, None, [], [])


# ----------------------------------------------------------------------------------------------------------------------
# BASELINE configs[3] (the headline: 64 views, images only) and configs[4] (128 views, fp16, cameras on the even views,
# depth on the second half) against the CPU oracle AS A FORWARD (round-2 review: only kernel-level checks and the bench's
# self cross-check existed at these sizes). Depth 1 / DINO 1 so the oracle finishes in ~0.5 / ~2.5 minutes on the GPU
# host's cores: one DINOv2 block, one frame block, camera injection, one global block over 87 936 / 175 872 tokens -- every
# kernel and every launch plan of the timed path (512-row tiles + tail launch, 256^2 GEMMs, split V^T stores, ...) at
# the real shapes. Reference call sites: omnivggt_aggregator.py:130-256, aggregator.py:312-341.
# ----------------------------------------------------------------------------------------------------------------------
def _big_config_parity(S, dgi, cgi, modes):
    sd = common.reduced_state_dict(1, 1)
    inp = orc.synthetic_inputs(S)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(64, os.cpu_count()))
    try:
        with torch.no_grad():
            ref, _ = orc.aggregator_forward(sd, inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], dgi, cgi,
                                            depth_layers=1, dino_layers=1)
    finally:
        torch.set_num_threads(threads)
    ref = ref[0]
    got = {}
    for dtype, tol in modes:
        m = build(sd, 1, 1, dtype)
        toks, start = run_agg(m, S, dgi, cgi)
        assert start == 5 and len(toks) == 1 and toks[0].shape == (1, S, 1374, 2048) and toks[0].dtype == torch.float32
        t = toks[0].cpu()
        assert torch.isfinite(t).all()
        err, rms = common.max_rel(t, ref), _rms_rel(t, ref)
        # the camera token (injection target) and the first / last views separately: the places a wrong table row or a wrong
        # tail tile would show
        cam = common.max_rel(t[0, :, 0], ref[0, :, 0])
        last = common.max_rel(t[0, -1], ref[0, -1])
        print("S=%d %s depth-1 forward vs oracle: max-rel %.2e rms-rel %.2e camera-token %.2e last-view %.2e (gate %.0e)"
              % (S, repr(dtype).replace("torch.", ""), err, rms, cam, last, tol))
        got[dtype] = (err, rms, cam, last)
        del m, toks, t
        torch.cuda.empty_cache()
    for dtype, tol in modes:
        err, rms, cam, last = got[dtype]
        assert err <= tol and cam <= tol and last <= tol and rms <= tol, (dtype, got[dtype])


def test_oracle_parity_headline_64_views_depth1():
    """configs[3] at N = 1: 64 views 518^2 images-only; f32 parity mode <= 1e-4, the bf16 mode the bench times <= 3e-2
    (the twin-calibrated gate of test_parity_at_baseline_view_counts_depth2; measured ~4e-3)."""
    _big_config_parity(64, [], [], [(torch.float32, F32_TOL), (L.F32X, F32_TOL), (torch.bfloat16, 3e-2)])


def test_oracle_parity_stress_128_views_f16_partial_aux_depth1():
    """configs[4] on one GPU: 128 views, cameras on range(0, 128, 2), depth on range(64, 128); f32 <= 1e-4,
    fp16 (lazy-rescale attention kernel, f16 V^T / hidden activations) <= 5e-3 (measured 6-9e-4 at small S)."""
    _big_config_parity(128, list(range(64, 128)), list(range(0, 128, 2)), [(torch.float32, F32_TOL), (L.F32X, F32_TOL), (torch.float16, 5e-3)])


def test_full_depth_8_views_partial_aux_vs_oracle_model_forward():
    """Closes the parity chain at a BASELINE view count (round-3 review): the FULL model -- 24 DINOv2 + 24 frame + 24 global blocks
    and the three heads -- on 8 views 518^2 with depth on views 2, 5 and cameras on views 0, 3, 7 (views with both, one, or no
    auxiliary modality) against oracle.model_forward (the bit-exact CPU restatement of omnivggt.py:20-68 /
    omnivggt_aggregator.py:130-256; ~2 min on the host cores). f32 and split-f16 modes: <= 1e-4 on the tokens of layers
    0 / 4 / 11 / 17 / 23 and on pose / depth / points; bf16 (the timed mode): the twin-calibrated 3e-2 on tokens AND on every prediction
    (round-5 review item 2; measured at 16 views: pose 8.4e-3, depth 5.9e-4, points 1.5e-2, confidences 3e-4)."""
    S, dgi, cgi = 8, [2, 5], [0, 3, 7]
    sd = common.full_state_dict()
    inp = orc.synthetic_inputs(S)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(64, os.cpu_count()))
    try:
        with torch.no_grad():
            ref = orc.model_forward(sd, inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], dgi, cgi)
    finally:
        torch.set_num_threads(threads)
    rtok = [ref["_tokens"][l][0, :, ::7, ::8] for l in common.TOK_LAYERS]
    keys = ("pose_enc", "depth", "depth_conf", "world_points", "world_points_conf")
    rpred = {k: ref[k] for k in keys}
    del ref
    m = build(sd, 24, 24, torch.float32)
    for dtype, tol in ((torch.float32, F32_TOL), (L.F32X, F32_TOL), (torch.bfloat16, 3e-2)):
        m.set_compute_dtype(dtype)
        out = run_full(m, S, dgi, cgi)
        toks, _ = run_agg(m, S, dgi, cgi)
        errs = {"tok_L%d" % l: common.max_rel(toks[l][0, :, ::7, ::8].cpu(), r) for l, r in zip(common.TOK_LAYERS, rtok)}
        for k in keys:
            assert torch.isfinite(out[k]).all(), (dtype, k)
            errs[k] = common.max_rel(out[k].float().cpu(), rpred[k])
        print("8 views partial aux, full depth, %s vs oracle.model_forward: %s" % (repr(dtype).replace("torch.", ""), ", ".join("%s %.2e" % kv for kv in errs.items())))
        assert max(errs.values()) <= tol, (dtype, errs)      # bf16: tokens AND predictions (pose / depth / points / confidences) <= 3e-2
        del out, toks
        torch.cuda.empty_cache()


def test_full_depth_16_views_full_aux_vs_oracle_model_forward():
    """Second full-depth BASELINE case, in the driver-run suite since round 6 (round-5 review item 2): configs[2] -- 16 views 518^2 with
    depth AND camera on every view -- through the full model (24 + 24 + 24 blocks and the three heads) against oracle.model_forward
    (omnivggt_aggregator.py:130-256, omnivggt.py:47-61; ~2 min incl. the oracle on the GPU box's host cores). f32 and split-f16 <= 1e-4
    on the tokens of layers 0 / 4 / 11 / 17 / 23 and on pose / depth / points / confidences; bf16 <= 3e-2 on all of them
    (measured: tokens 8.1e-3, pose 8.4e-3, depth 5.9e-4, points 1.5e-2)."""
    S, dgi, cgi = 16, list(range(16)), list(range(16))
    sd = common.full_state_dict()
    inp = orc.synthetic_inputs(S)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(64, os.cpu_count()))
    try:
        with torch.no_grad():
            ref = orc.model_forward(sd, inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], dgi, cgi)
    finally:
        torch.set_num_threads(threads)
    rtok = [ref["_tokens"][l][0, :, ::7, ::8] for l in common.TOK_LAYERS]
    keys = ("pose_enc", "depth", "depth_conf", "world_points", "world_points_conf")
    rpred = {k: ref[k] for k in keys}
    del ref
    m = build(sd, 24, 24, torch.float32)
    for dtype, tol in ((torch.float32, F32_TOL), (L.F32X, F32_TOL), (torch.bfloat16, 3e-2)):
        m.set_compute_dtype(dtype)
        out = run_full(m, S, dgi, cgi)
        toks, _ = run_agg(m, S, dgi, cgi)
        errs = {"tok_L%d" % l: common.max_rel(toks[l][0, :, ::7, ::8].cpu(), r) for l, r in zip(common.TOK_LAYERS, rtok)}
        for k in keys:
            assert torch.isfinite(out[k]).all(), (dtype, k)
            errs[k] = common.max_rel(out[k].float().cpu(), rpred[k])
        print("16 views full aux, full depth, %s vs oracle.model_forward: %s" % (repr(dtype).replace("torch.", ""), ", ".join("%s %.2e" % kv for kv in errs.items())))
        assert max(errs.values()) <= tol, (dtype, errs)
        del out, toks
        torch.cuda.empty_cache()


# ----------------------------------------------------------------------------------------------------------------------
# Batches of scenes (round-4 review item 2). OmniVGGT.forward takes (B, S, ...) (omnivggt.py:31-32): the global attention runs
# over S * P tokens PER batch entry (aggregator.py:317-318), the depth statistics are per batch entry over all selected views
# (omnivggt_aggregator.py:118-126), the camera normalisation is per batch entry (:85-105), and the modality index lists are
# shared by the batch. The oracle is bit-exact to the reference at B = 2 (judge-verified).
# ----------------------------------------------------------------------------------------------------------------------
def batch_inputs(B, S, device="cpu", hw=518):
    parts = [orc.synthetic_inputs(S, seed=1234 + 1111 * b, hw=hw) for b in range(B)]     # different scenes: different depth means, cameras
    return {k: torch.cat([q[k] for q in parts], 0).to(device) for k in parts[0]}


@pytest.mark.parametrize("S,dgi,cgi", [(2, [1], [0, 1]), (3, [0, 2], [0, 2]), (3, [], [])])
def test_batch_of_two_scenes_depth2(reduced, S, dgi, cgi):
    """B = 2 x S = 2 / 3 with different modality indices through the depth-2 aggregator: f32 and split-f16 <= 1e-4, bf16 <= 3e-2 vs
    orc.aggregator_forward on the same (2, S, ...) tensors; the second batch entry is also compared with a B = 1 run of its own scene
    (batch entries must not see each other: per-batch depth mean, per-batch camera frame, per-batch global attention)."""
    sd, m = reduced
    B = 2
    cpu = batch_inputs(B, S)
    with torch.no_grad():
        ref, _ = orc.aggregator_forward(sd, cpu["images"], cpu["extrinsics"], cpu["intrinsics"], cpu["depth"], cpu["mask"], dgi, cgi,
                                        depth_layers=2, dino_layers=2)
    dev = {k: v.to(DEV) for k, v in cpu.items()}
    second = {k: v[1:2].contiguous() for k, v in dev.items()}
    for dtype, tol in ((torch.float32, F32_TOL), (L.F32X, F32_TOL), (torch.bfloat16, 3e-2)):
        mm = m if dtype is torch.float32 else build(sd, 2, 2, dtype)
        with torch.no_grad():
            toks, start = mm.aggregator(dev["images"], dev["extrinsics"], dev["intrinsics"], dev["depth"], dev["mask"], dgi, cgi)
            solo, _ = mm.aggregator(second["images"], second["extrinsics"], second["intrinsics"], second["depth"], second["mask"], dgi, cgi)
        assert start == 5 and len(toks) == 2
        for l in range(2):
            assert toks[l].shape == (B, S, 1374, 2048) and toks[l].dtype == torch.float32
            assert torch.isfinite(toks[l]).all()
            errs = [common.max_rel(toks[l][b].cpu(), ref[l][b]) for b in range(B)]
            iso = common.max_rel(toks[l][1].cpu(), solo[l][0].cpu())
            print("B=2 S=%d %s layer %d max-rel vs oracle per batch entry %s; entry 1 vs its own B=1 run %.2e"
                  % (S, repr(dtype).replace("torch.", ""), l, ["%.2e" % e for e in errs], iso))
            assert max(errs) <= tol
            assert iso <= (1e-5 if dtype is not torch.bfloat16 else 3e-2)       # same kernels, different launch geometry: rounding-level


def test_batch_of_two_scenes_full_depth_model_forward():
    """B = 2 x S = 2 (depth on view 1, cameras on both) through the FULL model -- 24 + 24 + 24 blocks and the three heads -- against
    oracle.model_forward (omnivggt.py:20-68): f32 and split-f16 <= 1e-4, bf16 <= 3e-2, on tokens and on every prediction."""
    B, S, dgi, cgi = 2, 2, [1], [0, 1]
    sd = common.full_state_dict()
    cpu = batch_inputs(B, S)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(64, os.cpu_count()))
    try:
        with torch.no_grad():
            ref = orc.model_forward(sd, cpu["images"], cpu["extrinsics"], cpu["intrinsics"], cpu["depth"], cpu["mask"], dgi, cgi)
    finally:
        torch.set_num_threads(threads)
    rtok = [ref["_tokens"][l][:, :, ::7, ::8] for l in common.TOK_LAYERS]
    keys = ("pose_enc", "depth", "depth_conf", "world_points", "world_points_conf")
    rpred = {k: ref[k] for k in keys}
    del ref
    dev = {k: v.to(DEV) for k, v in cpu.items()}
    m = build(sd, 24, 24, torch.float32)
    for dtype, tol in ((torch.float32, F32_TOL), (L.F32X, F32_TOL), (torch.bfloat16, 3e-2)):
        m.set_compute_dtype(dtype)
        with torch.no_grad():
            out = m(dev["images"], dev["extrinsics"], dev["intrinsics"], dev["depth"], dev["mask"], dgi, cgi)
            toks, _ = m.aggregator(dev["images"], dev["extrinsics"], dev["intrinsics"], dev["depth"], dev["mask"], dgi, cgi)
        errs = {"tok_L%d" % l: common.max_rel(toks[l][:, :, ::7, ::8].cpu(), r) for l, r in zip(common.TOK_LAYERS, rtok)}
        for k in keys:
            assert out[k].shape[0] == B and torch.isfinite(out[k]).all(), (dtype, k)
            errs[k] = max(common.max_rel(out[k][b].float().cpu(), rpred[k][b]) for b in range(B))
        print("B=2 S=2 full depth, %s vs oracle.model_forward: %s" % (repr(dtype).replace("torch.", ""), ", ".join("%s %.2e" % kv for kv in errs.items())))
        assert max(errs.values()) <= tol, (dtype, errs)      # bf16: tokens AND predictions (pose / depth / points / confidences) <= 3e-2
        del out, toks
        torch.cuda.empty_cache()


def test_duplicate_camera_indices_follow_the_reference(reduced):
    """the reference tolerates a view listed twice in camera_gt_index (index_select + index assignment,
    omnivggt_aggregator.py:158-178): the duplicate counts twice in the mean camera distance of normalize_extrinsics (:85-105) and the
    scatter writes the same row twice. The HIP path used to refuse such a list; now it reproduces the reference (via the oracle)."""
    sd, m = reduced
    S, dgi, cgi = 3, [1], [0, 1, 2, 2]
    inp = orc.synthetic_inputs(S)
    with torch.no_grad():
        ref, _ = orc.aggregator_forward(sd, inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], dgi, cgi,
                                        depth_layers=2, dino_layers=2)
        ref_nodup, _ = orc.aggregator_forward(sd, inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], dgi, [0, 1, 2],
                                              depth_layers=2, dino_layers=2)
    assert common.max_rel(ref[1], ref_nodup[1]) > 1e-4            # the duplicate DOES change the result (scale of the translations)
    toks, _ = run_agg(m, S, dgi, cgi)
    for l in range(2):
        err = common.max_rel(toks[l].cpu(), ref[l])
        print("camera_gt_index [0, 1, 2, 2] layer %d max-rel vs oracle %.2e" % (l, err))
        assert err <= F32_TOL


def test_fp16_mode_with_outlier_activations():
    """fp16 range safety (SURVEY section 7 hard part; DINOv2-reg high-norm tokens, layers/vision_transformer.py:214-271):
    weights that reproduce the massive-activation pattern -- a few residual channels at |x| ~ 3e2..1e3 from the first DINOv2
    block on AND from the first frame block on (the latter reach the outputs), register tokens two orders above the patch
    tokens, hidden units of the first frame block at ~4e3 -- through the
    f16 mode at 8 views. The f32 mode of the same library (oracle-proven) is the reference: f16 must stay finite and within
    2x the error of the bf16 mode (whose exponent range cannot overflow) on every layer."""
    sd = {k: v.clone() for k, v in common.reduced_state_dict(2, 2).items()}
    g = torch.Generator().manual_seed(7)
    hot = torch.randperm(1024, generator=g)[:6]
    sd["aggregator.patch_embed.blocks.0.mlp.fc2.bias"][hot] = torch.tensor([300.0, -450.0, 800.0, -1000.0, 250.0, 600.0])
    sd["aggregator.register_token"] *= 0
    sd["aggregator.register_token"] += 40.0 * torch.randn(sd["aggregator.register_token"].shape, generator=g)
    units = torch.randperm(4096, generator=g)[:8]
    sd["aggregator.frame_blocks.0.mlp.fc1.bias"][units] = 4000.0
    sd["aggregator.frame_blocks.0.mlp.fc2.weight"][:, units] *= 1e-2      # keep their contribution O(10): the point is the f16 STORE of 4e3
    # the DINOv2 outliers above are renormalised by the backbone's final LayerNorm; these live in the AA trunk's residual stream
    # (LayerScale gamma ~ 1 in the synthetic weights) and therefore in every later LayerNorm / GEMM input and in the outputs
    hot2 = torch.randperm(1024, generator=g)[:4]
    sd["aggregator.frame_blocks.0.mlp.fc2.bias"][hot2] = torch.tensor([400.0, -700.0, 1000.0, -300.0])
    S, dgi, cgi = 8, [1, 6], [0, 4]
    outs = {}
    for dtype in (torch.float32, torch.bfloat16, torch.float16):
        m = build(sd, 2, 2, dtype)
        toks, _ = run_agg(m, S, dgi, cgi)
        outs[dtype] = [t.cpu() for t in toks]
        del m, toks
        torch.cuda.empty_cache()
    ref = outs[torch.float32]
    assert float(ref[0].abs().max()) > 2e2, "the fixture no longer produces outlier activations"
    for l in range(2):
        assert torch.isfinite(outs[torch.float16][l]).all() and torch.isfinite(outs[torch.bfloat16][l]).all()
        e16, eb = common.max_rel(outs[torch.float16][l], ref[l]), common.max_rel(outs[torch.bfloat16][l], ref[l])
        r16, rb = _rms_rel(outs[torch.float16][l], ref[l]), _rms_rel(outs[torch.bfloat16][l], ref[l])
        print("outlier fixture layer %d (max|x| %.0f): f16 max-rel %.2e rms-rel %.2e | bf16 max-rel %.2e rms-rel %.2e"
              % (l, float(ref[l].abs().max()), e16, r16, eb, rb))
        assert e16 <= 2 * max(eb, 1e-3) and r16 <= 2 * max(rb, 1e-3)


def test_camera_tables_entry_vs_host_twin():
    """ovg_camera_tables (selection, normalisation, pose encoding, 25 x Linear(9 -> 1024), 25 adapters as one batched exact-f32
    GEMM over the camera rows, bias rows elsewhere; no host round trip) against the same arithmetic in host PyTorch
    (camera_math.py + F.linear), B = 1 and 2, one camera / all / interleaved subsets, 3 .. 128 views, non-square frames."""
    import torch.nn.functional as Fn
    from omnivggt_official_amd import camera_math, ops
    G = 25
    g = torch.Generator().manual_seed(11)
    pose_w = torch.randn(G * 1024, 9, generator=g) * 0.3
    pose_b = torch.randn(G * 1024, generator=g) * 0.1
    adapt_w = torch.randn(G, 1024, 1024, generator=g) * 0.03
    adapt_b = torch.randn(G, 1024, generator=g) * 0.1
    dev = [t.to(DEV) for t in (pose_w, pose_b, adapt_w, adapt_b)]
    for B, S, idx, hw in ((1, 3, [0, 2], (518, 518)), (1, 8, [5], (518, 518)), (2, 5, [1, 0, 4], (392, 518)),
                          (1, 128, list(range(0, 128, 2)), (518, 518)), (1, 70, list(range(70)), (266, 350))):
        inp = [orc.synthetic_inputs(S, seed=100 + b, hw=hw) for b in range(B)]
        ext = torch.cat([i["extrinsics"] for i in inp])
        intr = torch.cat([i["intrinsics"] for i in inp])
        sel = torch.tensor(idx)
        enc = camera_math.pose_encoding(camera_math.normalize_extrinsics(torch.index_select(ext, 1, sel)), torch.index_select(intr, 1, sel), hw)
        want = adapt_b.unsqueeze(1).repeat(1, B * S, 1)
        rows = (torch.arange(B).unsqueeze(1) * S + sel.unsqueeze(0)).reshape(-1)
        for t in range(G):
            emb = Fn.linear(enc.reshape(-1, 9), pose_w[t * 1024:(t + 1) * 1024], pose_b[t * 1024:(t + 1) * 1024])
            want[t, rows] = Fn.linear(emb, adapt_w[t], adapt_b[t])
        got = ops.camera_tables(ext.to(DEV), intr.to(DEV), torch.tensor(idx, dtype=torch.int32, device=DEV), S, hw, *dev)
        torch.cuda.synchronize()
        assert got.shape == want.shape
        err = common.max_rel(got.cpu(), want)
        print("camera tables B=%d S=%d Sc=%d: max-rel vs host twin %.2e" % (B, S, len(idx), err))
        assert err <= 2e-6
        other = [v for v in range(S) if v not in idx]
        if other:
            assert torch.equal(got[:, other[0]].cpu(), adapt_b)             # a view without a camera holds the adapter bias, exactly
    none = ops.camera_tables(None, None, None, 6, (518, 518), *dev)
    assert torch.equal(none.cpu(), adapt_b.unsqueeze(1).repeat(1, 6, 1))


def test_forced_split_kv_on_the_single_gpu_path():
    """agg.attn_kv_splits in 2..8 on the unsharded path used to ask the plan for the AUTOMATIC factor while the
    launch carried the forced one (no scratch -> OVG_E_ARG, or an undersized one -> out-of-bounds partials). The workspace is
    now sized for the factor the launch carries, its byte counts travel with the pointers, and an undersized buffer is refused."""
    from omnivggt_official_amd import ops
    sd = common.reduced_state_dict(1, 1)
    m = build(sd, 1, 1, torch.bfloat16)
    S = 4
    ref, _ = run_agg(m, S, [], [1])
    for splits in (2, 3, 8):
        m.aggregator.attn_kv_splits = splits
        got, _ = run_agg(m, S, [], [1])
        err = common.max_rel(got[0].cpu(), ref[0].cpu())
        print("forced split-KV x%d on the unsharded forward: max-rel vs unsplit %.2e" % (splits, err))
        assert err <= 1e-2
    m.aggregator.attn_kv_splits = 0
    # an undersized split workspace is an error code, not an overrun
    BH, n = 16, 2 * 1374
    q, k, vt = ops.alloc_qkv(BH, n, n, torch.bfloat16, DEV)
    plan = ops.attn_plan(BH, n, [n], torch.bfloat16, 0, 4)
    assert plan["splits"] == 4
    part, lse = ops.alloc_split_ws(plan, DEV)
    ops.flash_attn(q, [(k, vt, n)], n, torch.bfloat16, kv_splits=4, split_ws=(part, lse))
    with pytest.raises(L.OvgError):
        ops.flash_attn(q, [(k, vt, n)], n, torch.bfloat16, kv_splits=4, split_ws=(part[: part.numel() // 2], lse))
    with pytest.raises(L.OvgError):
        ops.flash_attn(q, [(k, vt, n)], n, torch.bfloat16, kv_splits=4, split_ws=(part, lse[: lse.numel() // 2]))
    torch.cuda.synchronize()


def test_early_dpt_levels_are_bit_identical_to_the_late_order():
    """Round 6: OmniVGGT.forward starts the DPT pyramid levels that read aggregator layers 4 / 11 / 17 from a per-layer hook, on the heads' side
    streams, while the aggregator is still running (heads_hip.EarlyLevels). Same kernels, same inputs, same order per level: every prediction
    must be bit-identical to the forward that runs the whole head behind the last block -- in the bf16 mode (16-bit heads), in the split-f16
    mode (exact-f32 heads), for a batch of two scenes, and with the hook left uninstalled afterwards."""
    sd = common.full_state_dict()
    m = build(sd, 24, 24, torch.bfloat16)
    keys = ("pose_enc", "depth", "depth_conf", "world_points", "world_points_conf")
    for dtype in (torch.bfloat16, L.F32X):
        m.set_compute_dtype(dtype)
        for B, S, dgi, cgi in ((1, 3, [1], [0, 2]), (2, 2, [], [0])):
            parts = [orc.synthetic_inputs(S, seed=77 + b) for b in range(B)]
            inp = {k: torch.cat([q[k] for q in parts], 0).to(DEV) for k in parts[0]}
            outs = {}
            for early in (True, False):
                m.early_dpt_levels = early
                with torch.no_grad():
                    outs[early] = m(inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], dgi, cgi)
                torch.cuda.synchronize()
                assert m.aggregator.layer_hook is None
            for k in keys:
                assert torch.equal(outs[True][k], outs[False][k]), (dtype, B, S, k)
            del outs
            torch.cuda.empty_cache()
    m.early_dpt_levels = True
