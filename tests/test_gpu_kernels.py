"""-m gpu: every HIP kernel, called through the C ABI, against a PyTorch CPU fp32 evaluation
of the same op on the same dtype-rounded inputs (tolerances in gpu_selftest.TOL: bf16 2e-2,
f16 4e-3, f32 2e-5 max-rel; f32 GEMM/attention additionally <= 5e-5 through a whole block)."""
import pytest
import torch

import gpu_selftest as st
from omnivggt_official_amd import lib as L

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _need_gpu():
    L.require_gpu()
    st.results.clear()


def _assert_clean():
    bad = [r for r in st.results if not r["ok"]]
    assert st.results and not bad, bad


def test_mfma_lane_maps():
    st.test_probe()
    _assert_clean()


def test_layernorm():
    st.test_layernorm()
    _assert_clean()


def test_linear_epilogues():
    st.test_linear(False)
    _assert_clean()


def test_qkv_qknorm_rope():
    st.test_qkv(False)
    _assert_clean()


def test_flash_attention_incl_ragged_segments_and_rescale():
    st.test_attn(False)
    _assert_clean()


def test_embed_kernels():
    st.test_embed()
    _assert_clean()


def test_block_forward():
    st.test_block(False)
    _assert_clean()


def test_dpt_head_kernels_and_whole_head():
    st.test_heads(False)
    _assert_clean()


def test_camera_head_entry_vs_twin_and_f32_module():
    """SURVEY 8(f) N1 remainder: the whole iterative camera head on ovg_camera_head (split-K weight-stream GEMMs)."""
    st.test_camera_head()
    _assert_clean()


def test_gemm256_kernels_every_epilogue_ragged_m():
    """The 256 x 256 GEMMs the 64-view bench runs (qkv256_kernel, linear256_kernel<GELU|RES|STORE|PATCH>; free-running main loop since round 6)."""
    st.test_gemm256(False)
    _assert_clean()


def test_history_variants_are_not_in_the_product_library():
    """Round-4 review, hygiene: the A/B history (r02 epilogue forms, attention variants of rounds 1-3) is compiled only into
    tools/probes/build_alt.py builds (-DOVG_AB_VARIANTS); the product library refuses them by name. The round-5 lab GEMM selectors
    (4 = persistent 256 x 256, 10 = DMA-in-M) left the ABI in round 6: they are plain argument errors now."""
    import torch
    from omnivggt_official_amd import ops
    x = torch.zeros(512, 1024, device="cuda", dtype=torch.bfloat16)
    w = torch.zeros(1024, 1024, device="cuda", dtype=torch.bfloat16)
    for tile in (L.TILE_256X, L.TILE_128X):
        with pytest.raises(L.OvgError, match="UNSUPPORTED"):
            ops.linear(x, w, None, torch.bfloat16, tile=tile)
    for tile in (4, 10, 42):
        with pytest.raises(L.OvgError, match="ARG"):
            ops.linear(x, w, None, torch.bfloat16, tile=tile)
    q, k, vt = ops.alloc_qkv(16, 256, 256, torch.bfloat16, "cuda")
    for variant in (6, 21, 33, 51, 59):
        with pytest.raises(L.OvgError, match="UNSUPPORTED"):
            ops.flash_attn(q, [(k, vt, 256)], 256, torch.bfloat16, variant=variant)


def test_global_attention_at_bench_key_counts():
    """N = 10 992 / 21 984 in full, N = 87 936 on sampled rows: the launches the bench times."""
    st.test_attn_big(False)
    _assert_clean()


def test_attention_lse_output_and_merge_and_weight_pack():
    st.test_attn_lse_merge()
    _assert_clean()


def test_speculative_softmax_fallback_counter_with_attention_sinks():
    """fallback_count telemetry: exact results with attention-sink logits, 0 re-runs inside the anchored pass's exponent window,
    every workgroup reported beyond it."""
    st.test_attn_fallback_counter()
    _assert_clean()


def test_split_f16_mode_kernels_vs_float64():
    """OVG_F16X2 ("f32x", the <= 1e-4 mode with throughput): pack / LayerNorm / im2col splits, every GEMM epilogue on both tile sizes,
    the fused QKV epilogue and the three-MFMA flash attention (segments, ragged tails, forced rescale, log-sum-exp), each within 1e-5
    max-rel of a float64 evaluation of the same f32 inputs."""
    st.test_f32x(False)
    _assert_clean()


def test_loaded_library_is_the_in_tree_one():
    import os
    assert os.path.samefile(L.LIB_PATH, os.path.join(os.path.dirname(L.__file__), "libomnivggt_hip.so"))
    maps = open("/proc/self/maps").read()
    assert "libomnivggt_hip.so" in maps
