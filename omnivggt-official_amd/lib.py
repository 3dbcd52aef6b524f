"""ctypes binding of libomnivggt_hip.so (the C ABI declared in include/omnivggt_hip.h).

This is the ONLY way the Python host reaches compute: there is no eager/PyTorch
fallback.  Importing this module never needs a GPU (the library loads on a CPU-only
host so the symbol table can be checked); calling a kernel without one fails loudly.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libomnivggt_hip.so")

OVG_BF16, OVG_F16, OVG_F32, OVG_F16X2 = 0, 1, 2, 3
EPI_STORE, EPI_GELU, EPI_RES, EPI_PATCH = 0, 1, 2, 3
OVG_MAX_SEG = 8
KV_TILE = 64
ABI_VERSION = 11
TILE_AUTO, TILE_128, TILE_256 = 0, 1, 2
ATTN_F32X_FAST_PV = 92                                         # ovg_attn_params.variant in the split-f16 mode (opt-in): PV without P_lo x V_hi, +16 % at 3e-5 .. 1e-4 instead of 1e-5 .. 5e-5
TILE_R02_EPILOGUE, TILE_128X, TILE_256X = 16, 17, 18      # A/B flag (r02 epilogue forms) OR-ed onto a tile selector

ERRORS = {0: "OVG_OK", -1: "OVG_E_ARG", -2: "OVG_E_DTYPE", -3: "OVG_E_LAUNCH", -4: "OVG_E_UNSUPPORTED"}

vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float


class LayerNormParams(C.Structure):
    _fields_ = [("x", vp), ("ldx", i64), ("y", vp), ("ldy", i64), ("weight", vp), ("bias", vp),
                ("rows", i64), ("eps", f32), ("dtype", i32), ("out_f32", i32), ("y_lo", vp)]


class LinearParams(C.Structure):
    _fields_ = [("x", vp), ("ldx", i64), ("w", vp), ("ldw", i64), ("bias", vp), ("y", vp), ("ldy", i64),
                ("M", i64), ("N", i64), ("K", i64), ("dtype", i32), ("epilogue", i32), ("out_f32", i32),
                ("res", vp), ("ldres", i64), ("gamma", vp), ("inject", vp), ("inj_period", i64),
                ("table", vp), ("p0", i64), ("p1", i64), ("row_off", i64), ("tile", i32),
                ("x_lo", vp), ("w_lo", vp), ("y_lo", vp)]


class QkvParams(C.Structure):
    _fields_ = [("x", vp), ("ldx", i64), ("w", vp), ("bias", vp), ("q", vp), ("k", vp), ("vt", vp),
                ("M", i64), ("seq", i64), ("nq_pad", i64), ("nk_pad", i64), ("dtype", i32),
                ("qk_norm", i32), ("qn_w", vp), ("qn_b", vp), ("kn_w", vp), ("kn_b", vp), ("qk_eps", f32),
                ("rope", i32), ("rope_cos", vp), ("rope_sin", vp), ("max_pos", i32),
                ("tokens_per_view", i64), ("grid_w", i32), ("n_special", i32), ("q_scale", f32), ("part", i32), ("tile", i32),
                ("x_lo", vp), ("w_lo", vp), ("q_lo", vp), ("k_lo", vp), ("vt_lo", vp)]


class KvSegment(C.Structure):
    _fields_ = [("k", vp), ("vt", vp), ("nk", i64), ("nk_pad", i64), ("k_lo", vp), ("vt_lo", vp)]


class AttnParams(C.Structure):
    _fields_ = [("q", vp), ("nq", i64), ("nq_pad", i64), ("seg", KvSegment * OVG_MAX_SEG), ("nseg", i32),
                ("out", vp), ("ldo", i64), ("BH", i64), ("dtype", i32), ("variant", i32),
                ("kv_heads", i32), ("out_bh_stride", i64), ("lse", vp), ("kv_splits", i32), ("ws_part", vp), ("ws_lse", vp),
                ("ws_part_bytes", i64), ("ws_lse_bytes", i64), ("q_lo", vp), ("out_lo", vp), ("fallback_count", vp), ("cus", i32)]


class AttnPlanOut(C.Structure):
    _fields_ = [("splits", i32), ("q_tile", i32), ("part_bytes", i64), ("lse_bytes", i64), ("main_rows", i64), ("tail_q_tile", i32)]


class AttnMergeParams(C.Structure):
    _fields_ = [("a", vp), ("lda", i64), ("lse_a", vp), ("b", vp), ("ldb", i64), ("lse_b", vp),
                ("out", vp), ("ldo", i64), ("rows", i64), ("n_pad", i64), ("dtype", i32), ("a_lo", vp), ("b_lo", vp), ("out_lo", vp)]


class BlockWeights(C.Structure):
    _fields_ = [("n1_w", vp), ("n1_b", vp), ("qkv_w", vp), ("qkv_b", vp),
                ("qn_w", vp), ("qn_b", vp), ("kn_w", vp), ("kn_b", vp),
                ("proj_w", vp), ("proj_b", vp), ("ls1", vp), ("n2_w", vp), ("n2_b", vp),
                ("fc1_w", vp), ("fc1_b", vp), ("fc2_w", vp), ("fc2_b", vp), ("ls2", vp),
                ("qkv_w_lo", vp), ("proj_w_lo", vp), ("fc1_w_lo", vp), ("fc2_w_lo", vp)]


class BlockParams(C.Structure):
    _fields_ = [("w", BlockWeights), ("x_in", vp), ("ld_in", i64), ("x_out", vp), ("ld_out", i64),
                ("M", i64), ("seq", i64), ("BH", i64), ("nq_pad", i64), ("nk_pad", i64),
                ("dtype", i32), ("ln_eps", f32), ("qk_norm", i32), ("rope", i32), ("qk_eps", f32),
                ("rope_cos", vp), ("rope_sin", vp), ("max_pos", i32),
                ("tokens_per_view", i64), ("grid_w", i32), ("n_special", i32),
                ("inject", vp), ("inj_period", i64),
                ("ws_xn", vp), ("ws_q", vp), ("ws_k", vp), ("ws_vt", vp), ("ws_attn", vp), ("ws_hid", vp),
                ("extra", KvSegment * OVG_MAX_SEG), ("nseg_extra", i32), ("local_seg_index", i32),
                ("attn_variant", i32), ("qkv_part", i32), ("ev_attn_start", vp), ("ev_attn_stop", vp),
                ("skip_attention", i32), ("gemm_tile", i32), ("ws_attn_part", vp), ("ws_attn_lse", vp), ("attn_kv_splits", i32),
                ("ws_attn_part_bytes", i64), ("ws_attn_lse_bytes", i64),
                ("ws_xn_lo", vp), ("ws_q_lo", vp), ("ws_k_lo", vp), ("ws_vt_lo", vp), ("ws_attn_lo", vp), ("ws_hid_lo", vp),
                ("attn_fallback_count", vp), ("attn_cus", i32)]


class BlockWorkspace(C.Structure):
    _fields_ = [("xn", i64), ("q", i64), ("k", i64), ("vt", i64), ("attn", i64), ("hid", i64), ("total", i64)]


class PackWeightsParams(C.Structure):
    _fields_ = [("src", vp), ("lds", i64), ("dst", vp), ("ldd", i64), ("rows", i64), ("k", i64), ("k_pad", i64), ("dtype", i32), ("dst_lo", vp)]


class Im2colParams(C.Structure):
    _fields_ = [("img", vp), ("img2", vp), ("out", vp), ("k_pad", i64), ("V", i64), ("C", i32), ("Hpx", i32),
                ("Wpx", i32), ("dtype", i32), ("mode", i32), ("mean", f32 * 3), ("std", f32 * 3),
                ("depth_stats", vp), ("views_per_batch", i64), ("out_lo", vp)]


class DepthStatsParams(C.Structure):
    _fields_ = [("depth", vp), ("mask", vp), ("B", i64), ("n_per_batch", i64), ("stats", vp), ("partial", vp),
                ("nblocks", i32)]


class DinoSpecialsParams(C.Structure):
    _fields_ = [("x", vp), ("ldx", i64), ("V", i64), ("tokens_per_view", i64), ("cls", vp), ("pos0", vp),
                ("reg", vp), ("n_reg", i32)]


class AssembleParams(C.Structure):
    _fields_ = [("xd", vp), ("ldxd", i64), ("norm_w", vp), ("norm_b", vp), ("eps", f32),
                ("camera_token", vp), ("register_token", vp), ("cam_add", vp), ("depth_tok", vp),
                ("depth_row", vp), ("placeholder", vp), ("out", vp), ("ldo", i64),
                ("V", i64), ("S", i64), ("tokens_per_view", i64), ("n_special", i32), ("view0", i64)]


class CopyRowsParams(C.Structure):
    _fields_ = [("x", vp), ("ldx", i64), ("y", vp), ("ldy", i64), ("rows", i64), ("n", i64)]


class HeadLayerNormParams(C.Structure):
    _fields_ = [("x", vp), ("ldx", i64), ("y", vp), ("ldy", i64), ("weight", vp), ("bias", vp),
                ("rows", i64), ("p0", i64), ("p1", i64), ("row_off", i64), ("eps", f32), ("dtype", i32)]


class ConvParams(C.Structure):
    _fields_ = [("x", vp), ("ldx", i64), ("w", vp), ("bias", vp), ("y", vp), ("ldy", i64),
                ("add1", vp), ("ld1", i64), ("add2", vp), ("ld2", i64), ("pos_x", vp), ("pos_y", vp),
                ("n_img", i64), ("H", i32), ("W", i32), ("Cin", i32), ("Cout", i32), ("w_rows", i32), ("ksize", i32),
                ("stride", i32), ("upshuffle", i32), ("relu", i32), ("out_f32", i32), ("dtype", i32)]


class UpsampleParams(C.Structure):
    _fields_ = [("x", vp), ("ldx", i64), ("y", vp), ("ldy", i64), ("pos_x", vp), ("pos_y", vp),
                ("n_img", i64), ("H", i32), ("W", i32), ("OH", i32), ("OW", i32), ("C", i32), ("dtype", i32)]


class DptOutParams(C.Structure):
    _fields_ = [("h", vp), ("w2", vp), ("b2", vp), ("val", vp), ("conf", vp), ("npix", i64), ("out_dim", i32), ("activation", i32)]


class DptTailParams(C.Structure):
    _fields_ = [("x", vp), ("ldx", i64), ("pos_x", vp), ("pos_y", vp), ("w1", vp), ("ldw1", i64), ("b1", vp), ("w2", vp), ("b2", vp),
                ("val", vp), ("conf", vp), ("n_img", i64), ("H", i32), ("W", i32), ("OH", i32), ("OW", i32), ("C", i32),
                ("out_dim", i32), ("activation", i32), ("dtype", i32)]


class HeadsToTokensParams(C.Structure):
    _fields_ = [("x", vp), ("n_pad", i64), ("y", vp), ("ldy", i64), ("n", i64), ("heads", i32), ("dtype", i32)]


class UnprojectParams(C.Structure):
    _fields_ = [("depth", vp), ("cam", vp), ("out", vp), ("S", i64), ("H", i32), ("W", i32)]


CAMERA_MAX_TRUNK = 4


class CameraBlockWeights(C.Structure):
    _fields_ = [("n1_w", vp), ("n1_b", vp), ("n2_w", vp), ("n2_b", vp), ("ls1", vp), ("ls2", vp),
                ("qkv_w", vp), ("qkv_b", vp), ("proj_w", vp), ("proj_b", vp), ("fc1_w", vp), ("fc1_b", vp), ("fc2_w", vp), ("fc2_b", vp)]


class CameraHeadParams(C.Structure):
    _fields_ = [("tokens", vp), ("ld_tokens", i64), ("S", i32), ("iters", i32), ("dtype", i32), ("trunk_depth", i32), ("dim", i32), ("heads", i32),
                ("token_norm_w", vp), ("token_norm_b", vp), ("trunk_norm_w", vp), ("trunk_norm_b", vp), ("empty_pose", vp),
                ("embed_w", vp), ("embed_b", vp), ("mod_w", vp), ("mod_b", vp), ("blk", CameraBlockWeights * CAMERA_MAX_TRUNK),
                ("pb1_w", vp), ("pb1_b", vp), ("pb2_w", vp), ("pb2_b", vp), ("ws", vp), ("ws_bytes", i64), ("out", vp)]


class CameraTablesParams(C.Structure):
    _fields_ = [("extrinsics", vp), ("intrinsics", vp), ("index", vp), ("B", i32), ("S", i32), ("Sc", i32), ("H", i32), ("W", i32), ("G", i32),
                ("pose_w", vp), ("pose_b", vp), ("adapt_w", vp), ("adapt_b", vp), ("enc", vp), ("emb", vp), ("tables", vp)]


# every entry point of include/omnivggt_hip.h: name -> (restype, argtypes)
SYMBOLS = {
    "ovg_abi_version": (i32, []),
    "ovg_build_info": (C.c_char_p, []),
    "ovg_layernorm": (i32, [C.POINTER(LayerNormParams), vp]),
    "ovg_linear": (i32, [C.POINTER(LinearParams), vp]),
    "ovg_qkv": (i32, [C.POINTER(QkvParams), vp]),
    "ovg_flash_attn": (i32, [C.POINTER(AttnParams), vp]),
    "ovg_block_forward": (i32, [C.POINTER(BlockParams), vp]),
    "ovg_block_attn_prologue": (i32, [C.POINTER(BlockParams), vp]),
    "ovg_block_attn_epilogue": (i32, [C.POINTER(BlockParams), vp]),
    "ovg_im2col": (i32, [C.POINTER(Im2colParams), vp]),
    "ovg_depth_stats": (i32, [C.POINTER(DepthStatsParams), vp]),
    "ovg_dino_specials": (i32, [C.POINTER(DinoSpecialsParams), vp]),
    "ovg_assemble_tokens": (i32, [C.POINTER(AssembleParams), vp]),
    "ovg_copy_rows": (i32, [C.POINTER(CopyRowsParams), vp]),
    "ovg_head_layernorm": (i32, [C.POINTER(HeadLayerNormParams), vp]),
    "ovg_conv": (i32, [C.POINTER(ConvParams), vp]),
    "ovg_upsample": (i32, [C.POINTER(UpsampleParams), vp]),
    "ovg_dpt_out": (i32, [C.POINTER(DptOutParams), vp]),
    "ovg_dpt_tail": (i32, [C.POINTER(DptTailParams), vp]),
    "ovg_unproject": (i32, [C.POINTER(UnprojectParams), vp]),
    "ovg_heads_to_tokens": (i32, [C.POINTER(HeadsToTokensParams), vp]),
    "ovg_probe_mfma": (i32, [vp, vp, vp, i32, vp]),
    "ovg_attn_merge": (i32, [C.POINTER(AttnMergeParams), vp]),
    "ovg_attn_plan": (i32, [C.POINTER(AttnParams), C.POINTER(AttnPlanOut)]),
    "ovg_block_workspace_bytes": (i32, [C.POINTER(BlockParams), C.POINTER(BlockWorkspace)]),
    "ovg_pack_weights": (i32, [C.POINTER(PackWeightsParams), vp]),
    "ovg_camera_head": (i32, [C.POINTER(CameraHeadParams), vp]),
    "ovg_camera_head_workspace_bytes": (i64, [i32, i32]),
    "ovg_camera_tables": (i32, [C.POINTER(CameraTablesParams), vp]),
}


class OvgError(RuntimeError):
    pass


_lib = None


def load(build_if_missing=True):
    """Load the shared library. A missing or STALE library (its build stamp differs from the digest of csrc/,
    the header and the compiler flags) is rebuilt in-tree when hipcc is available, otherwise loading fails
    loudly: kernels from an older source tree must never be validated or benchmarked by accident."""
    global _lib
    if _lib is not None:
        return _lib
    from . import build as B
    if not os.path.exists(LIB_PATH) or not B.is_current():
        if not build_if_missing or not B.have_hipcc():
            raise OvgError("libomnivggt_hip.so is %s (run __graft_entry__.build()); there is no fallback path"
                           % ("missing" if not os.path.exists(LIB_PATH) else "stale: csrc/ changed since it was built"))
        B.build()      # takes an exclusive file lock and re-checks the stamp under it: N ranks starting together build once
    # torch bundles its own libamdhip64.so.7; it MUST be in the process before our library is
    # dlopen'ed, otherwise the loader maps /opt/rocm's copy for us and torch's copy for torch:
    # two HIP runtimes, and our launches on torch's streams fail (OVG_E_LAUNCH).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)      # AttributeError here == ABI mismatch, fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.ovg_abi_version() != ABI_VERSION:
        raise OvgError("ABI version mismatch: library %d, binding %d" % (lib.ovg_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise OvgError("%s failed: %s (%d)" % (what, ERRORS.get(rc, "?"), rc))


def call(name, params, stream):
    """Invoke `name(&params, stream)`; raises OvgError on a non-zero return code."""
    lib = load()
    rc = getattr(lib, name)(C.byref(params), stream)
    check(rc, name)


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


class _SplitF16:
    """compute_dtype sentinel of the split-f16 mode (C ABI: OVG_F16X2): every 16-bit tensor is a (hi, lo) pair of f16 planes, every
    contraction three f16 MFMAs with f32 accumulation -- outputs within 1e-4 of the reference like the f32 mode, at ~3x its speed."""
    _instance = None

    def __new__(cls):                         # ONE instance per process: the mode is tested with `is` (is_split, dtype_code, cache keys), so
        if cls._instance is None:             # copy.deepcopy(model), pickle / torch.save + load and multiprocessing must all hand back F32X itself
            cls._instance = super().__new__(cls)
        return cls._instance

    def __repr__(self):
        return "f32x"

    def __reduce__(self):
        return (_SplitF16, ())

    def __copy__(self):
        return self

    def __deepcopy__(self, memo):
        return self


F32X = _SplitF16()


def is_split(dtype):
    return dtype is F32X


def storage_dtype(dtype):
    """torch dtype of the planes / tensors that hold `dtype` activations and GEMM weights."""
    import torch
    return torch.float16 if dtype is F32X else dtype


def head_dtype(dtype):
    """dtype the prediction heads run in for an aggregator compute dtype (the split mode keeps the heads on the exact-f32 kernels)."""
    import torch
    return torch.float32 if dtype is F32X else dtype


def dtype_code(torch_dtype):
    import torch
    if torch_dtype is F32X:
        return OVG_F16X2
    return {torch.bfloat16: OVG_BF16, torch.float16: OVG_F16, torch.float32: OVG_F32}[torch_dtype]


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise OvgError("no HIP device visible: the aggregator hot path has no CPU fallback")
