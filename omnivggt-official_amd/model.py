"""OmniVGGT facade with the reference's constructor / forward / state-dict contract
(omnivggt/models/omnivggt.py:10-68), the aggregator replaced by the gfx950 HIP path.

    model = OmniVGGT()                               # no torch.hub call, no network
    model.load_state_dict(load_file("OmniVGGT.safetensors"), strict=True)
    model = model.to("cuda").eval()
    out = model(images, extrinsics, intrinsics, depth, mask, depth_gt_index, camera_gt_index)

`compute_dtype` selects the aggregator arithmetic: torch.float32 (the DEFAULT, like the reference, which runs
fp32 end to end -- inference.py has no autocast, omnivggt.py:45 disables it around the heads: exact-f32 MFMA,
matches the reference CPU path to <= 1e-4 relative), torch.bfloat16 or torch.float16 (throughput modes, an explicit
opt-in: ~14x faster, tokens within the bf16-autocast twin's own error, profiles/r02_lowprec_parity.txt).  Heads: in the f32 parity
mode the two DPT heads and the camera head run on the same HIP kernels in f32 (exact-f32 MFMA implicit-GEMM convolutions and
weight streams, no rounding point below f32, like the reference which disables autocast around the heads, omnivggt.py:45;
`hip_heads_f32=False` puts all three back on the PyTorch / MIOpen modules). In the 16-bit
modes the two DPT heads run on the HIP kernels (heads_hip.py: 16-bit NHWC implicit-GEMM convolutions) and the camera head on
ovg_camera_head (16-bit weight streams; residual stream, statistics, softmax and the pose accumulation in f32);
`hip_heads=False` forces the PyTorch heads everywhere, `hip_camera_head=False` only the camera head (pose_enc then carries
no 16-bit head error: 3.6 ms instead of 1.2 ms at 8 views).
"""
import torch
import torch.nn as nn

from . import lib as L
from .aggregator import ZeroAggregator
from .heads import CameraHead, DPTHead
from .heads_hip import HipCameraHead, HipDPTHead

try:  # the reference mixes in huggingface_hub.PyTorchModelHubMixin (omnivggt.py:3,10)
    from huggingface_hub import PyTorchModelHubMixin as _HubMixin
except Exception:  # pragma: no cover - optional
    class _HubMixin:  # type: ignore
        pass


_HEAD_STREAMS = {}


class OmniVGGT(nn.Module, _HubMixin):
    def __init__(self, img_size=518, patch_size=14, embed_dim=1024, depth=24, dino_depth=24,
                 compute_dtype=torch.float32, dpt_layers=(4, 11, 17, 23), hip_heads=True, hip_camera_head=True, hip_heads_f32=True,
                 head_dtype=None, dpt_frames_chunk=64, concurrent_heads=True):
        super().__init__()
        # the three heads are independent given the aggregator's tokens: on one GPU they run on three side streams (forked from / joined to the
        # caller's stream), so the camera head's ~250 tiny launches and the level-3 / level-4 convolutions of the two DPT heads (46-184 workgroups
        # on 256 CUs) overlap instead of queueing behind each other. Results are bit-identical to the sequential order.
        self.concurrent_heads = concurrent_heads
        # round 6: the DPT pyramid levels that read aggregator layers 4 / 11 / 17 start as soon as that layer is done, on the heads' side streams,
        # in the shadow of the remaining aggregator blocks (heads_hip.EarlyLevels); results are bit-identical to the late order
        self.early_dpt_levels = True
        # frames per pass of the HIP DPT heads. The reference walks the views in chunks of 8 (dpt_head.py:133,163) to bound activation memory on
        # 24-80 GB parts; per-frame results do not depend on the chunking, the 288 GB of an MI355X hold 64 frames of head activations (~17 GB in
        # bf16), and the level-3 / level-4 convolutions (19^2 and 37^2 pixels per frame) only fill the chip from a few dozen frames on
        self.dpt_frames_chunk = dpt_frames_chunk
        # head_dtype: None = the heads follow the aggregator's compute dtype (bf16 / f16 heads in the 16-bit modes, exact-f32 heads in the
        # f32 and split-f16 modes); torch.float32 = always the exact-f32 HIP heads, i.e. the reference's own arrangement (it disables
        # autocast around the heads, omnivggt.py:45) at ~5x the heads' time -- README "16-bit heads" has the error table behind the default
        self.head_dtype = head_dtype
        self.hip_heads = hip_heads
        self.hip_camera_head = hip_camera_head
        self.hip_heads_f32 = hip_heads_f32          # f32 parity mode: all three heads on the HIP f32 kernels (False: PyTorch / MIOpen modules)
        self.aggregator = ZeroAggregator(img_size=img_size, patch_size=patch_size, embed_dim=embed_dim, depth=depth,
                                         dino_depth=dino_depth, pose_hidden_dim=9, compute_dtype=compute_dtype)
        layers = tuple(min(l, depth - 1) for l in dpt_layers)
        self.camera_head = CameraHead(dim_in=2 * embed_dim)
        self.point_head = DPTHead(dim_in=2 * embed_dim, output_dim=4, activation="inv_log", conf_activation="expp1",
                                  intermediate_layer_idx=layers)
        self.depth_head = DPTHead(dim_in=2 * embed_dim, output_dim=2, activation="exp", conf_activation="expp1",
                                  intermediate_layer_idx=layers)
        # HIP front ends of the two DPT heads; plain objects (not sub-modules): the parameters and the
        # state-dict keys stay those of point_head / depth_head
        self._hip_dpt = {"point": HipDPTHead(self.point_head), "depth": HipDPTHead(self.depth_head)}
        self._hip_cam = HipCameraHead(self.camera_head)

    @staticmethod
    def _head_streams(dev):
        key = str(dev)
        if key not in _HEAD_STREAMS:                   # created once per device, shared by all models of the process (not model state: nothing to copy / pickle)
            _HEAD_STREAMS[key] = [torch.cuda.Stream(device=dev) for _ in range(3)]
        return _HEAD_STREAMS[key]

    def _run_heads(self, jobs, concurrent):
        """Run the head closures; concurrently = each on its own side stream between a fork event and a join on the caller's stream.
        Job k runs on side stream k (camera, depth, point): the early pyramid levels of a DPT head (forward) were queued on the same stream."""
        if not concurrent or len(jobs) < 2:
            return {name: fn() for name, fn in jobs}
        cur = torch.cuda.current_stream()
        dev = cur.device
        streams = self._head_streams(dev)
        fork = torch.cuda.Event()
        fork.record(cur)
        res = {}
        for name, fn in jobs:
            st = streams[self._STREAM_OF[name]]
            st.wait_event(fork)
            with torch.cuda.stream(st):
                res[name] = fn()
        for name, _ in jobs:
            cur.wait_stream(streams[self._STREAM_OF[name]])
        for val in res.values():                       # the outputs were allocated on a side stream and live on on the caller's
            for t in val:
                if torch.is_tensor(t):
                    t.record_stream(cur)
        return res

    _STREAM_OF = {"camera": 0, "depth": 1, "point": 2}

    def _camera(self, cam_tokens):
        dt = self.head_dtype or L.head_dtype(self.aggregator.compute_dtype)      # split-f16 aggregator: the heads run on the exact-f32 kernels
        toks = cam_tokens[-1]
        lowp = dt in (torch.bfloat16, torch.float16)
        if self.hip_heads and self.hip_camera_head and (lowp or self.hip_heads_f32) and toks.is_cuda and toks.shape[1] <= 4096:
            return self._hip_cam(cam_tokens, dtype=dt)
        return self.camera_head(cam_tokens)

    # activation bytes of ONE frame inside one HIP DPT head at 518 x 518 in a 16-bit dtype (64 frames measure ~17 GB); f32 heads hold twice that
    _DPT_BYTES_PER_FRAME_16 = 0.28e9

    def _frames_per_pass(self, imgs32, dt, concurrent):
        """`dpt_frames_chunk` is an upper bound sized for the 288 GB of an MI355X; the pass actually taken also fits HALF of the memory that is
        free right now (both DPT heads at once when they run concurrently; blocks freed on a side stream stay in that stream's allocator pool), so a
        smaller part, or a process that shares its GPU, degrades to the reference's 8 frames per pass (dpt_head.py:133,163) instead of failing."""
        cap = int(self.dpt_frames_chunk)
        if not imgs32.is_cuda or cap <= 8:
            return cap
        free, _ = torch.cuda.mem_get_info(imgs32.device)
        px = imgs32.shape[-1] * imgs32.shape[-2] / float(518 * 518)
        per_frame = self._DPT_BYTES_PER_FRAME_16 * px * (1 if dt in (torch.bfloat16, torch.float16) else 2) * (2 if concurrent else 1)
        fit = int(0.5 * free / per_frame)
        while cap > 8 and cap > fit:
            cap //= 2
        return max(cap, 8)

    def _dpt(self, which, head, tokens, imgs32, patch_start_idx, concurrent=False, early=None):
        dt = self.head_dtype or L.head_dtype(self.aggregator.compute_dtype)
        if self.hip_heads and imgs32.is_cuda and (dt in (torch.bfloat16, torch.float16) or self.hip_heads_f32):
            return self._hip_dpt[which](tokens, imgs32, patch_start_idx, frames_chunk_size=self._frames_per_pass(imgs32, dt, concurrent), dtype=dt,
                                        early=early)   # f32: exact-f32 MFMA convolutions (r03)
        return head(tokens, images=imgs32, patch_start_idx=patch_start_idx)

    def _begin_early_levels(self, images):
        """Round 6: while the aggregator still runs blocks 5 .. 23, the pyramid levels of the two DPT heads that read layers 4 / 11 / 17 are
        computed on the heads' side streams (heads_hip.EarlyLevels) -- a per-layer hook of the aggregator waits for the layer on the side
        stream and queues the level there. Only where the heads run concurrently on the HIP kernels and one pass covers all frames.
        Returns ({head name: EarlyLevels}, hook) or ({}, None)."""
        agg = self.aggregator
        dt = self.head_dtype or L.head_dtype(agg.compute_dtype)
        B, S, _, H, W = images.shape
        ok = (self.early_dpt_levels and self.concurrent_heads and self.hip_heads and images.is_cuda and agg.shard is None
              and (dt in (torch.bfloat16, torch.float16) or self.hip_heads_f32) and H % 14 == 0 and W % 14 == 0)
        if not ok or self._frames_per_pass(images, dt, True) < S:
            return {}, None
        early = {}
        for name, head in (("depth", self.depth_head), ("point", self.point_head)):
            if head is not None:
                early[name] = self._hip_dpt[name].begin_early(B, S, H, W, agg.patch_start_idx, dt)
        if not early or not any(e.wants(i) for e in early.values() for i in range(agg.depth)):
            return {}, None
        streams = self._head_streams(images.device)

        def hook(layer_index, layer_tokens):
            takers = [(n, e) for n, e in early.items() if e.wants(layer_index)]
            if not takers:
                return
            cur = torch.cuda.current_stream()
            ready = torch.cuda.Event()
            ready.record(cur)
            for name, e in takers:
                st = streams[self._STREAM_OF[name]]
                st.wait_event(ready)
                with torch.cuda.stream(st), torch.no_grad(), torch.amp.autocast("cuda", enabled=False):
                    e.feed(layer_index, layer_tokens)
        return early, hook

    def set_compute_dtype(self, dtype):
        self.aggregator.set_compute_dtype(dtype)
        return self

    @classmethod
    def from_safetensors(cls, path, device="cuda", **kwargs):
        """Construct without running any initialiser (meta device -> to_empty) and load a checkpoint with the
        reference's key set, e.g. checkpoints/OmniVGGT.safetensors (inference.py:321-325; SURVEY 8(f) N4: the
        reference spends ~25 s in trunc_normal_ init plus a torch.hub call before it loads the same file)."""
        from safetensors.torch import load_file
        with torch.device("meta"):
            model = cls(**kwargs)
        model = model.to_empty(device="cpu")
        model.load_state_dict(load_file(path), strict=True)
        return model.to(device).eval()

    # ---- persisted pre-packed weights ---------------------------------------------------------------
    def save_packed(self, path, device="cuda"):
        """Write this model's weights in packed form: aggregator GEMM weights in the current 16-bit compute dtype (half the
        bytes of the f32 checkpoint), conv patch weights as padded GEMM rows, every other tensor (norms, biases, tokens,
        heads) f32 -- one safetensors file that from_packed() maps back without any conversion."""
        from safetensors.torch import save_file
        dt = self.aggregator.compute_dtype
        if dt is torch.float32 or L.is_split(dt):
            raise ValueError("packed files are for the 16-bit modes; the f32 and split-f16 modes load the original checkpoint")
        tensors = {"aggregator." + k: v.detach().cpu().contiguous() for k, v in self.aggregator.export_packed(torch.device(device)).items()}
        for k, v in self.state_dict().items():
            if not k.startswith("aggregator."):
                tensors[k] = v.detach().cpu().contiguous()
        save_file(tensors, path, metadata={"format": "omnivggt_official_amd.packed", "version": "1", "compute_dtype": str(dt).replace("torch.", ""),
                                           "depth": str(self.aggregator.depth), "dino_depth": str(self.aggregator.dino_depth)})
        return path

    @classmethod
    def from_packed(cls, path, device="cuda", **kwargs):
        """Start-up path of a serving process: meta-device construction (no initialiser, no torch.hub), the heads materialised
        from the file, the aggregator's packed weights installed directly (no f32 masters of the GEMM weights, no
        pack_weights pass). The aggregator's nn.Parameters stay on the meta device: use the original checkpoint +
        load_state_dict when a state_dict() of real tensors is needed."""
        from safetensors import safe_open
        from safetensors.torch import load_file
        with safe_open(path, framework="pt") as f:
            meta = f.metadata() or {}
        if meta.get("format") != "omnivggt_official_amd.packed":
            raise ValueError("%s is not a packed OmniVGGT file (use from_safetensors for reference checkpoints)" % path)
        dt = getattr(torch, meta["compute_dtype"])
        with torch.device("meta"):
            model = cls(compute_dtype=dt, depth=int(meta["depth"]), dino_depth=int(meta["dino_depth"]), **kwargs)
        tensors = load_file(path, device=str(device))
        for name in ("camera_head", "point_head", "depth_head"):
            head = getattr(model, name)
            head.to_empty(device=device)
            head.load_state_dict({k[len(name) + 1:]: v for k, v in tensors.items() if k.startswith(name + ".")}, strict=True)
        model.aggregator.load_packed({k[len("aggregator."):]: v for k, v in tensors.items() if k.startswith("aggregator.")}, torch.device(device))
        return model.eval()

    def forward(self, images, extrinsics=None, intrinsics=None, depth=None, mask=None, depth_gt_index=None,
                camera_gt_index=None):
        if images.dim() == 4:
            images = images.unsqueeze(0)
        # the reference dereferences these unconditionally (omnivggt_aggregator.py:158-185); accept the
        # loader's "absent modality" convention (zero tensors + empty lists, visual_util.py:793-824)
        depth_gt_index = list(depth_gt_index) if depth_gt_index is not None else []
        camera_gt_index = list(camera_gt_index) if camera_gt_index is not None else []
        if depth_gt_index and (depth is None or mask is None):
            raise ValueError("depth_gt_index given without depth/mask tensors")
        if camera_gt_index and (extrinsics is None or intrinsics is None):
            raise ValueError("camera_gt_index given without extrinsics/intrinsics tensors")

        early, hook = self._begin_early_levels(images)
        self.aggregator.layer_hook = hook
        try:
            tokens, patch_start_idx = self.aggregator(images=images, extrinsics=extrinsics, intrinsics=intrinsics, depth=depth,
                                                      mask=mask, depth_gt_index=depth_gt_index, camera_gt_index=camera_gt_index)
        finally:
            self.aggregator.layer_hook = None
        out = {}
        shard = self.aggregator.shard
        sharded = shard is not None and not shard.gather_output
        with torch.no_grad(), torch.amp.autocast("cuda", enabled=False):
            imgs32 = images.float()
            cam_tokens = tokens
            if sharded:
                # tokens are this rank's views only: the camera head attends across ALL views, so
                # gather the (S,2C) camera tokens of the last layer; DPT heads are per-frame.
                parts = shard.last_partition
                lo, hi = parts[shard.rank]
                cam_tokens = [shard.gather_views(tokens[-1][:, :, :1].contiguous(), parts)]
                imgs32 = imgs32[:, lo:hi]
            jobs = []
            conc = bool(self.concurrent_heads and self.hip_heads and not sharded and imgs32.is_cuda)
            if self.camera_head is not None:
                jobs.append(("camera", lambda: self._camera(cam_tokens)))
            if self.depth_head is not None:
                jobs.append(("depth", lambda: self._dpt("depth", self.depth_head, tokens, imgs32, patch_start_idx, conc, early.get("depth"))))
            if self.point_head is not None:
                jobs.append(("point", lambda: self._dpt("point", self.point_head, tokens, imgs32, patch_start_idx, conc, early.get("point"))))
            # (queueing the two DPT heads in front of the camera head's ~165 tiny launches measured no difference: the host is far ahead of the device)
            res = self._run_heads(jobs, concurrent=conc)
            if "camera" in res:
                out["pose_enc"], out["pose_enc_list"] = res["camera"][-1], res["camera"]
            if "depth" in res:
                out["depth"], out["depth_conf"] = res["depth"]
            if "point" in res:
                out["world_points"], out["world_points_conf"] = res["point"]
            if sharded:
                for key in ("depth", "depth_conf", "world_points", "world_points_conf"):
                    if key in out:
                        out[key] = shard.gather_views(out[key].contiguous(), parts)
        out["images"] = images
        return out
