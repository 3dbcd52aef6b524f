"""DPT dense-prediction head on the gfx950 HIP kernels.

`HipDPTHead(dpt_head_module)` runs the forward of `heads.DPTHead` (reference heads/dpt_head.py:128-304) on
the `ovg_head_layernorm` / `ovg_conv` / `ovg_upsample` / `ovg_dpt_out` entries of libomnivggt_hip.so:
NHWC 16-bit activations, every convolution an implicit GEMM on the MFMA with bias / ReLU / residual /
skip / UV-position-embedding / ConvTranspose pixel scatter folded into its epilogue. The parameters
stay in the wrapped `DPTHead` (same state-dict keys); they are re-packed (NHWC tap order, 16-bit) on
first use per device / dtype.

Used by `OmniVGGT` in all three compute dtypes (f32 parity mode: exact-f32 MFMA convolutions, r03; `OmniVGGT(hip_heads_f32=False)`
keeps the PyTorch head there).
Differences from the reference's arithmetic (all exact in real arithmetic, measured in the tests):
  * the 1x1 `out_conv` of each fusion block runs BEFORE the bilinear upsampling instead of after
    (both are linear and the interpolation weights sum to one): 4x fewer FLOPs;
  * the in-place ReLU that opens every ResidualConvUnit (dpt_head.py:379-399) is applied by the
    producer of that unit's input.
"""
import torch

from . import lib as L
from . import ops
from .heads import uv_position_embedding


def _taps_major(w):
    """Conv2d weight [co, ci, kh, kw] -> [co, kh*kw*ci] (tap-major, channels innermost: NHWC implicit GEMM)."""
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)


def _pad_rows(w, mult=128):
    rows = (w.shape[0] + mult - 1) // mult * mult
    if rows == w.shape[0]:
        return w
    out = torch.zeros(rows, w.shape[1], dtype=w.dtype, device=w.device)
    out[: w.shape[0]] = w
    return out


def uv_tables(ch, ph, pw, W, H, device):
    """Separable form of uv_position_embedding: (pos_x [pw, ch/2], pos_y [ph, ch/2]) f32."""
    emb = uv_position_embedding(torch.zeros(1, ch, ph, pw), W, H)[0]          # [ch, ph, pw] on CPU, exact
    half = ch // 2
    pos_x = emb[:half, 0, :].t().contiguous()
    pos_y = emb[half:, :, 0].t().contiguous()
    return pos_x.to(device), pos_y.to(device)


class HipDPTHead:
    def __init__(self, head):
        self.head = head
        self._packed = None
        self._key = None
        self._pos = {}
        self.fused_tail = True       # False: the three-launch output stage (ovg_upsample -> ovg_conv -> ovg_dpt_out), kept for A/B and the f32 modes

    # -- weights ---------------------------------------------------------------------------------
    def _pack(self, dtype, device):
        h, sc = self.head, self.head.scratch
        cv = lambda w: w.detach().to(device=device, dtype=dtype).contiguous()
        f32 = lambda t: None if t is None else t.detach().to(device=device, dtype=torch.float32).contiguous()
        P = {"norm_w": f32(h.norm.weight), "norm_b": f32(h.norm.bias)}
        P["proj"] = [(cv(_taps_major(m.weight)), f32(m.bias)) for m in h.projects]
        r0, r1, r3 = h.resize_layers[0], h.resize_layers[1], h.resize_layers[3]
        # ConvTranspose2d weight [ci, co, kh, kw] -> rows (dy, dx, co), columns ci
        P["up0"] = (cv(r0.weight.permute(2, 3, 1, 0).reshape(-1, r0.weight.shape[0])), f32(r0.bias))
        P["up1"] = (cv(r1.weight.permute(2, 3, 1, 0).reshape(-1, r1.weight.shape[0])), f32(r1.bias))
        P["down3"] = (cv(_taps_major(r3.weight)), f32(r3.bias))
        P["rn"] = [cv(_taps_major(getattr(sc, "layer%d_rn" % (i + 1)).weight)) for i in range(4)]

        def rcu(m):
            return (cv(_taps_major(m.conv1.weight)), f32(m.conv1.bias), cv(_taps_major(m.conv2.weight)), f32(m.conv2.bias))

        P["fusion"] = {}
        for i in (1, 2, 3, 4):
            f = getattr(sc, "refinenet%d" % i)
            P["fusion"][i] = {"out": (cv(_taps_major(f.out_conv.weight)), f32(f.out_conv.bias)),
                              "rcu1": rcu(f.resConfUnit1) if f.with_skip else None, "rcu2": rcu(f.resConfUnit2)}
        P["oc1"] = (cv(_taps_major(sc.output_conv1.weight)), f32(sc.output_conv1.bias))
        c2a, c2b = sc.output_conv2[0], sc.output_conv2[2]
        P["oc2a"] = (cv(_pad_rows(_taps_major(c2a.weight))), f32(c2a.bias))
        P["oc2b"] = (f32(c2b.weight.reshape(c2b.weight.shape[0], -1)), f32(c2b.bias))
        return P

    def _weights(self, dtype, device):
        # parameter versions: load_state_dict / .to() / in-place edits of the wrapped module invalidate the pack
        key = (dtype, str(device), tuple((q.data_ptr(), q._version) for q in self.head.parameters()))
        if self._key != key:
            self._packed, self._key = self._pack(dtype, device), key
        return self._packed

    def repack(self):
        """Call after loading new parameters into the wrapped module."""
        self._key = None

    def _postab(self, ch, ph, pw, W, H, device):
        key = (ch, ph, pw, W, H, str(device))
        if key not in self._pos:
            self._pos[key] = uv_tables(ch, ph, pw, W, H, device)
        return self._pos[key]

    # -- forward ---------------------------------------------------------------------------------
    def __call__(self, aggregated_tokens_list, images, patch_start_idx, frames_chunk_size=8, dtype=torch.bfloat16, early=None):
        """early: an EarlyLevels object of THIS head (begin_early) that was fed the intermediate layers while the aggregator was still running;
        its pyramid levels are used instead of recomputing them (same kernels on the same data: bit-identical)."""
        B, S, _, H, W = images.shape
        step = S if (not frames_chunk_size or frames_chunk_size >= S) else frames_chunk_size
        vals, confs = [], []
        for b in range(B):
            pv, pc = [], []
            for s0 in range(0, S, step):
                pre = early.levels(b) if (early is not None and step == S and early.matches(B, S, H, W, dtype)) else None
                v, c = self._chunk(aggregated_tokens_list, b, s0, min(s0 + step, S), H, W, patch_start_idx, dtype, pre=pre)
                pv.append(v)
                pc.append(c)
            vals.append(torch.cat(pv, 0) if len(pv) > 1 else pv[0])
            confs.append(torch.cat(pc, 0) if len(pc) > 1 else pc[0])
        return torch.stack(vals, 0), torch.stack(confs, 0)

    def begin_early(self, B, S, H, W, patch_start_idx, dtype):
        """Start a forward whose pyramid levels are computed AS SOON AS their aggregator layer exists (OmniVGGT.forward feeds layers 4 / 11 / 17
        from a hook on this head's side stream while the aggregator runs layers 5 .. 23): dpt_head.py:185-226 needs nothing but that layer."""
        return EarlyLevels(self, B, S, H, W, patch_start_idx, dtype)

    def _rcu(self, w, x, dtype, add2=None, relu_out=False, first=None):
        """ResidualConvUnit on an already ReLU'd input x: conv2(relu(conv1(x))) + x (+ add2) (ReLU'd if relu_out).
        first: relu(conv1(x)) if it was computed ahead (EarlyLevels)."""
        w1, b1, w2, b2 = w
        t = first if first is not None else ops.conv(x, w1, b1, dtype, 256, ksize=3, relu=True)
        return ops.conv(t, w2, b2, dtype, 256, ksize=3, add1=x, add2=add2, relu=relu_out)

    def _level(self, P, i, t, n, H, W, start, dtype):
        """Pyramid level i (dpt_head.py:185-226: norm -> projects[i] + position embedding -> resize_layers[i] -> layer{i+1}_rn, ReLU'd for its
        consumers) from that layer's tokens t [n, tokens, 2C] f32."""
        head = self.head
        ps = head.patch_size
        ph, pw = H // ps, W // ps
        dev = t.device
        oc = [m.weight.shape[0] for m in head.projects]
        tpv = t.shape[1]
        x = ops.head_layernorm(t.reshape(n * tpv, t.shape[2]), P["norm_w"], P["norm_b"], head.norm.eps, dtype, n,
                               tokens_per_view=tpv, n_special=start)
        x = x.view(n, ph, pw, -1)
        w, bias = P["proj"][i]
        x = ops.conv(x, w, bias, dtype, oc[i], ksize=1, pos=self._postab(oc[i], ph, pw, W, H, dev))
        if i == 0:
            x = ops.conv(x, P["up0"][0], P["up0"][1], dtype, oc[0], ksize=1, upshuffle=4)
        elif i == 1:
            x = ops.conv(x, P["up1"][0], P["up1"][1], dtype, oc[1], ksize=1, upshuffle=2)
        elif i == 3:
            x = ops.conv(x, P["down3"][0], P["down3"][1], dtype, oc[3], ksize=3, stride=2)
        # layerN_rn (no bias); its only consumers open with the in-place ReLU -> emit relu(x)
        return ops.conv(x, P["rn"][i], None, dtype, 256, ksize=3, relu=True)

    def _chunk(self, toks, b, s0, s1, H, W, start, dtype, pre=None):
        head = self.head
        n, ps = s1 - s0, head.patch_size
        ph, pw = H // ps, W // ps
        dev = toks[0].device
        P = self._weights(dtype, dev)
        pyramid, first = [], {}
        for i, layer in enumerate(head.intermediate_layer_idx):
            if pre is not None and i in pre:
                pyramid.append(pre[i][0])
                first[i] = pre[i][1]
                continue
            pyramid.append(self._level(P, i, toks[layer][b, s0:s1], n, H, W, start, dtype))

        F = P["fusion"]
        # refinenet4: no skip
        u = self._rcu(F[4]["rcu2"], pyramid[3], dtype)
        u = ops.conv(u, F[4]["out"][0], F[4]["out"][1], dtype, 256, ksize=1)
        y = ops.upsample(u, pyramid[2].shape[1], pyramid[2].shape[2], dtype)
        for lvl, skip, size in ((3, pyramid[2], pyramid[1].shape[1:3]), (2, pyramid[1], pyramid[0].shape[1:3]),
                                (1, pyramid[0], (2 * pyramid[0].shape[1], 2 * pyramid[0].shape[2]))):
            xs = self._rcu(F[lvl]["rcu1"], skip, dtype, add2=y, relu_out=True, first=first.get(lvl - 1))      # relu(y + RCU1(skip))
            u = self._rcu(F[lvl]["rcu2"], xs, dtype)
            u = ops.conv(u, F[lvl]["out"][0], F[lvl]["out"][1], dtype, 256, ksize=1)
            y = ops.upsample(u, size[0], size[1], dtype)
        y = ops.conv(y, P["oc1"][0], P["oc1"][1], dtype, 128, ksize=3)
        pos = self._postab(128, ph * ps, pw * ps, W, H, dev)
        if self.fused_tail and ops.dpt_tail_supported(y, dtype, ph * ps, pw * ps):
            # upsample + position embedding + output_conv2 + activation in one launch (csrc/ovg_dpt_tail.h): the image-resolution maps stay on chip
            return ops.dpt_tail(y, ph * ps, pw * ps, dtype, pos, P["oc2a"][0], P["oc2a"][1], P["oc2b"][0], P["oc2b"][1], head.activation)
        y = ops.upsample(y, ph * ps, pw * ps, dtype, pos=pos)
        hmap = ops.conv(y, P["oc2a"][0], P["oc2a"][1], dtype, 32, ksize=3, relu=True, out_f32=True)
        return ops.dpt_out(hmap, P["oc2b"][0], P["oc2b"][1], head.activation)


class EarlyLevels:
    """Pyramid levels of one HipDPTHead forward that were computed ahead of the head's own call (round 6). Level i of the DPT pyramid reads
    ONE aggregator layer (intermediate_layer_idx[i] = 4 / 11 / 17 / 23): levels 0-2 -- LayerNorm, the 1x1 projection, the resize convolution,
    layer{i+1}_rn and the first convolution of the fusion block's ResidualConvUnit on that skip -- are independent of everything the aggregator
    computes after that layer. OmniVGGT.forward feeds them from a per-layer hook on the head's side stream, so ~40 % of a head's launches run
    in the shadow of aggregator blocks 5 .. 23 (their partially filled last rounds leave CUs idle) instead of behind the last block.
    The kernels, their inputs and their order per level are those of HipDPTHead._chunk: the results are bit-identical."""

    def __init__(self, hip_head, B, S, H, W, patch_start_idx, dtype):
        self.h, self.key, self.start, self.dtype = hip_head, (B, S, H, W, dtype), patch_start_idx, dtype
        self.layers = list(hip_head.head.intermediate_layer_idx)
        self._lv = {}                               # (b, i) -> (relu(layer_rn(...)), relu(conv1 of rcu1 on it))

    def wants(self, layer_index):
        """True when aggregator layer `layer_index` feeds one of the three early pyramid levels (never the last one: its layer ends the aggregator)."""
        return layer_index in self.layers[:3] and layer_index != self.layers[3]

    def feed(self, layer_index, tokens):
        """tokens [B, S, P, 2C] f32 of aggregator layer `layer_index` (complete on the current stream)."""
        B, S, H, W, dtype = self.key
        P = self.h._weights(dtype, tokens.device)
        for i, layer in enumerate(self.layers[:3]):
            if layer != layer_index:
                continue
            for b in range(B):
                pyr = self.h._level(P, i, tokens[b], S, H, W, self.start, dtype)
                w1, b1, _, _ = P["fusion"][i + 1]["rcu1"]
                self._lv[(b, i)] = (pyr, ops.conv(pyr, w1, b1, dtype, 256, ksize=3, relu=True))

    def matches(self, B, S, H, W, dtype):
        return self.key == (B, S, H, W, dtype)

    def levels(self, b):
        return {i: v for (bb, i), v in self._lv.items() if bb == b}


class HipCameraHead:
    """Camera head on the `ovg_camera_head` entry (csrc/ovg_camhead.hip): the iterative pose regressor of
    heads/camera_head.py:84-154 with its six kinds of GEMM as split-K weight streams on the MFMA, in the compute
    dtype (bf16 / f16, or f32 on the exact-f32 MFMA in the parity mode); the parameters stay in the wrapped `heads.CameraHead` (same state-dict keys) and are re-packed
    on first use per device / dtype. One C call per batch element issues all ~165 launches of the four rounds."""

    def __init__(self, head):
        # the kernel hard-codes the reference's default head: absT_quaR_FoV activations (translation / quaternion linear,
        # field of view ReLU: heads/camera_head.py:37-41, head_act.py:12-35), dim 2048, <= 4 trunk blocks. A wrapped module
        # that says otherwise (e.g. the reference's own CameraHead built with other arguments) is refused, not mis-served.
        for attr, want in (("trans_act", "linear"), ("quat_act", "linear"), ("fl_act", "relu")):
            if hasattr(head, attr) and getattr(head, attr) != want:
                raise ValueError("HipCameraHead implements %s='%s' only (got '%s')" % (attr, want, getattr(head, attr)))
        if len(head.trunk) > L.CAMERA_MAX_TRUNK or head.token_norm.weight.numel() != 2048:
            raise ValueError("HipCameraHead needs dim 2048 and at most %d trunk blocks" % L.CAMERA_MAX_TRUNK)
        self.head = head
        self._packed = None
        self._key = None
        self._ws = {}

    def _pack(self, dtype, device):
        h = self.head
        cv = lambda w: w.detach().to(device=device, dtype=dtype).contiguous()
        f32 = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()
        mod = h.poseLN_modulation[1]
        P = {"token_norm_w": f32(h.token_norm.weight), "token_norm_b": f32(h.token_norm.bias),
             "trunk_norm_w": f32(h.trunk_norm.weight), "trunk_norm_b": f32(h.trunk_norm.bias),
             "empty_pose": f32(h.empty_pose_tokens.reshape(-1)), "embed_w": f32(h.embed_pose.weight), "embed_b": f32(h.embed_pose.bias),
             "mod_w": cv(mod.weight), "mod_b": f32(mod.bias),
             "pb1_w": cv(h.pose_branch.fc1.weight), "pb1_b": f32(h.pose_branch.fc1.bias),
             "pb2_w": f32(h.pose_branch.fc2.weight), "pb2_b": f32(h.pose_branch.fc2.bias),
             "heads": h.trunk[0].attn.heads, "blocks": []}
        for blk in h.trunk:
            P["blocks"].append({"n1_w": f32(blk.norm1.weight), "n1_b": f32(blk.norm1.bias), "n2_w": f32(blk.norm2.weight), "n2_b": f32(blk.norm2.bias),
                                "ls1": f32(blk.ls1.gamma), "ls2": f32(blk.ls2.gamma),
                                "qkv_w": cv(blk.attn.qkv.weight), "qkv_b": f32(blk.attn.qkv.bias),
                                "proj_w": cv(blk.attn.proj.weight), "proj_b": f32(blk.attn.proj.bias),
                                "fc1_w": cv(blk.mlp.fc1.weight), "fc1_b": f32(blk.mlp.fc1.bias),
                                "fc2_w": cv(blk.mlp.fc2.weight), "fc2_b": f32(blk.mlp.fc2.bias)})
        return P

    def _weights(self, dtype, device):
        key = (dtype, str(device), tuple((q.data_ptr(), q._version) for q in self.head.parameters()))
        if self._key != key:
            self._packed, self._key = self._pack(dtype, device), key
        return self._packed

    def repack(self):
        self._key = None

    def __call__(self, aggregated_tokens_list, num_iterations=4, dtype=torch.bfloat16):
        """-> list of `num_iterations` tensors (B, S, 9), like CameraHead.forward."""
        toks = aggregated_tokens_list[-1]
        B, S = toks.shape[:2]
        W = self._weights(dtype, toks.device)
        outs = []
        for b in range(B):
            cam = toks[b, :, 0]                                        # [S, 2C] f32 view, row stride = tokens_per_view * 2C
            if cam.stride(-1) != 1:
                cam = cam.contiguous()
            key = (S, dtype, str(toks.device))
            if key not in self._ws:
                self._ws.clear()
                self._ws[key] = torch.empty(ops.camera_head_workspace_bytes(S, dtype), device=toks.device, dtype=torch.uint8)
            outs.append(ops.camera_head(cam, W, dtype, iters=num_iterations, ws=self._ws[key]))
        full = torch.stack(outs, dim=1)                                # [iters, B, S, 9]
        return [full[i] for i in range(num_iterations)]
