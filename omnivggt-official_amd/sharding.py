"""View-sharded aggregator across the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference has no parallelism at all; this is the one strategy the
path admits (section 8e): everything except global attention is per-view, so rank r owns a
contiguous range of views -- i.e. a contiguous slice of the (1, S*1374, C) global sequence
(models/aggregator.py:317-318) -- and the only exchange is, per global block, an
all-gather of the post-norm/post-RoPE K and of V^T.  On the 8-GPU full mesh an all-gather
is 7 concurrent point-to-point writes of the local shard (per-link bound, ~45 MB per rank
per layer at S=64), issued right after the K/V part of the QKV GEMM and overlapped with
the Q part (async collective on RCCL's stream; the compute stream only waits before the
attention kernel).  Attention then runs over `world` K/V^T segments in rank order (the
online softmax makes the result independent of how keys are split).

Uneven S % world is handled by padding every rank's K/V^T buffer to the largest shard and
passing the per-rank valid key count; the kernel masks each segment's tail.

The numeric steps go through an *executor* (HipExecutor below); tests drive the same
control flow over gloo on CPU with an oracle-backed executor (tests/test_sharding_gloo.py).
"""
import torch
import torch.distributed as dist

from . import ops

C = 1024


def partition(n_views, world):
    """Contiguous view ranges [(lo,hi)] per rank; the first n_views % world ranks get one extra."""
    base, extra = divmod(n_views, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


class HipExecutor:
    """Numeric steps of the sharded forward on the gfx950 kernels."""

    def __init__(self, agg, device):
        self.agg, self.device = agg, device
        self.pk = agg.pack(device)

    def embed(self, inputs, view_slice):
        return self.agg.embed(self.pk, *inputs, view_slice=view_slice)

    def new_outputs(self, n_local, P):
        return [torch.empty(1, n_local, P, 2 * C, device=self.device, dtype=torch.float32) for _ in range(self.agg.depth)]

    def workspaces(self, n_local, max_local, P):
        from .aggregator import Workspace
        ws_f = self.agg.workspace(n_local * P, P, self.device)
        # global workspace: K/V^T padded to the LARGEST shard so all ranks gather equal-sized buffers
        ws_g = Workspace(n_local * P, n_local * P, self.agg.compute_dtype, self.device).share_from(ws_f)
        pad = ops.pad_to(max_local * P, ops.KV_TILE)
        ws_g.q, ws_g.k, ws_g.vt = ops.alloc_qkv(16, pad, pad, self.agg.compute_dtype, self.device)
        return ws_f, ws_g

    def gather_buffers(self, ws_g, world):
        return (torch.empty((world,) + tuple(ws_g.k.shape), device=self.device, dtype=ws_g.k.dtype),
                torch.empty((world,) + tuple(ws_g.vt.shape), device=self.device, dtype=ws_g.vt.dtype))

    def _geo(self):
        return dict(tokens_per_view=self.agg.tokens_per_view, grid_w=self.agg.grid_hw[1])

    def frame_block(self, i, ws, x_in, x_out, inject, P):
        self.pk["frame"][i].forward(ws, x_in, x_out, inject=inject, inj_period=P, **self._geo())

    def global_kv(self, i, ws, x_in, x_out):
        from . import lib as L
        p = self.pk["global"][i].params(ws, x_in, x_out, **self._geo())
        p.qkv_part = 1
        L.call("ovg_block_attn_prologue", p, torch.cuda.current_stream().cuda_stream)
        return ws.k, ws.vt

    def global_q(self, i, ws, x_in, x_out):
        from . import lib as L
        p = self.pk["global"][i].params(ws, x_in, x_out, **self._geo())
        p.qkv_part = 2
        L.call("ovg_block_attn_prologue", p, torch.cuda.current_stream().cuda_stream)

    def global_rest(self, i, ws, x_in, x_out, kg, vg, counts, rank):
        from . import lib as L
        p = self.pk["global"][i].params(ws, x_in, x_out, **self._geo())
        e = 0
        for r, nk in enumerate(counts):
            if r == rank:
                continue
            p.extra[e].k, p.extra[e].vt, p.extra[e].nk, p.extra[e].nk_pad = kg[r].data_ptr(), vg[r].data_ptr(), nk, kg.shape[2]
            e += 1
        p.nseg_extra, p.local_seg_index = e, rank
        ev = self.agg.next_attention_events()
        if ev is not None:
            p.ev_attn_start, p.ev_attn_stop = ev[0].cuda_event, ev[1].cuda_event
        L.call("ovg_block_attn_epilogue", p, torch.cuda.current_stream().cuda_stream)


class ViewSharding:
    """Attach to a ZeroAggregator (`agg.shard = ViewSharding(group)`) to run it view-sharded."""

    def __init__(self, group=None, executor_factory=None, gather_output=False):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.executor_factory = executor_factory or (lambda agg, device: HipExecutor(agg, device))
        self.gather_output = gather_output
        self.last_partition = None

    def forward(self, agg, images, extrinsics, intrinsics, depth, mask, depth_gt_index, camera_gt_index):
        B, S = images.shape[:2]
        if B != 1:
            raise NotImplementedError("view sharding needs B == 1 (views must be a contiguous slice of one sequence)")
        if S < self.world:
            raise ValueError("fewer views (%d) than ranks (%d)" % (S, self.world))
        if self.world > ops.L.OVG_MAX_SEG:
            raise ValueError("at most %d ranks per attention call" % ops.L.OVG_MAX_SEG)
        agg.set_geometry(images.shape[-2], images.shape[-1])
        P = agg.tokens_per_view
        parts = partition(S, self.world)
        self.last_partition = parts
        lo, hi = parts[self.rank]
        n_local, max_local = hi - lo, max(h - l for l, h in parts)
        counts = [(h - l) * P for l, h in parts]
        ex = self.executor_factory(agg, images.device)

        with torch.no_grad():
            tokens0, tables = ex.embed((images, extrinsics, intrinsics, depth, mask, depth_gt_index, camera_gt_index), (lo, hi))
            ws_f, ws_g = ex.workspaces(n_local, max_local, P)
            kg, vg = ex.gather_buffers(ws_g, self.world)
            outs = ex.new_outputs(n_local, P)
            x = tokens0
            for i in range(agg.depth):
                buf = outs[i].view(n_local * P, 2 * C)
                ex.frame_block(i, ws_f, x, buf[:, :C], tables[i + 1][lo:hi].contiguous(), P)
                k_loc, vt_loc = ex.global_kv(i, ws_g, buf[:, :C], buf[:, C:])
                wk = dist.all_gather_into_tensor(kg.flatten(0, 1), k_loc, group=self.group, async_op=True)
                wv = dist.all_gather_into_tensor(vg.flatten(0, 1), vt_loc, group=self.group, async_op=True)
                ex.global_q(i, ws_g, buf[:, :C], buf[:, C:])          # overlaps the all-gather
                wk.wait()
                wv.wait()
                ex.global_rest(i, ws_g, buf[:, :C], buf[:, C:], kg, vg, counts, self.rank)
                x = buf[:, C:]
            if self.gather_output:
                outs = [self.gather_views(o, parts) for o in outs]
        return outs, agg.patch_start_idx

    def gather_views(self, local, parts):
        """all-gather a (1, n_local, ...) tensor along the view axis (uneven shards padded)."""
        max_local = max(h - l for l, h in parts)
        pad = torch.zeros((1, max_local) + tuple(local.shape[2:]), device=local.device, dtype=local.dtype)
        pad[:, : local.shape[1]] = local
        full = torch.empty((self.world,) + tuple(pad.shape), device=local.device, dtype=local.dtype)
        dist.all_gather_into_tensor(full.flatten(0, 1), pad, group=self.group)
        return torch.cat([full[r][:, : h - l] for r, (l, h) in enumerate(parts)], dim=1)
