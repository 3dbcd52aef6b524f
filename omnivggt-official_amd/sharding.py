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

Head-parallel exchange (mode "heads", the default whenever the views split evenly and 16 % world == 0):
the all-gather moves (world-1) x 45 MB INTO every rank per layer (~2 ms at the ~350 GB/s an 8-GPU xGMI
all-gather sustains) against ~3.4 ms of attention, and nothing but the tiny Q projection can hide it.
Instead each rank keeps all 16 heads of ITS tokens through the QKV GEMM, then three all-to-alls hand every
rank the q, k, v^T of ALL tokens for ITS 16/world heads (a rank sends (world-1)/world of 67 MB, each peer
pair exchanges 1/world of it over its own xGMI link), attention runs over 16 (source rank, head) batch
entries with `world` K/V^T segments (`kv_heads`: entry bh reads head bh % (16/world)), and one all-to-all
returns the head-major outputs: 4x fewer bytes per rank than the all-gather and every link busy at once.
The q / k / v^T buffers are already head-major, so the send chunks are contiguous and the received K/V^T
chunks are used in place as segments; only the returned O needs one head-major -> token-major copy.

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


    # ---- head-parallel (all-to-all) global attention -----------------------------------------------------
    def heads_workspaces(self, n_local, P):
        """(frame workspace, global workspace with q/k/vt [16, pad, 64] of the LOCAL tokens, exchange buffers)."""
        ws_f = self.agg.workspace(n_local * P, P, self.device)
        ws_g = self.agg.workspace(n_local * P, n_local * P, self.device)
        ex = {"q": torch.empty_like(ws_g.q), "k": torch.empty_like(ws_g.k), "vt": torch.empty_like(ws_g.vt),
              "o": torch.empty_like(ws_g.q), "o_back": torch.empty_like(ws_g.q)}
        return ws_f, ws_g, ex

    def global_qkv(self, i, ws, x_in, x_out):
        from . import lib as L
        p = self.pk["global"][i].params(ws, x_in, x_out, **self._geo())
        p.qkv_part = 0
        L.call("ovg_block_attn_prologue", p, torch.cuda.current_stream().cuda_stream)
        return ws.q, ws.k, ws.vt

    def head_attention(self, qr, kr, vr, out, n, world):
        """qr / kr [world, 16/world, pad, 64], vr [world, 16/world, 64, pad] as received (flattened on dim 0)."""
        hpr = qr.shape[0] // world
        segs = [(kr[r * hpr:(r + 1) * hpr], vr[r * hpr:(r + 1) * hpr], n) for r in range(world)]
        ev = self.agg.next_attention_events()
        if ev is not None:
            ev[0].record()
        ops.flash_attn(qr, segs, n, self.agg.compute_dtype, out=out, variant=self.agg.attn_variant, kv_heads=hpr, head_major=True)
        if ev is not None:
            ev[1].record()
        return out

    def global_finish(self, i, ws, x_in, x_out, o_back, n):
        from . import lib as L
        ops.heads_to_tokens(o_back, n, self.agg.compute_dtype, out=ws.attn)
        p = self.pk["global"][i].params(ws, x_in, x_out, **self._geo())
        p.skip_attention = 1
        L.call("ovg_block_attn_epilogue", p, torch.cuda.current_stream().cuda_stream)


class ViewSharding:
    """Attach to a ZeroAggregator (`agg.shard = ViewSharding(group)`) to run it view-sharded.
    mode: "auto" (head-parallel all-to-all when the views split evenly and 16 % world == 0, else K/V all-gather),
    "heads" or "allgather"."""

    def __init__(self, group=None, executor_factory=None, gather_output=False, mode="auto"):
        if mode not in ("auto", "heads", "allgather"):
            raise ValueError("mode must be auto, heads or allgather")
        self.mode = mode
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.executor_factory = executor_factory or (lambda agg, device: HipExecutor(agg, device))
        self.gather_output = gather_output
        self.last_partition = None

    # ---- collectives: RCCL on device tensors; with a host backend (gloo) device tensors are staged through the
    #      host, so the same control flow also runs where RCCL cannot (tests, several ranks sharing one GPU) -------
    class _Done:
        def wait(self):
            return None

    def _staged(self, t):
        return t.is_cuda and dist.get_backend(self.group) != "nccl"

    def _all_gather(self, out, inp, async_op=False):
        if self._staged(inp):
            tmp = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(tmp, inp.cpu(), group=self.group)
            out.copy_(tmp)
            return self._Done()
        w = dist.all_gather_into_tensor(out, inp, group=self.group, async_op=async_op)
        return w if async_op else self._Done()

    def _all_to_all(self, out, inp):
        if self._staged(inp):
            tmp = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(tmp, inp.cpu(), group=self.group)
            out.copy_(tmp)
        else:
            dist.all_to_all_single(out, inp, group=self.group)

    def _all_reduce_max(self, t):
        if self._staged(t):
            tmp = t.cpu()
            dist.all_reduce(tmp, op=dist.ReduceOp.MAX, group=self.group)
            t.copy_(tmp)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)

    def choose_mode(self, run, n_views, tol=5e-2):
        """Self-check for mode "auto": `run()` must execute one sharded forward and return a tensor of it (e.g. the
        last layer). Runs the K/V all-gather form and the head-parallel all-to-all form once each, compares them, lets
        all ranks agree (MAX all-reduce of a failure flag) and pins `self.mode` to the all-to-all form only if it ran
        and matched. Returns a small report dict."""
        report = {}
        if n_views % self.world != 0 or 16 % self.world != 0:
            self.mode = "allgather"
            report["exchange"] = "K/V all-gather"
            return report
        self.mode = "allgather"
        ref = run().float().clone()
        self.mode = "heads"
        bad = torch.zeros(1, device=ref.device)
        err = float("nan")
        try:
            got = run().float()
            err = float((got - ref).abs().max() / ref.abs().max().clamp(min=1e-30))
            bad[0] = 0.0 if err < tol else 1.0
        except Exception as e:                             # only on a broken collective / kernel
            bad[0] = 1.0
            report["error"] = repr(e)[:200]
        self._all_reduce_max(bad)
        if float(bad.item()) > 0:
            self.mode = "allgather"
        report["selfcheck_max_rel_vs_allgather"] = err
        report["exchange"] = "K/V all-gather" if self.mode == "allgather" else "head-parallel all-to-all"
        return report

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
        even = S % self.world == 0 and 16 % self.world == 0
        if self.mode == "heads" and not even:
            raise ValueError("head-parallel sharding needs S % world == 0 and 16 % world == 0")
        f32_path = getattr(agg, "compute_dtype", None) == torch.float32     # heads_to_tokens is 16-bit only
        if self.mode == "heads" or (self.mode == "auto" and even and self.world > 1 and not f32_path):
            return self._forward_heads(agg, ex, images, extrinsics, intrinsics, depth, mask, depth_gt_index, camera_gt_index,
                                       parts, lo, hi, P)

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
                wk = self._all_gather(kg.flatten(0, 1), k_loc, async_op=True)
                wv = self._all_gather(vg.flatten(0, 1), vt_loc, async_op=True)
                ex.global_q(i, ws_g, buf[:, :C], buf[:, C:])          # overlaps the all-gather
                wk.wait()
                wv.wait()
                ex.global_rest(i, ws_g, buf[:, :C], buf[:, C:], kg, vg, counts, self.rank)
                x = buf[:, C:]
            if self.gather_output:
                outs = [self.gather_views(o, parts) for o in outs]
        return outs, agg.patch_start_idx

    def _forward_heads(self, agg, ex, images, extrinsics, intrinsics, depth, mask, depth_gt_index, camera_gt_index, parts, lo, hi, P):
        """Head-parallel global attention: q / k / v^T all-to-all -> attention over this rank's heads for ALL
        tokens -> all-to-all of the head-major outputs back to the token owners (module docstring)."""
        n_local = hi - lo
        n = n_local * P
        a2a = self._all_to_all
        with torch.no_grad():
            tokens0, tables = ex.embed((images, extrinsics, intrinsics, depth, mask, depth_gt_index, camera_gt_index), (lo, hi))
            ws_f, ws_g, xb = ex.heads_workspaces(n_local, P)
            outs = ex.new_outputs(n_local, P)
            x = tokens0
            for i in range(agg.depth):
                buf = outs[i].view(n, 2 * C)
                ex.frame_block(i, ws_f, x, buf[:, :C], tables[i + 1][lo:hi].contiguous(), P)
                q, k, vt = ex.global_qkv(i, ws_g, buf[:, :C], buf[:, C:])
                a2a(xb["q"], q)
                a2a(xb["k"], k)
                a2a(xb["vt"], vt)
                ex.head_attention(xb["q"], xb["k"], xb["vt"], xb["o"], n, self.world)
                a2a(xb["o_back"], xb["o"])
                ex.global_finish(i, ws_g, buf[:, :C], buf[:, C:], xb["o_back"], n)
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
        self._all_gather(full.flatten(0, 1), pad)
        return torch.cat([full[r][:, : h - l] for r, (l, h) in enumerate(parts)], dim=1)
