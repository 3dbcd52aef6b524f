"""View-sharded aggregator across the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference has no parallelism at all; this is the one strategy the
path admits (section 8e): everything except global attention is per-view, so rank r owns a
contiguous range of views -- i.e. a contiguous slice of the (1, S*1374, C) global sequence
(models/aggregator.py:317-318) -- and the only data-path exchange sits inside the global blocks.
Two exchange forms, both overlapped with compute (ViewSharding(mode=...)):

"allgather" -- the north star's collective: all-gather of the post-norm / post-RoPE K and of V^T.
    LN1 + K/V part of the QKV GEMM -> async all-gather on RCCL's stream -> Q part of the QKV GEMM
    -> attention launch A over the LOCAL keys only (needs nothing from the wire, writes O_a and the
    per-row log-sum-exp) -> wait for the gather -> attention launch B over the world-1 REMOTE
    segments (O_b, lse_b) -> ovg_attn_merge (exact: softmax over disjoint key sets combines through
    the two log-sum-exps) -> proj + MLP. The gather hides behind the Q projection and launch A
    (1/world of the attention work). Uneven S % world: every rank's K/V^T buffer is padded to the
    largest shard, per-segment valid key counts mask the tails. Works for every dtype incl. f32.

"heads" -- head-parallel all-to-all (the default whenever S % world == 0, 16 % world == 0, 16-bit):
    the all-gather moves (world-1) x 45 MB INTO every rank per layer at S=64; the all-to-all form
    moves 4x less: each rank runs the fused QKV GEMM for all 16 heads of ITS tokens, the q / k / v^T
    of ALL tokens for ITS 16/world heads arrive by all-to-all (the buffers are head-major, so every
    send chunk is contiguous and the received K / V^T chunks are used in place as `world` kernel
    segments), attention runs over (source rank, head) batch entries (`kv_heads`), one all-to-all
    returns the head-major outputs, ovg_heads_to_tokens restores the token-major layout.
    q, k and v^T of a head group travel as ONE grouped RCCL launch (batch_isend_irecv: 3 x (world - 1) sends + receives in a single
    ncclGroup) and the attention launch plans are told how many CUs RCCL's channels leave them (available_cus -> ovg_attn_params.cus).
    Pipelined in HEAD GROUPS where that pays (head_groups(): 2 groups while each group's attention
    launch still covers >= 2 rounds of the chip's 512 workgroup slots -- 2 or 4 ranks at 64 views;
    at 8 ranks x 8 views a rank owns 2 heads, a per-head launch would be 344 workgroups and the
    quantisation loss would exceed the exchange it hides, so it stays one launch): all inbound
    exchanges are issued up front on RCCL's stream; the compute stream waits for group 0 only, runs
    attention(0) while group 1 is still arriving, returns O(0) under attention(1), ... -- about half
    of the exchange time is hidden, the exposed part is one group's inbound + the last group's return.
    Every per-rank attention launch is also cut along the keys where the library's plan says so
    (split-KV, ovg_attn_plan): 688 workgroups on 512 slots become 3440 fifths.

The mode is a pure function of (S, world, dtype, requested mode), identical on every rank, so the
ranks agree without talking; an impossible request raises BEFORE any collective is issued.

The numeric steps go through an *executor* (HipExecutor below); tests drive the same control flow
over gloo on CPU with an oracle-backed executor (tests/test_sharding_gloo.py).
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


def resolve_mode(requested, n_views, world, is_f32):
    """Exchange form for this call; deterministic in its arguments (all ranks compute the same answer).
    Raises ValueError for an explicit request the shapes / dtype cannot run. is_f32: the compute mode is one of the <= 1e-4 modes
    (f32 or split-f16) -- those shard through the K / V^T all-gather form only."""
    eligible = n_views % world == 0 and 16 % world == 0 and not is_f32
    if requested == "heads":
        if not eligible:
            raise ValueError("head-parallel sharding needs S %% world == 0, 16 %% world == 0 and a 16-bit compute dtype "
                             "(S=%d, world=%d, f32=%s)" % (n_views, world, is_f32))
        return "heads"
    if requested == "allgather":
        return "allgather"
    return "heads" if (eligible and world > 1) else "allgather"


def _parity_mode(agg):
    """True for the <= 1e-4 compute modes (f32, split-f16): they shard through the K / V^T all-gather form only."""
    dt = getattr(agg, "compute_dtype", None)
    return dt is torch.float32 or ops.L.is_split(dt)


DEVICE_CUS = 256          # MI355X; head_groups / reserved_cus take the real count where a device is visible


def rccl_channels():
    """Channels (= workgroups, one CU each while the kernel runs) an RCCL exchange may occupy: NCCL_MAX_NCHANNELS when the job sets
    it, else RCCL's own ceiling of 32 on this class of GPU. The exchange of the sharded forward is in flight DURING the attention
    launches it overlaps, so their launch plans must not count on those CUs."""
    import os
    v = os.environ.get("NCCL_MAX_NCHANNELS", "")
    return max(1, int(v)) if v.isdigit() else 32


def available_cus(world, device_cus=DEVICE_CUS):
    """CUs the attention launch plans count on (ovg_attn_params.cus): all of them on one GPU, device minus RCCL's channels in a
    sharded run (never less than three quarters of the chip: a mis-set environment must not cripple the plan)."""
    if world <= 1:
        return device_cus
    return max(device_cus - rccl_channels(), (3 * device_cus) // 4)


def head_groups(heads_per_rank, world=1, n_tokens=None, cus=DEVICE_CUS):
    """Pipeline groups of the heads form: [(first head, count)] inside a rank's head range. Two groups (exchange of
    group 1 under attention of group 0) only while EACH group's attention launch still fills the chip: its
    world * heads * ceil(n / 256) workgroups must cover >= 2 rounds of the 2 x `cus` resident slots -- at 8 ranks x 8 views a
    rank owns 2 heads and one launch per head would be 344 workgroups (0.67 of a round): the quantisation loss would
    dwarf the ~0.15 ms of exchange it hides, so that case stays one group (split-KV keeps its single launch even)."""
    if heads_per_rank < 2:
        return [(0, heads_per_rank)]
    half = heads_per_rank // 2
    if n_tokens is not None and world * half * ((n_tokens + 255) // 256) < 4 * cus:
        return [(0, heads_per_rank)]
    return [(0, half), (half, heads_per_rank - half)]


class HipExecutor:
    """Numeric steps of the sharded forward on the gfx950 kernels. Scratch and exchange buffers are cached per shape
    (nothing is allocated inside the layer loop or per forward)."""

    def __init__(self, agg, device):
        self.agg, self.device = agg, device
        self.pk = agg.pack(device)
        self._cache = {}
        self.cus = 0        # ovg_attn_params.cus of the global-attention launches (0 = whole device); ViewSharding.executor sets it for RCCL runs

    def _stream(self):
        return torch.cuda.current_stream().cuda_stream

    def embed(self, inputs, view_slice):
        return self.agg.embed(self.pk, *inputs, view_slice=view_slice)

    def new_outputs(self, n_local, P):
        return [torch.empty(1, n_local, P, 2 * C, device=self.device, dtype=torch.float32) for _ in range(self.agg.depth)]

    def _geo(self):
        return dict(tokens_per_view=self.agg.tokens_per_view, grid_w=self.agg.grid_hw[1])

    def _cached(self, key, make):
        key = key + (self.agg.compute_dtype,)
        if key not in self._cache:
            if len(self._cache) >= 12:
                self._cache.pop(next(iter(self._cache)))
            self._cache[key] = make()
        return self._cache[key]

    # ---- K / V^T all-gather form ---------------------------------------------------------------------------
    def workspaces(self, n_local, max_local, P):
        """(frame workspace, global workspace): the global one shares the LN / attention / hidden scratch and pads
        q / k / v^T to the LARGEST shard so all ranks gather equal-sized buffers; plus o_b / lse for the two-launch merge."""
        from .aggregator import Workspace

        def make():
            dt = self.agg.compute_dtype
            ws_f = self.agg.workspace(n_local * P, P, self.device)
            pad = ops.pad_to(max_local * P, ops.KV_TILE)
            ws_g = Workspace(n_local * P, n_local * P, dt, self.device, share=ws_f, kv_rows=pad)
            ws_g.o_b = ops.empty_like_dtype((n_local * P, C), dt, self.device)      # split-f16 mode: an ops.HiLo pair like ws.attn
            ws_g.lse_a = torch.empty(16, pad, device=self.device, dtype=torch.float32)
            ws_g.lse_b = torch.empty(16, pad, device=self.device, dtype=torch.float32)
            return ws_f, ws_g
        return self._cached(("ag", n_local, max_local, P), make)

    @staticmethod
    def _wire(t):
        """The tensor that goes on the wire for a K / V^T buffer: the buffer itself, or both planes of a split-f16 pair [2, ...]."""
        return t.planes if isinstance(t, ops.HiLo) else t

    @staticmethod
    def _unwire(t, like):
        return ops.HiLo(t) if isinstance(like, ops.HiLo) else t

    def gather_buffers(self, ws_g, world):
        k, vt = self._wire(ws_g.k), self._wire(ws_g.vt)
        return self._cached(("agbuf", tuple(k.shape), world), lambda: (
            torch.empty((world,) + tuple(k.shape), device=self.device, dtype=k.dtype),
            torch.empty((world,) + tuple(vt.shape), device=self.device, dtype=vt.dtype)))

    def frame_block(self, i, ws, x_in, x_out, inject, P):
        self.pk["frame"][i].forward(ws, x_in, x_out, inject=inject, inj_period=P, **self._geo())

    def _prologue(self, i, ws, x_in, x_out, part):
        from . import lib as L
        p = self.pk["global"][i].params(ws, x_in, x_out, **self._geo())
        p.qkv_part = part
        L.call("ovg_block_attn_prologue", p, self._stream())

    def global_kv(self, i, ws, x_in, x_out):
        self._prologue(i, ws, x_in, x_out, 1)
        return self._wire(ws.k), self._wire(ws.vt)

    def global_q(self, i, ws, x_in, x_out):
        self._prologue(i, ws, x_in, x_out, 2)

    def _attention(self, q, segs, n, out, lse=None, kv_heads=0, head_major=False):
        """One flash-attention launch, timed with a HIP event pair when bench.py asked for it."""
        nk = sum(s[2] for s in segs)
        dt, variant = self.agg.compute_dtype, self.agg.attn_variant
        if ops.L.is_split(dt) and variant == 0 and getattr(self.agg, "f32x_fast_pv", False):
            variant = ops.L.ATTN_F32X_FAST_PV
        splits = getattr(self.agg, "attn_kv_splits", 0)
        cus = self.cus                                               # CUs left to the attention plans beside RCCL's channels (ViewSharding sets it)
        split_ws = None
        if splits != 1 and dt in (torch.bfloat16, torch.float16):     # per-rank launches are the ones that quantise badly (688 workgroups on 512 slots)
            key = ("split", q.shape[0], n, q.shape[1], tuple(s[2] for s in segs), variant, splits, cus)
            split_ws = self._cached(key, lambda: ops.alloc_split_ws(
                ops.attn_plan(q.shape[0], n, [s[2] for s in segs], dt, variant, splits, nq_pad=q.shape[1], cus=cus), self.device))
        ev = self.agg.next_attention_events(4.0 * q.shape[0] * n * nk * 64)
        if ev is not None:
            ev[0].record()
        ops.flash_attn(q, segs, n, dt, out=out, variant=variant, kv_heads=kv_heads, head_major=head_major, lse=lse,
                       kv_splits=splits, split_ws=split_ws, fallback_count=getattr(self.agg, "fallback_counter", None), cus=cus)
        if ev is not None:
            ev[1].record()
        return out

    def attend_local(self, i, ws, n, want_lse):
        """Launch A: this rank's queries against its OWN keys (nothing from the wire) -> ws.attn (+ lse_a)."""
        self._attention(ws.q, [(ws.k, ws.vt, n)], n, ws.attn, lse=ws.lse_a if want_lse else None)

    def attend_remote(self, i, ws, kg, vg, counts, rank, n):
        """Launch B: the same queries against the gathered segments of the other ranks -> ws.o_b, lse_b."""
        segs = [(self._unwire(kg[r], ws.k), self._unwire(vg[r], ws.vt), c) for r, c in enumerate(counts) if r != rank]
        self._attention(ws.q, segs, n, ws.o_b, lse=ws.lse_b)

    def merge_finish(self, i, ws, x_in, x_out, n, merged):
        """Combine launches A and B (if there was a B) into ws.attn, then proj + MLP (epilogue without attention)."""
        from . import lib as L
        if merged:
            ops.attn_merge(ws.attn, ws.lse_a, ws.o_b, ws.lse_b, self.agg.compute_dtype, out=ws.attn)
        p = self.pk["global"][i].params(ws, x_in, x_out, **self._geo())
        p.skip_attention = 1
        L.call("ovg_block_attn_epilogue", p, self._stream())

    # ---- head-parallel (all-to-all) form -------------------------------------------------------------------
    def heads_workspaces(self, n_local, P, world):
        """(frame workspace, global workspace with q/k/vt [16, pad, 64] of the LOCAL tokens, exchange buffers).
        Exchange buffers per head group g of `gs` heads: inbound q / k [world, gs, pad, 64], vt [world, gs, 64, pad]
        (chunk s = from source rank s), the attention output o (same shape as q) and the returned o_back [16, pad, 64]
        in global head order."""
        def make():
            ws_f = self.agg.workspace(n_local * P, P, self.device)
            ws_g = self.agg.workspace(n_local * P, n_local * P, self.device)
            pad = ws_g.q.shape[1]
            dt, dev = ws_g.q.dtype, self.device
            groups = []
            for h0, gs in head_groups(16 // world, world, n_local * P, cus=self.cus or DEVICE_CUS):
                groups.append({"h0": h0, "gs": gs,
                               "q": torch.zeros(world, gs, pad, 64, device=dev, dtype=dt), "k": torch.zeros(world, gs, pad, 64, device=dev, dtype=dt),
                               "vt": torch.zeros(world, gs, 64, pad, device=dev, dtype=dt), "o": torch.zeros(world, gs, pad, 64, device=dev, dtype=dt)})
            return ws_f, ws_g, {"groups": groups, "o_back": torch.zeros_like(ws_g.q)}
        return self._cached(("heads", n_local, P, world, self.cus), make)

    def global_qkv(self, i, ws, x_in, x_out):
        self._prologue(i, ws, x_in, x_out, 0)
        return ws.q, ws.k, ws.vt

    def head_attention(self, qr, kr, vr, out, n, world):
        """qr / kr [world, gs, pad, 64], vr [world, gs, 64, pad] as received; batch entry (s, g) attends to head g of
        every source's K / V^T chunk (kv_heads = gs, `world` segments); out like qr (head-major)."""
        gs = qr.shape[1]
        segs = [(kr[r], vr[r], n) for r in range(world)]
        return self._attention(qr.flatten(0, 1), segs, n, out.flatten(0, 1), kv_heads=gs, head_major=True)

    def global_finish(self, i, ws, x_in, x_out, o_back, n):
        from . import lib as L
        ops.heads_to_tokens(o_back, n, self.agg.compute_dtype, out=ws.attn)
        p = self.pk["global"][i].params(ws, x_in, x_out, **self._geo())
        p.skip_attention = 1
        L.call("ovg_block_attn_epilogue", p, self._stream())


class ViewSharding:
    """Attach to a ZeroAggregator (`agg.shard = ViewSharding(group)`) to run it view-sharded.
    mode: "auto" (head-parallel all-to-all when the views split evenly, 16 % world == 0 and the dtype is 16-bit, else
    K/V all-gather), "heads" or "allgather".  skip_comm (bench.py only): issue no collective at all -- the step then
    runs the same kernels on whatever the exchange buffers hold, which times the compute of a sharded step alone."""

    def __init__(self, group=None, executor_factory=None, gather_output=False, mode="auto", reserve_cus="auto"):
        """reserve_cus: plan the global-attention launches for the CUs RCCL's channels leave free (available_cus) -- "auto": when the
        backend is RCCL (its kernels share the GPU with the launches they overlap), True / False: always / never (tests)."""
        if mode not in ("auto", "heads", "allgather"):
            raise ValueError("mode must be auto, heads or allgather")
        self.mode = mode
        self.reserve_cus = reserve_cus
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.executor_factory = executor_factory or (lambda agg, device: HipExecutor(agg, device))
        self.gather_output = gather_output
        self.last_partition = None
        self.last_mode = None
        self.skip_comm = False
        self._executors = {}

    def executor(self, agg, device):
        key = (id(agg), str(device), getattr(agg, "compute_dtype", None), id(getattr(agg, "_packed", None)))
        if key not in self._executors:
            self._executors.clear()                      # a re-pack / dtype change invalidates the old executor's buffers
            ex = self.executor_factory(agg, device)
            reserve = self._nccl() if self.reserve_cus == "auto" else bool(self.reserve_cus)
            if reserve and self.world > 1 and hasattr(ex, "cus"):
                dev_cus = (torch.cuda.get_device_properties(device).multi_processor_count
                           if (torch.cuda.is_available() and str(device).startswith("cuda")) else DEVICE_CUS)
                # the budget feeds head_groups(), i.e. HOW MANY grouped exchanges a rank issues per layer: ranks that read different
                # environments (NCCL_MAX_NCHANNELS) or CU counts would issue different numbers of collectives and hang (round-5 advisor).
                # Agree once, here -- every rank creates its executor at the same point of its first forward: the MINIMUM over the ranks.
                ex.cus = self._agree_min(available_cus(self.world, dev_cus), device)
            self._executors[key] = ex
        return self._executors[key]

    def comm_report(self, device_cus=DEVICE_CUS):
        """What bench.py prints under `comm`: backend, RCCL's channel ceiling and the CU budget the attention plans were given."""
        import os
        ex = next(iter(self._executors.values()), None)
        return {"backend": dist.get_backend(self.group), "rccl_channels": rccl_channels(), "NCCL_MAX_NCHANNELS": os.environ.get("NCCL_MAX_NCHANNELS"),
                "attention_plan_cus": int(getattr(ex, "cus", 0)) or device_cus, "device_cus": device_cus}

    # ---- collectives: RCCL on device tensors; with a host backend (gloo) device tensors are staged through the
    #      host, so the same control flow also runs where RCCL cannot (tests, several ranks sharing one GPU) -------
    class _Done:
        def wait(self):
            return None

    def _nccl(self):
        return dist.get_backend(self.group) == "nccl"

    def _all_gather(self, out, inp, async_op=False):
        if self.skip_comm:
            return self._Done()
        if not self._nccl():
            tmp = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(tmp, inp.cpu().contiguous(), group=self.group)
            out.copy_(tmp)
            return self._Done()
        w = dist.all_gather_into_tensor(out, inp, group=self.group, async_op=async_op)
        return w if async_op else self._Done()

    def _all_to_all_chunks(self, outs, ins, async_op=False):
        """List-form all-to-all: ins[r] (contiguous) goes to rank r, outs[s] receives rank s's chunk for this rank.
        RCCL: one grouped send/recv (no staging copy); host backends: stacked all_to_all_single through the host."""
        if self.skip_comm:
            return self._Done()
        if not self._nccl():
            send = torch.stack([t.cpu() for t in ins]).contiguous()
            recv = torch.empty_like(send)
            dist.all_to_all_single(recv, send, group=self.group)
            for r, o in enumerate(outs):
                o.copy_(recv[r])
            return self._Done()
        w = dist.all_to_all(outs, ins, group=self.group, async_op=async_op)
        return w if async_op else self._Done()

    def _exchange_many(self, pairs, async_op=False):
        """Several list-form all-to-alls as ONE grouped RCCL launch: pairs = [(outs, ins), ...] with outs[s] / ins[r] as in
        _all_to_all_chunks. The heads form moves q, k and v^T of a head group this way -- one batch of point-to-point sends / receives
        inside a single ncclGroupStart / End (torch.distributed.batch_isend_irecv) instead of three grouped launches per group and layer
        (round-4 review: 96-192 exchange launches per forward). The chunk a rank keeps for itself is a device copy on the compute stream.
        Host backends run the pairs one after another through _all_to_all_chunks."""
        if self.skip_comm:
            return self._Done()
        if not self._nccl() and any(t.is_cuda for outs, ins in pairs for t in ins):   # host backend + device tensors: staged through the host
            for outs, ins in pairs:
                self._all_to_all_chunks(outs, ins)
            return self._Done()
        p2p = []                                         # RCCL, or a host backend on host tensors (the gloo tests run this very code)
        for outs, ins in pairs:
            for r in range(self.world):
                if r == self.rank:
                    outs[r].copy_(ins[r])
                    continue
                peer = dist.get_global_rank(self.group, r) if self.group is not None else r
                p2p.append(dist.P2POp(dist.isend, ins[r], peer, self.group))
                p2p.append(dist.P2POp(dist.irecv, outs[r], peer, self.group))
        works = dist.batch_isend_irecv(p2p) if p2p else []

        class _Many:
            def wait(self_inner):
                for w in works:
                    w.wait()
        done = _Many()
        if not async_op:
            done.wait()
            return self._Done()
        return done

    def _agree_min(self, value, device):
        """One integer, the minimum over the ranks of the group (a collective: call it at the same point on every rank)."""
        on_dev = self._nccl() and torch.cuda.is_available() and str(device).startswith("cuda")
        t = torch.tensor([int(value)], dtype=torch.int64, device=device if on_dev else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return int(t.item())

    def attention_cus_probe(self, agg, S, device, cus_values, reps=3):
        """bench.py pre-flight at N > 1 (round-5 review item 7): time THIS rank's global-attention launch of the current exchange form
        with its launch plan sized for each CU budget in `cus_values` -- alone, and with one layer's inbound exchange in flight beside
        it -- so that the first multi-GPU log shows whether reserving CUs for RCCL's channels (available_cus) helps, hurts or does nothing.
        Returns {cus: {"alone_ms", "under_exchange_ms"}} (MAX over ranks), or None for executors without a CU budget (CPU tests).
        Every rank issues the same collectives in the same order; nothing here changes the executor's state afterwards."""
        ex = self.executor(agg, device)
        if not hasattr(ex, "cus") or not (torch.cuda.is_available() and str(device).startswith("cuda")):
            return None
        mode = resolve_mode(self.last_mode or self.mode, S, self.world, _parity_mode(agg))
        P = agg.tokens_per_view
        parts = partition(S, self.world)
        lo, hi = parts[self.rank]
        n = (hi - lo) * P
        W, hpr = self.world, 16 // self.world if 16 % self.world == 0 else 0
        if mode == "heads":
            _, ws_g, xb = ex.heads_workspaces(hi - lo, P, W)
            g = xb["groups"][0]
            sl = [slice(r * hpr + g["h0"], r * hpr + g["h0"] + g["gs"]) for r in range(W)]
            launch = lambda: ex.head_attention(g["q"], g["k"], g["vt"], g["o"], n, W)
            exchange = lambda: [self._exchange_many([(list(g["q"].unbind(0)), [ws_g.q[s_] for s_ in sl]), (list(g["k"].unbind(0)), [ws_g.k[s_] for s_ in sl]),
                                                     (list(g["vt"].unbind(0)), [ws_g.vt[s_] for s_ in sl])], async_op=True)]
        else:
            _, ws_g = ex.workspaces(hi - lo, max(h - l for l, h in parts), P)
            kg, vg = ex.gather_buffers(ws_g, W)
            counts = [(h - l) * P for l, h in parts]
            launch = lambda: ex.attend_remote(0, ws_g, kg, vg, counts, self.rank, n)
            exchange = lambda: [self._all_gather(kg.flatten(0, 1), ex._wire(ws_g.k), async_op=True), self._all_gather(vg.flatten(0, 1), ex._wire(ws_g.vt), async_op=True)]
        saved, table = ex.cus, {}
        dev_cus = torch.cuda.get_device_properties(device).multi_processor_count
        try:
            for c in cus_values:
                ex.cus = 0 if c >= dev_cus else int(c)
                launch()                                               # plan + split workspace for this budget (cached per budget)
                torch.cuda.synchronize(device)
                e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                e[0].record()
                for _ in range(reps):
                    launch()
                e[1].record()
                torch.cuda.synchronize(device)
                alone = e[0].elapsed_time(e[1]) / reps
                under = 0.0
                for _ in range(reps):
                    works = exchange()
                    e[0].record()
                    launch()
                    e[1].record()
                    for w in works:
                        w.wait()
                    torch.cuda.synchronize(device)
                    under += e[0].elapsed_time(e[1]) / reps
                t = torch.tensor([alone, under], dtype=torch.float64, device=device if self._nccl() else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
                table[int(c)] = {"alone_ms": round(float(t[0]), 4), "under_exchange_ms": round(float(t[1]), 4)}
        finally:
            ex.cus = saved
        return table

    def _all_reduce_max(self, t):
        if not self._nccl() and t.is_cuda:
            tmp = t.cpu()
            dist.all_reduce(tmp, op=dist.ReduceOp.MAX, group=self.group)
            t.copy_(tmp)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)

    def compare_modes(self, run, n_views, is_f32=False, on_stage=None):
        """Diagnostic used by bench.py (before it times anything): run one sharded forward in each exchange form the shapes
        admit and report the max-rel difference between them (MAX over ranks) and each form's wall time. `run()` must execute
        one forward and return a tensor of it; on_stage(text) is called before each form (bench.py's watchdog heartbeat).
        No exception handling around collectives: a failing RCCL call must crash the job, not desynchronise it."""
        import time
        saved = self.mode
        report = {"modes": ["allgather"], "seconds": {}}

        def timed(mode):
            self.mode = mode
            if on_stage is not None:
                on_stage("compare_modes: one forward in the %s form" % mode)
            t0 = time.perf_counter()
            out = run().float().clone()
            if out.is_cuda:
                torch.cuda.synchronize(out.device)
            report["seconds"][mode] = round(time.perf_counter() - t0, 4)
            return out

        try:
            ref = timed("allgather")
            try:
                resolve_mode("heads", n_views, self.world, is_f32)
            except ValueError:
                return report
            got = timed("heads")
            err = torch.tensor([float((got - ref).abs().max() / ref.abs().max().clamp(min=1e-30))], device=ref.device)
            self._all_reduce_max(err)
            report["modes"].append("heads")
            report["max_rel_heads_vs_allgather"] = float(err.item())
            return report
        finally:
            self.mode = saved

    # ---------------------------------------------------------------------------------------------------------
    def forward(self, agg, images, extrinsics, intrinsics, depth, mask, depth_gt_index, camera_gt_index):
        B, S = images.shape[:2]
        if B != 1:
            # (B, S, ...) batches (omnivggt.py:31-32): every batch entry is its own global sequence (aggregator.py:317-318), its own depth
            # statistics and its own camera frame, and the index lists are shared -- so the view axis of EACH entry is sharded exactly like a
            # B = 1 call, one entry after the other (same collectives in the same order on every rank), and the per-entry outputs are stacked.
            # An entry's S views already occupy every rank; running the entries concurrently would only interleave their exchanges.
            def entry(t, b):
                return None if t is None else t[b:b + 1]
            per = [self.forward(agg, images[b:b + 1], entry(extrinsics, b), entry(intrinsics, b), entry(depth, b), entry(mask, b),
                                depth_gt_index, camera_gt_index)[0] for b in range(B)]
            return [torch.cat([o[i] for o in per], dim=0) for i in range(len(per[0]))], agg.patch_start_idx
        if S < self.world:
            raise ValueError("fewer views (%d) than ranks (%d)" % (S, self.world))
        if self.world > ops.L.OVG_MAX_SEG:
            raise ValueError("at most %d ranks per attention call" % ops.L.OVG_MAX_SEG)
        # every check that can fail happens here, identically on every rank, BEFORE the first collective
        mode = resolve_mode(self.mode, S, self.world, _parity_mode(agg))
        self.last_mode = mode
        agg.set_geometry(images.shape[-2], images.shape[-1])
        P = agg.tokens_per_view
        parts = partition(S, self.world)
        self.last_partition = parts
        lo, hi = parts[self.rank]
        ex = self.executor(agg, images.device)
        inputs = (images, extrinsics, intrinsics, depth, mask, depth_gt_index, camera_gt_index)
        with torch.no_grad():
            if mode == "heads":
                outs = self._forward_heads(agg, ex, inputs, lo, hi, P)
            else:
                outs = self._forward_allgather(agg, ex, inputs, parts, lo, hi, P)
            if self.gather_output:
                outs = [self.gather_views(o, parts) for o in outs]
        return outs, agg.patch_start_idx

    def _forward_allgather(self, agg, ex, inputs, parts, lo, hi, P):
        n_local, max_local = hi - lo, max(h - l for l, h in parts)
        n = n_local * P
        counts = [(h - l) * P for l, h in parts]
        remote = self.world > 1
        tokens0, tables = ex.embed(inputs, (lo, hi))
        ws_f, ws_g = ex.workspaces(n_local, max_local, P)
        kg, vg = ex.gather_buffers(ws_g, self.world)
        outs = ex.new_outputs(n_local, P)
        x = tokens0
        for i in range(agg.depth):
            buf = outs[i].view(n, 2 * C)
            ex.frame_block(i, ws_f, x, buf[:, :C], tables[i + 1][lo:hi].contiguous(), P)
            k_loc, vt_loc = ex.global_kv(i, ws_g, buf[:, :C], buf[:, C:])
            if remote:
                wk = self._all_gather(kg.flatten(0, 1), k_loc, async_op=True)
                wv = self._all_gather(vg.flatten(0, 1), vt_loc, async_op=True)
            ex.global_q(i, ws_g, buf[:, :C], buf[:, C:])              # overlaps the all-gather
            ex.attend_local(i, ws_g, n, want_lse=remote)              # so does the attention over the local keys
            if remote:
                wk.wait()
                wv.wait()
                ex.attend_remote(i, ws_g, kg, vg, counts, self.rank, n)
            ex.merge_finish(i, ws_g, buf[:, :C], buf[:, C:], n, merged=remote)
            x = buf[:, C:]
        return outs

    def _forward_heads(self, agg, ex, inputs, lo, hi, P):
        """Head-parallel global attention, pipelined in head groups (module docstring)."""
        n_local = hi - lo
        n = n_local * P
        W = self.world
        hpr = 16 // W
        tokens0, tables = ex.embed(inputs, (lo, hi))
        ws_f, ws_g, xb = ex.heads_workspaces(n_local, P, W)
        groups, o_back = xb["groups"], xb["o_back"]
        outs = ex.new_outputs(n_local, P)
        x = tokens0
        for i in range(agg.depth):
            buf = outs[i].view(n, 2 * C)
            ex.frame_block(i, ws_f, x, buf[:, :C], tables[i + 1][lo:hi].contiguous(), P)
            q, k, vt = ex.global_qkv(i, ws_g, buf[:, :C], buf[:, C:])
            inbound = []
            for g in groups:                                          # all inbound exchanges queue up on RCCL's stream
                sl = [slice(r * hpr + g["h0"], r * hpr + g["h0"] + g["gs"]) for r in range(W)]
                inbound.append([self._exchange_many([(list(g["q"].unbind(0)), [q[s] for s in sl]), (list(g["k"].unbind(0)), [k[s] for s in sl]),
                                                     (list(g["vt"].unbind(0)), [vt[s] for s in sl])], async_op=True)])   # q | k | v^T: one grouped launch
            returns = []
            for g, works in zip(groups, inbound):
                for w in works:
                    w.wait()                                          # compute stream waits for THIS group only
                ex.head_attention(g["q"], g["k"], g["vt"], g["o"], n, W)
                back = [o_back[r * hpr + g["h0"]: r * hpr + g["h0"] + g["gs"]] for r in range(W)]
                returns.append(self._all_to_all_chunks(back, list(g["o"].unbind(0)), async_op=True))   # under the next group's attention
            for w in returns:
                w.wait()
            ex.global_finish(i, ws_g, buf[:, :C], buf[:, C:], o_back, n)
            x = buf[:, C:]
        return outs

    def exchange_only(self, agg, S, device, mode=None, layers=24):
        """bench.py: issue ONLY the collectives of `layers` global blocks (same sizes, same order, nothing to hide
        behind) on the cached buffers -- the un-overlapped cost of the exchange."""
        mode = resolve_mode(mode or self.mode, S, self.world, _parity_mode(agg))
        P = agg.tokens_per_view
        parts = partition(S, self.world)
        lo, hi = parts[self.rank]
        ex = self.executor(agg, device)
        W, hpr = self.world, 16 // self.world
        if mode == "heads":
            _, ws_g, xb = ex.heads_workspaces(hi - lo, P, W)
            for _ in range(layers):
                works = []
                for g in xb["groups"]:
                    sl = [slice(r * hpr + g["h0"], r * hpr + g["h0"] + g["gs"]) for r in range(W)]
                    works.append(self._exchange_many([(list(g["q"].unbind(0)), [ws_g.q[s] for s in sl]), (list(g["k"].unbind(0)), [ws_g.k[s] for s in sl]),
                                                      (list(g["vt"].unbind(0)), [ws_g.vt[s] for s in sl])], async_op=True))
                for g in xb["groups"]:
                    back = [xb["o_back"][r * hpr + g["h0"]: r * hpr + g["h0"] + g["gs"]] for r in range(W)]
                    works.append(self._all_to_all_chunks(back, list(g["o"].unbind(0)), async_op=True))
                for w in works:
                    w.wait()
        else:
            _, ws_g = ex.workspaces(hi - lo, max(h - l for l, h in parts), P)
            kg, vg = ex.gather_buffers(ws_g, W)
            for _ in range(layers):
                wk = self._all_gather(kg.flatten(0, 1), ex._wire(ws_g.k), async_op=True)
                wv = self._all_gather(vg.flatten(0, 1), ex._wire(ws_g.vt), async_op=True)
                wk.wait()
                wv.wait()
        return mode

    def gather_views(self, local, parts):
        """all-gather a (B, n_local, ...) tensor along the view axis (uneven shards padded)."""
        max_local = max(h - l for l, h in parts)
        pad = torch.zeros((local.shape[0], max_local) + tuple(local.shape[2:]), device=local.device, dtype=local.dtype)
        pad[:, : local.shape[1]] = local
        full = torch.empty((self.world,) + tuple(pad.shape), device=local.device, dtype=local.dtype)
        skip, self.skip_comm = self.skip_comm, False
        try:
            self._all_gather(full.flatten(0, 1), pad)
        finally:
            self.skip_comm = skip
        return torch.cat([full[r][:, : h - l] for r, (l, h) in enumerate(parts)], dim=1)
