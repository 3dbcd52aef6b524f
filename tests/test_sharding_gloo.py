"""World-size-2 / 4 / 8 CPU (gloo) runs of the view-sharded control flow
(omnivggt-official_amd/sharding.py) with an ORACLE-backed executor in place of the HIP one:
validates the partition, the K/V^T all-gather plumbing (uneven shards, padded buffers,
per-segment valid counts, rank-ordered segments), the camera-token gather and the gathered
output -- against the monolithic oracle.  Test infrastructure only: the product executor
(HipExecutor) has no CPU path."""
import math
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

import aggregator_oracle as orc
import common
from omnivggt_official_amd import sharding

DEPTH, DINO = 2, 1
LOG2E = 1.4426950408889634


class OracleExecutor:
    """Same interface as sharding.HipExecutor, numerics by the CPU oracle (fp32)."""

    def __init__(self, sd, depth):
        self.sd, self.depth = sd, depth

    def embed(self, inputs, view_slice):
        st = orc.aggregator_prepare(self.sd, *inputs, dino_layers=DINO)
        self.st = st
        lo, hi = view_slice
        self.lo, self.hi = lo, hi
        tables = [orc.camera_injection(self.sd, i, st).reshape(st["K"], -1) for i in range(self.depth + 1)]
        return st["tokens"][lo:hi].reshape(-1, 1024).clone(), tables

    def new_outputs(self, n_local, P):
        return [torch.zeros(1, n_local, P, 2048) for _ in range(self.depth)]

    def workspaces(self, n_local, max_local, P):
        pad = (max_local * P + 63) // 64 * 64
        ws = type("WS", (), {})()
        ws.n, ws.P = n_local, P
        ws.q = torch.zeros(16, pad, 64)
        ws.k = torch.zeros(16, pad, 64)
        ws.vt = torch.zeros(16, 64, pad)
        return ws, ws

    def gather_buffers(self, ws, world):
        return torch.zeros((world,) + tuple(ws.k.shape)), torch.zeros((world,) + tuple(ws.vt.shape))

    def frame_block(self, i, ws, x_in, x_out, inject, P):
        n = x_in.shape[0] // P
        pos = self.st["pos"][self.lo:self.hi]
        y = orc.block(x_in.reshape(n, P, 1024), self.sd, "aggregator.frame_blocks.%d" % i, pos, self.st["rope"], True)
        y = y.clone()
        y[:, 0] += inject
        x_out.copy_(y.reshape(-1, 1024))

    def _qkv(self, i, x_in):
        pre = "aggregator.global_blocks.%d" % i
        n = x_in.shape[0]
        xn = orc.layer_norm(x_in, self.sd, pre + ".norm1", 1e-5)
        qkv = F.linear(xn, self.sd[pre + ".attn.qkv.weight"], self.sd[pre + ".attn.qkv.bias"]).reshape(1, n, 3, 16, 64).permute(2, 0, 3, 1, 4)
        q = orc.layer_norm(qkv[0], self.sd, pre + ".attn.q_norm", 1e-5)
        k = orc.layer_norm(qkv[1], self.sd, pre + ".attn.k_norm", 1e-5)
        pos = self.st["pos"][self.lo:self.hi].reshape(1, n, 2)
        q = orc.rope_2d(q, pos, *self.st["rope"])
        k = orc.rope_2d(k, pos, *self.st["rope"])
        return q[0] * (0.125 * LOG2E), k[0], qkv[2][0]

    def global_kv(self, i, ws, x_in, x_out):
        q, k, v = self._qkv(i, x_in.contiguous())
        n = k.shape[1]
        ws.k.zero_()
        ws.vt.zero_()
        ws.k[:, :n] = k
        ws.vt[:, :, :n] = v.transpose(1, 2)
        self._q = q
        return ws.k, ws.vt

    def global_q(self, i, ws, x_in, x_out):
        ws.q[:, : self._q.shape[1]] = self._q

    @staticmethod
    def _attend(q, keys, vals):
        """softmax attention in base 2 (q pre-scaled by log2 e) + the per-row log2-sum-exp, like the HIP kernel."""
        s = q @ keys.transpose(1, 2)                                   # [16, n, nk] log2-domain logits
        lse = torch.logsumexp(s * math.log(2.0), dim=-1) / math.log(2.0)
        return torch.softmax(s * math.log(2.0), dim=-1) @ vals, lse

    def attend_local(self, i, ws, n, want_lse):
        o, lse = self._attend(ws.q[:, :n], ws.k[:, :n], ws.vt[:, :, :n].transpose(1, 2))
        self._oa, self._lse_a = o, lse

    def attend_remote(self, i, ws, kg, vg, counts, rank, n):
        assert torch.equal(kg[rank], ws.k) and torch.equal(vg[rank], ws.vt)
        keys = torch.cat([kg[r][:, :c] for r, c in enumerate(counts) if r != rank], dim=1)
        vals = torch.cat([vg[r][:, :, :c].transpose(1, 2) for r, c in enumerate(counts) if r != rank], dim=1)
        self._ob, self._lse_b = self._attend(ws.q[:, :n], keys, vals)

    def merge_finish(self, i, ws, x_in, x_out, n, merged):
        o = self._oa
        if merged:                                                     # ovg_attn_merge's formula
            m = torch.maximum(self._lse_a, self._lse_b)
            wa, wb = torch.exp2(self._lse_a - m).unsqueeze(-1), torch.exp2(self._lse_b - m).unsqueeze(-1)
            o = (wa * self._oa + wb * self._ob) / (wa + wb)
        self._post(i, x_in, x_out, o.permute(1, 0, 2).reshape(n, 1024))

    def _post(self, i, x_in, x_out, o):
        pre = "aggregator.global_blocks.%d" % i
        x = x_in + F.linear(o, self.sd[pre + ".attn.proj.weight"], self.sd[pre + ".attn.proj.bias"]) * self.sd[pre + ".ls1.gamma"]
        m = orc.mlp(orc.layer_norm(x, self.sd, pre + ".norm2", 1e-5), self.sd, pre + ".mlp")
        x_out.copy_(x + m * self.sd[pre + ".ls2.gamma"])

    # ---- head-parallel (all-to-all) mode ----
    def heads_workspaces(self, n_local, P, world):
        ws, _ = self.workspaces(n_local, n_local, P)
        pad = ws.q.shape[1]
        nan = lambda *s: torch.full(s, float("nan"))
        groups = [{"h0": h0, "gs": gs, "q": nan(world, gs, pad, 64), "k": nan(world, gs, pad, 64), "vt": nan(world, gs, 64, pad),
                   "o": nan(world, gs, pad, 64)} for h0, gs in sharding.head_groups(16 // world, world, n_local * P)]
        return ws, ws, {"groups": groups, "o_back": nan(16, pad, 64)}

    def global_qkv(self, i, ws, x_in, x_out):
        self.global_kv(i, ws, x_in, x_out)
        ws.q.zero_()
        ws.q[:, : self._q.shape[1]] = self._q
        return ws.q, ws.k, ws.vt

    def head_attention(self, qr, kr, vr, out, n, world):
        gs = qr.shape[1]
        for s in range(world):
            for h in range(gs):
                keys = torch.cat([kr[r, h][:n] for r in range(world)], dim=0)
                vals = torch.cat([vr[r, h][:, :n].t() for r in range(world)], dim=0)
                sc = (qr[s, h][:n] @ keys.t()) * math.log(2.0)
                out[s, h].zero_()
                out[s, h][:n] = torch.softmax(sc, dim=-1) @ vals
        return out

    def global_finish(self, i, ws, x_in, x_out, o_back, n):
        assert not torch.isnan(o_back[:, :n]).any()                   # every head of every group came back
        self._post(i, x_in, x_out, o_back[:, :n].permute(1, 0, 2).reshape(n, 1024))


class FakeAgg:
    depth, tokens_per_view, patch_start_idx = DEPTH, 1374, 5

    def set_geometry(self, H, W):                      # same contract as ZeroAggregator.set_geometry
        self.grid_hw = (H // 14, W // 14)
        self.tokens_per_view = self.grid_hw[0] * self.grid_hw[1] + self.patch_start_idx
        return self.grid_hw


def batch_inputs(B, S, hw):
    """B different scenes stacked on the batch axis (as tests/test_gpu_aggregator.py batch_inputs)."""
    parts = [orc.synthetic_inputs(S, seed=1234 + 1111 * b, hw=hw) for b in range(B)]
    return {k: torch.cat([q[k] for q in parts], 0) for k in parts[0]}


def _worker(rank, world, port, S, dgi, cgi, result_dir, hw=518, mode="auto", B=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 2) // world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sd = common.reduced_state_dict(DEPTH, DINO)
        inp = orc.synthetic_inputs(S, hw=hw) if B == 1 else batch_inputs(B, S, hw)
        sh = sharding.ViewSharding(executor_factory=lambda agg, dev: OracleExecutor(sd, DEPTH), gather_output=True,
                                   mode="auto" if mode == "choose" else ("heads" if mode == "heads_bad" else mode))
        fa = FakeAgg()
        fwd = lambda: sh.forward(fa, inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], dgi, cgi)
        if mode == "choose":                           # bench.py's diagnostic: both exchange forms on the same input must agree
            rep = sh.compare_modes(lambda: fwd()[0][-1], S)
            assert rep["modes"] == ["allgather", "heads"] and rep["max_rel_heads_vs_allgather"] < 1e-5, rep
            assert sh.mode == "auto"
        if mode == "heads_bad":                        # impossible request: every rank raises BEFORE any collective
            with pytest.raises(ValueError):
                fwd()
            dist.barrier()                             # ... and the group is still usable afterwards
            sh.mode = "auto"
        outs, start = fwd()
        assert start == 5 and sh.last_partition == sharding.partition(S, world)
        expect = {"heads_bad": "allgather", "choose": "heads", "auto": "heads" if S % world == 0 else "allgather"}.get(mode, mode)
        assert sh.last_mode == expect, (sh.last_mode, expect)
        if rank == 0:
            torch.save([o.clone() for o in outs], os.path.join(result_dir, "sharded.pt"))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("S,dgi,cgi,hw,mode", [(3, [1], [0, 2], 518, "auto"), (3, [0], [1], (266, 350), "allgather"),
                                                 (2, [1], [0, 1], (266, 350), "choose"), (4, [0, 3], [1, 2], (210, 266), "heads"),
                                                 (3, [], [0], (210, 266), "heads_bad")])
def test_view_sharded_forward_matches_monolithic_oracle(tmp_path, S, dgi, cgi, hw, mode):
    """2 ranks: uneven splits (3 views -> K/V all-gather path: local-first launch, remote launch, log-sum-exp merge)
    and even splits (-> head-parallel all-to-all path, 8 heads per rank in two pipelined groups of 4); non-square,
    non-trained patch grids in all but the first case; an impossible explicit mode fails before any collective."""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), S, dgi, cgi, str(tmp_path), hw, mode), nprocs=world, join=True)
    sharded = torch.load(os.path.join(str(tmp_path), "sharded.pt"))
    sd = common.reduced_state_dict(DEPTH, DINO)
    inp = orc.synthetic_inputs(S, hw=hw)
    P = 1374 if hw == 518 else (hw[0] // 14) * (hw[1] // 14) + 5
    with torch.no_grad():
        ref, _ = orc.aggregator_forward(sd, inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], dgi, cgi,
                                        depth_layers=DEPTH, dino_layers=DINO)
    assert len(sharded) == DEPTH
    for a, b in zip(sharded, ref):
        assert a.shape == b.shape == (1, S, P, 2048)
        assert common.max_rel(a, b) < 2e-5


@pytest.mark.parametrize("S,dgi,cgi,mode", [(2, [1], [0, 1], "heads"), (3, [0, 2], [1], "allgather")])
def test_view_sharded_batch_of_two_scenes(tmp_path, S, dgi, cgi, mode):
    """Round-5 review item 5: (B, S, ...) batches on the sharded path (omnivggt.py:31-32; the single-GPU form is covered by
    tests/test_gpu_aggregator.py test_batch_of_two_scenes_*). 2 ranks x B = 2 different scenes, even split through the head-parallel form
    and an uneven split (3 views) through the K / V^T all-gather form, against the monolithic oracle on the same (2, S, ...) tensors:
    per-entry depth statistics, per-entry camera frame, per-entry global sequence, shared index lists."""
    world, hw, B = 2, (210, 266), 2
    mp.spawn(_worker, args=(world, _free_port(), S, dgi, cgi, str(tmp_path), hw, mode, B), nprocs=world, join=True)
    sharded = torch.load(os.path.join(str(tmp_path), "sharded.pt"))
    sd = common.reduced_state_dict(DEPTH, DINO)
    inp = batch_inputs(B, S, hw)
    P = (hw[0] // 14) * (hw[1] // 14) + 5
    with torch.no_grad():
        ref, _ = orc.aggregator_forward(sd, inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], dgi, cgi,
                                        depth_layers=DEPTH, dino_layers=DINO)
    assert len(sharded) == DEPTH
    for a, b in zip(sharded, ref):
        assert a.shape == b.shape == (B, S, P, 2048)
        for e in range(B):
            assert common.max_rel(a[e], b[e]) < 2e-5
    assert common.max_rel(ref[-1][0], ref[-1][1]) > 1e-2          # the two entries ARE different scenes


@pytest.mark.parametrize("world,S,dgi,cgi,mode", [(4, 8, [1, 6], [0, 5], "auto"), (4, 6, [5], [0, 3], "auto"),
                                                    (8, 8, [2], [0, 7], "heads"), (8, 11, [0, 10], [1, 9], "allgather"),
                                                    (8, 16, [], [3], "choose")])
def test_view_sharded_forward_at_world_4_and_8(tmp_path, world, S, dgi, cgi, mode):
    """The rank counts of the scaling bench (round-2 review: only world 2 was covered): 4 ranks even (head-parallel, 4 heads
    per rank in two pipelined groups of 2) and uneven (2/2/1/1 -> all-gather with padded shards); 8 ranks head-parallel
    (2 heads per rank, one group, 8 inbound chunks per exchange), 8 ranks uneven (2/2/2/1/1/1/1/1: launch B walks 7 remote
    segments of two different lengths) and bench.py's pre-flight (both forms on the same input, compared across ranks).
    Small frames (210 x 266) keep 8 single-threaded oracle processes inside the CPU budget."""
    hw = (210, 266)
    mp.spawn(_worker, args=(world, _free_port(), S, dgi, cgi, str(tmp_path), hw, mode), nprocs=world, join=True)
    sharded = torch.load(os.path.join(str(tmp_path), "sharded.pt"))
    sd = common.reduced_state_dict(DEPTH, DINO)
    inp = orc.synthetic_inputs(S, hw=hw)
    P = (hw[0] // 14) * (hw[1] // 14) + 5
    with torch.no_grad():
        ref, _ = orc.aggregator_forward(sd, inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], dgi, cgi,
                                        depth_layers=DEPTH, dino_layers=DINO)
    for a, b in zip(sharded, ref):
        assert a.shape == b.shape == (1, S, P, 2048)
        assert common.max_rel(a, b) < 2e-5
